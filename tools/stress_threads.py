"""Dev stress run: several host threads at once -- each with a context of its own (batched calls: pictures and textures in,
frames out, decode, compare) or through the plain hap.h entry points (the default-context pool) -- for a given time.
    python tools/stress_threads.py [threads] [seconds]"""
import os, sys, threading, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _libs as L, hap_amd
nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
os.environ.setdefault("HAP_AMD_PLACING_MIN_FRAMES", "2")
fails, rounds = [], [0] * nthreads
geoms = [(256, 128), (512, 256), (1024, 256), (640, 360 // 4 * 4)]
pics = {g: [D.rgba(g[0], g[1], frame=i) for i in range(4)] for g in geoms}
texs = {(g, f): [D.oracle_bc_encode(p, f) for p in pics[g]] for g in geoms for f in (L.FMT_YCOCG, L.FMT_DXT5, L.FMT_DXT1)}

def batched(k):
    rng = np.random.default_rng(100 + k)
    ctx = hap_amd.Context(0)
    t0 = time.time()
    while time.time() - t0 < budget and not fails:
        g = geoms[int(rng.integers(0, len(geoms)))]; fmt = [L.FMT_YCOCG, L.FMT_DXT5, L.FMT_DXT1][int(rng.integers(0, 3))]
        nf = int(rng.integers(1, 5)); chunks = int(rng.integers(1, 7)); w, h = g
        want = texs[(g, fmt)][:nf]; n = len(want[0])
        cap = hap_amd.HapMaxEncodedLength([n], [fmt], [chunks])
        dev = [torch.from_numpy(np.ascontiguousarray(p).reshape(-1)).cuda() for p in pics[g][:nf]]
        outs = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
        torch.cuda.synchronize()
        flags = hap_amd.ENCODE_FRAGMENT_INDEX if rng.integers(0, 2) else 0
        r, used, res = ctx.encode_frames_rgba(dev, w, h, w * 4, [fmt], [1], [chunks], outs, flags=flags)
        ok = r == 0 and all(x == 0 for x in res)
        if ok:
            decs = [torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(nf)]
            torch.cuda.synchronize()
            r, du, df, dr = ctx.decode_frames(outs, used, 0, decs)
            ok = r == 0 and all(decs[i].cpu().numpy().tobytes() == want[i] for i in range(nf))
        if not ok:
            fails.append(("batched", k, g, hex(fmt), nf, chunks, flags))
        rounds[k] += 1

def plain(k):
    rng = np.random.default_rng(200 + k)
    t0 = time.time()
    while time.time() - t0 < budget and not fails:
        g = geoms[int(rng.integers(0, len(geoms)))]; fmt = [L.FMT_YCOCG, L.FMT_DXT5, L.FMT_DXT1][int(rng.integers(0, 3))]
        tex = texs[(g, fmt)][int(rng.integers(0, 4))]; chunks = int(rng.integers(1, 7))
        r, frame = hap_amd.HapEncode([tex], [fmt], [1], [chunks])
        ok = r == 0
        if ok:
            r, out, f2 = hap_amd.HapDecode(frame, 0, outputBufferBytes=len(tex))
            ok = r == 0 and bytes(out) == tex and f2 == fmt
        if not ok:
            fails.append(("plain", k, g, hex(fmt), chunks))
        rounds[k] += 1

th = [threading.Thread(target=(batched if k % 2 == 0 else plain), args=(k,)) for k in range(nthreads)]
[t.start() for t in th]; [t.join(budget + 120) for t in th]
print("stress_threads: %d threads, rounds %s, stuck %d, failures %s" % (nthreads, rounds, sum(t.is_alive() for t in th), fails[:3]))
