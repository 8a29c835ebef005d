"""Dev stress run: many random textures through encode -> decode (field streams, records in LDS and in memory,
generic streams) and checker-made frames through the block scan, batches of several frames, every result compared.
    python tools/stress.py [seed] [seconds]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _data as D, _libs as L, hap_amd
ORA = L.oracle_api()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
ctx = hap_amd.Context(0)
FMTS = [(L.FMT_DXT1, 8), (L.FMT_DXT5, 16), (L.FMT_YCOCG, 16), (L.FMT_RGTC1, 8), (L.FMT_BC7, 16)]
t0 = time.time(); rounds = frames = fails = 0
while time.time() - t0 < budget:
    fmt, block = FMTS[int(rng.integers(0, len(FMTS)))]
    nf = int(rng.integers(1, 5))
    nblocks = int(rng.integers(1, 40000)) if rng.integers(0, 4) else int(rng.integers(100000, 300000))
    chunks = int(rng.integers(1, 9))
    texs = []
    for i in range(nf):
        kind = ["zero", "random", "mixed", "runs"][int(rng.integers(0, 4))]
        t = bytearray(D.stream_bytes(nblocks * block, kind, seed=int(rng.integers(0, 1 << 30))))
        if rng.integers(0, 2) and block == 16:            # constant endpoints, noisy indices: poorly compressible field streams
            a = np.frombuffer(bytes(t), dtype=np.uint8).reshape(-1, 16).copy()
            a[:, 0:2] = 7; a[:, 8:12] = 9
            a[:, 2:8] = rng.integers(0, 256, (a.shape[0], 6), dtype=np.uint8)
            t = bytearray(a.tobytes())
        texs.append(bytes(t))
    n = nblocks * block
    fine = rng.integers(0, 4) == 0                      # (round 5: a chunk per fragment, decoded with and without the pre-pass)
    if fine:
        chunks = max(1, hap_amd.fine_chunk_count(n, fmt))
    cap = hap_amd.HapMaxEncodedLength([n], [fmt], [chunks]) + 4096
    # ours: encode a batch, decode the batch, compare; the checker must agree on the first frame
    outs = [np.zeros(cap, dtype=np.uint8) for _ in range(nf)]
    flags = hap_amd.ENCODE_FRAGMENT_INDEX if rng.integers(0, 4) else 0
    if rng.integers(0, 3) == 0:
        flags |= hap_amd.ENCODE_COARSE_MATCHES          # (BC7 / BC6H: the block kernels with four dwords per block)
    if fine:
        flags |= hap_amd.ENCODE_FINE_CHUNKS
    dflags = [0, hap_amd.DECODE_GUESS_FIELDS, hap_amd.DECODE_NO_FIELD_GUESS][int(rng.integers(0, 3))]
    r, used, res = ctx.encode_frames([[t] for t in texs], [fmt], [1], [chunks], outs, flags=flags)
    ok = r == 0 and all(x == 0 for x in res)
    if ok:
        encoded = [outs[i][: used[i]].tobytes() for i in range(nf)]
        # the same call twice more: the second records the launch sequence, the third replays it -- same frames
        for _again in range(2):
            outs2 = [np.zeros(cap, dtype=np.uint8) for _ in range(nf)]
            r2, used2, res2 = ctx.encode_frames([[t] for t in texs], [fmt], [1], [chunks], outs2, flags=flags)
            ok = ok and r2 == 0 and [outs2[i][: used2[i]].tobytes() for i in range(nf)] == encoded
        decs = [np.zeros(n, dtype=np.uint8) for _ in range(nf)]
        r, du, df, dr = ctx.decode_frames(encoded, [len(e) for e in encoded], 0, decs, flags=dflags)
        ok = r == 0 and all(decs[i].tobytes() == texs[i] for i in range(nf)) and ORA.decode(encoded[0], 0, n) == (0, texs[0], fmt)
    # the checker's frames: block scan path (several blocks per chunk when the texture is large)
    if ok:
        foreign = [ORA.encode([t], [fmt], [1], [chunks])[1] for t in texs]
        decs = [np.zeros(n, dtype=np.uint8) for _ in range(nf)]
        r, du, df, dr = ctx.decode_frames(foreign, [len(e) for e in foreign], 0, decs, flags=dflags)
        ok = r == 0 and all(decs[i].tobytes() == texs[i] for i in range(nf))
    # pictures: blocks made inside the compressor (DXT5 / YCoCg), or by the block encoder's own pass; noise pictures
    # give chunks that do not shrink (frames encoded again through slots when the fragments were placed)
    if ok and rng.integers(0, 3) == 0:
        pf = [L.FMT_DXT1, L.FMT_DXT5, L.FMT_YCOCG][int(rng.integers(0, 3))]
        w, h = 4 * int(rng.integers(1, 160)), 4 * int(rng.integers(1, 120))
        pics = []
        for i in range(nf):
            if rng.integers(0, 3) == 0:
                pics.append(rng.integers(0, 256, (h, w, 4), dtype=np.uint8))
            else:
                pics.append(D.rgba(w, h, frame=int(rng.integers(0, 1000))))
        ptex = [D.oracle_bc_encode(p, pf) for p in pics]
        pn = len(ptex[0])
        pcap = hap_amd.HapMaxEncodedLength([pn], [pf], [chunks]) + 4096
        pouts = [np.zeros(pcap, dtype=np.uint8) for _ in range(nf)]
        r, used, res = ctx.encode_frames_rgba([np.ascontiguousarray(p).reshape(-1) for p in pics], w, h, w * 4, [pf], [1], [chunks], pouts, flags=flags & hap_amd.ENCODE_FRAGMENT_INDEX)
        ok = r == 0 and all(x == 0 for x in res)
        if ok:
            pe = [pouts[i][: used[i]].tobytes() for i in range(nf)]
            dpics = [torch.from_numpy(np.ascontiguousarray(p).reshape(-1)).cuda() for p in pics]
            for _again in range(3):                   # device buffers: the recorded sequence (plain, recorded, replayed)
                dout = [torch.zeros(pcap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
                torch.cuda.synchronize()
                r2, used2, res2 = ctx.encode_frames_rgba(dpics, w, h, w * 4, [pf], [1], [chunks], dout, flags=flags & hap_amd.ENCODE_FRAGMENT_INDEX)
                ok = ok and r2 == 0 and [dout[i][: used2[i]].cpu().numpy().tobytes() for i in range(nf)] == pe
            ok = all(ORA.decode(pe[i], 0, pn) == (0, ptex[i], pf) for i in range(nf))
            decs = [np.zeros(pn, dtype=np.uint8) for _ in range(nf)]
            r, du, df, dr = ctx.decode_frames(pe, [len(e) for e in pe], 0, decs)
            ok = ok and r == 0 and all(decs[i].tobytes() == ptex[i] for i in range(nf))
        if not ok:
            print("  (pictures %dx%d fmt %#x)" % (w, h, pf))
    rounds += 1; frames += nf
    if not ok:
        fails += 1
        print("FAIL round", rounds, "fmt", hex(fmt), "blocks", nblocks, "chunks", chunks, "frames", nf, "flags", flags, "dflags", dflags)
print("stress: %d rounds, %d frames, %d failures, fallbacks %d, placement retries %d, %.0f s" % (
    rounds, frames, fails, ctx.table_fallbacks(), ctx.placement_retries(), time.time() - t0))
