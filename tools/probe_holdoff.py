"""GPU probe: retries per call of a context that holds off placing after a call of mostly incompressible frames."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hap_amd
import _data as D, _libs as L
os.environ["HAP_AMD_PLACING_MIN_FRAMES"] = "1"
ctx = hap_amd.Context(0)
w, h = 1024, 256
size = (w // 4) * (h // 4) * 16
rng = np.random.RandomState(11)
flat = D.oracle_bc_encode(D.rgba(w, h, frame=3), L.FMT_YCOCG)
noise = rng.randint(0, 256, size, dtype=np.uint8).tobytes()
half = flat[: size // 2] + noise[size // 2:]
cap = hap_amd.HapMaxEncodedLength([size], [L.FMT_YCOCG], [4])
order = sys.argv[1] if len(sys.argv) > 1 else "nhf"
texs = {"n": noise, "h": half, "f": flat}
for call in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    douts = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in order]
    dtex = [torch.from_numpy(np.frombuffer(texs[c], dtype=np.uint8).copy()).cuda() for c in order]
    torch.cuda.synchronize()
    r0 = ctx.placement_retries()
    r, used, res = ctx.encode_frames([[t] for t in dtex], [L.FMT_YCOCG], [1], [4], douts, flags=0)
    print(call, "retries", ctx.placement_retries() - r0, r, res, used)
