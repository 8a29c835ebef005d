"""GPU probe: fragments written straight to their places in the frame (HapGpuTexEnc.reserved bit 27): timing of the
encode call, how many frames had to be encoded again.    python tools/probe_placed.py [C4|C2|C3|C5] [frames] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
import bench as B
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 60
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
ctx = hap_amd.Context(0)
s = B.Stream(hap_amd, ctx, dev, cfg, list(range(nf)), hap_amd.ENCODE_FRAGMENT_INDEX)
s.used = s.encode()
r0 = ctx.placement_retries()
best = None
for _ in range(reps):
    ctx.timer_start(); s.used = s.encode(); ms = ctx.timer_stop()
    best = ms if best is None else min(best, ms)
ctx.set_profiling(True); ctx.collect_profile()
s.used = s.encode()
prof = ctx.collect_profile(); ctx.set_profiling(False)
try:
    s.decode(s.used); exact = s.bit_exact()
except Exception as exc:
    exact = "decode failed"
print(cfg, nf, "frames: encode call %.3f ms, retries per call %.1f, bit_exact %s, kernels: %s" % (
    best, (ctx.placement_retries() - r0) / (reps + 1.0), exact,
    " ".join("%s %.3f/%d" % (k, v[1], v[0]) for k, v in prof.items() if v[0])))
