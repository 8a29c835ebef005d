"""Dev tool (GPU box): encode-only timing of a bench workload, results not checked (for diagnostic library variants).
   python tools/time_encode.py C4 [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, hap_amd, bench

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else bench.CONFIGS[cfg][4]
ctx = hap_amd.Context()
s = bench.Stream(hap_amd, ctx, "cuda", cfg, list(range(nf)), hap_amd.ENCODE_FRAGMENT_INDEX)
for _ in range(2):
    try:
        s.encode()
    except RuntimeError as e:
        print("encode:", e)
ctx.set_profiling(True)
ctx.collect_profile()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 4
for _ in range(n):
    try:
        s.encode()
    except RuntimeError:
        pass
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
prof = {k: round(v[1] / max(v[0], 1), 3) for k, v in ctx.collect_profile().items() if v[0]}
print(os.environ.get("HAP_AMD_LIBRARY", "default").split("/")[-1], cfg, "encode call ms %.3f" % (dt * 1e3), prof)
