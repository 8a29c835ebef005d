"""Dev tool: per-kernel-class milliseconds for one C4 batch (encode + decode) with the library selected by
HAP_AMD_LIBRARY -- used for A/B runs of experimental builds.  usage: time_kernels.py [frames] [--no-decode]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hap_amd
from hap_amd import synth
nf = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
w, h, fmts, chunks = 7680, 4320, [0x01], [24]
ctx = hap_amd.Context(0)
for a in sys.argv:
    if a.startswith("--frag="):
        ctx.set_fragment_log2(int(a[7:]))
rgba = [synth.rgba_frame(w, h, i, device="cuda") for i in range(nf)]
tb = w * h
cap = hap_amd.HapMaxEncodedLength([tb], fmts, chunks)
frames = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
dec = [torch.empty(tb, dtype=torch.uint8, device="cuda") for _ in range(nf)]
torch.cuda.synchronize()
for rep in range(3):
    if rep == 1:
        ctx.set_profiling(True); ctx.collect_profile()
    r, used, res = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, [1], chunks, frames, flags=1)
    if "--no-decode" not in sys.argv:
        ctx.decode_frames(frames, used, 0, dec)
prof = ctx.collect_profile()
print(os.path.basename(os.environ.get("HAP_AMD_LIBRARY", "default")), "frames", nf, "ratio %.4f" % (sum(used) / (tb * nf)),
      " ".join("%s=%.3f" % (k, v[1] / 2) for k, v in prof.items() if v[0]))
