"""Dev tool: fixed host-side cost of the batched calls (tiny frames, so that GPU work is negligible)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
from hap_amd import synth
w, h, fmt, chunks = 64, 64, 0x01, 1
ctx = hap_amd.Context(0)
for nf in (1, 60, 240):
    tb = (w // 4) * (h // 4) * 16
    rgba = hap_amd.BufferList([synth.rgba_frame(w, h, i % 6, device="cuda") for i in range(nf)])
    cap = hap_amd.HapMaxEncodedLength([tb], [fmt], [chunks])
    frames = hap_amd.BufferList([torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)])
    dec = hap_amd.BufferList([torch.empty(tb, dtype=torch.uint8, device="cuda") for _ in range(nf)])
    torch.cuda.synchronize()
    best_e = best_d = 1e9
    for rep in range(20):
        t0 = time.perf_counter()
        r, used, _ = ctx.encode_frames_rgba(rgba, w, h, w * 4, [fmt], [1], [chunks], frames, flags=1)
        t1 = time.perf_counter()
        ctx.decode_frames(frames, used, 0, dec)
        t2 = time.perf_counter()
        best_e = min(best_e, t1 - t0); best_d = min(best_d, t2 - t1)
    print("frames %3d: encode call %.3f ms, decode call %.3f ms" % (nf, best_e * 1e3, best_d * 1e3))
