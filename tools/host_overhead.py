"""Dev tool: wall time of the batched calls against the GPU time of the same calls (events on the library's stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
from hap_amd import synth
for name, (w, h, fmt, chunks, nf) in {"C2": (3840, 2160, 0x83F0, 1, 60), "C4": (7680, 4320, 0x01, 24, 60)}.items():
    ctx = hap_amd.Context(0)
    tb = (w // 4) * (h // 4) * (8 if fmt == 0x83F0 else 16)
    rgba = [synth.rgba_frame(w, h, i % 6, device="cuda") for i in range(nf)]
    cap = hap_amd.HapMaxEncodedLength([tb], [fmt], [chunks])
    frames = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
    dec = [torch.empty(tb, dtype=torch.uint8, device="cuda") for _ in range(nf)]
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter(); ctx.timer_start()
        r, used, _ = ctx.encode_frames_rgba(rgba, w, h, w * 4, [fmt], [1], [chunks], frames, flags=1)
        ge = ctx.timer_stop(); t1 = time.perf_counter(); ctx.timer_start()
        ctx.decode_frames(frames, used, 0, dec)
        gd = ctx.timer_stop(); t2 = time.perf_counter()
    print(name, "encode wall %.3f ms gpu %.3f ms | decode wall %.3f ms gpu %.3f ms" % ((t1 - t0) * 1e3, ge, (t2 - t1) * 1e3, gd))
