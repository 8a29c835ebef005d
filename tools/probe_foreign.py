"""Frames from the CPU checker's encoder (reference + libsnappy where oracle/_ref is built) decoded on the GPU:
time of the block scan and of the decode kernels, for one frame and for a batch, with and without the scan.
    python tools/probe_foreign.py [frames] [width height]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hap_amd
from hap_amd import synth
import _libs as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (8192, 4096)
fmt, chunks = L.FMT_DXT5, 24
ctx = hap_amd.Context(0)
nbytes = (w // 4) * (h // 4) * 16
api = L.ref_api() or L.oracle_api()
tex, frames = [], []
for f in range(n):
    rgba = synth.rgba_frame(w, h, f, device="cuda")
    t = torch.zeros(nbytes, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    assert ctx.compress_rgba(rgba, w, h, w * 4, fmt, t) == (0, nbytes)
    tex.append(t)
    r, frame = api.encode_np([t.cpu().numpy()], [fmt], [1], [chunks])
    assert r == 0
    frames.append(torch.from_numpy(frame).cuda())
print("frame bytes", frames[0].numel(), "ratio %.4f" % (frames[0].numel() / nbytes))
outs = [torch.zeros(nbytes, dtype=torch.uint8, device="cuda") for _ in range(n)]
torch.cuda.synchronize()
for count in sorted({1, n}):
    for flags, name in ((hap_amd.DECODE_NO_BLOCK_SCAN, "whole streams"), (0, "block scan")):
        fr, ou = frames[:count], outs[:count]
        lens = [x.numel() for x in fr]
        ctx.decode_frames(fr, lens, 0, ou, flags)
        ctx.set_profiling(True); ctx.collect_profile()
        reps = 3
        for _ in range(reps):
            assert ctx.decode_frames(fr, lens, 0, ou, flags)[0] == 0
        prof = ctx.collect_profile(); ctx.set_profiling(False)
        ctx.timer_start()
        ctx.decode_frames(fr, lens, 0, ou, flags)
        wall = ctx.timer_stop()
        ok = all(torch.equal(ou[i], tex[i]) for i in range(count))
        blocks = sum((nbytes // chunks + 65535) // 65536 * chunks for _ in range(count))
        r0 = ctx.resolved_blocks()
        for o in ou:
            o.zero_()
        torch.cuda.synchronize()
        ctx.decode_frames(fr, lens, 0, ou, flags)
        ok = ok and all(torch.equal(ou[i], tex[i]) for i in range(count))
        print("%2d frame(s) %-13s: scan %.3f ms  decode %.3f ms  plan %.3f ms  call %.3f ms  %s  (64 KiB blocks %d, by a workgroup each %d)" % (
            count, name, prof["block_scan"][1] / reps, prof["snappy_decode"][1] / reps, prof["decode_plan"][1] / reps,
            wall, "ok" if ok else "MISMATCH", blocks, ctx.resolved_blocks() - r0))
