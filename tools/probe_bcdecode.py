"""GPU probe: kernel time of the DXT -> RGBA decoder (bc_decode.hip) per format, HIP events around every launch.
    python tools/probe_bcdecode.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hap_amd
from hap_amd import synth
import _libs as L, _data as D
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = hap_amd.Context(0)
cases = [("8K YCoCg", 7680, 4320, 0x01, None), ("8K DXT5", 7680, 4320, 0x83F3, None), ("8K DXT1", 7680, 4320, 0x83F0, None),
         ("8K YCoCg + RGTC1 alpha", 7680, 4320, 0x01, 0x8DBB), ("16K YCoCg + RGTC1 alpha", 16384, 16384, 0x01, 0x8DBB)]
bb = {0x01: 16, 0x83F3: 16, 0x8DBB: 8, 0x83F0: 8}
for name, w, h, fmt, afmt in cases:
    n = 2 if w > 8000 else nf
    rgba = [synth.rgba_frame(w, h, i, device="cuda") for i in range(n)]
    tex = [torch.empty((w // 4) * (h // 4) * bb[fmt], dtype=torch.uint8, device="cuda") for _ in range(n)]
    alp = [torch.empty((w // 4) * (h // 4) * 8, dtype=torch.uint8, device="cuda") for _ in range(n)] if afmt else None
    out = [torch.empty(w * h * 4, dtype=torch.uint8, device="cuda") for _ in range(n)]
    torch.cuda.synchronize()
    for i in range(n):
        assert ctx.compress_rgba(rgba[i], w, h, w * 4, fmt, tex[i])[0] == 0
        if afmt:
            assert ctx.compress_rgba(rgba[i], w, h, w * 4, afmt, alp[i])[0] == 0
    for i in range(n):
        ctx.decompress_rgba(tex[i], fmt, w, h, rgba=out[i], alpha=alp[i] if afmt else None)
    ctx.set_profiling(True); ctx.collect_profile()
    for rep in range(3):
        for i in range(n):
            ctx.decompress_rgba(tex[i], fmt, w, h, rgba=out[i], alpha=alp[i] if afmt else None)
    cnt, ms = ctx.collect_profile().get("block_decode", (0, 0.0)); ctx.set_profiling(False)
    us = ms / cnt * 1e3
    alg = (w // 4) * (h // 4) * (bb[fmt] + (8 if afmt else 0) + 64)
    rows = 64      # check against the oracle decoder on the first 64 rows of frame 0
    got = out[0][: w * rows * 4].cpu().numpy().reshape(rows, w, 4)
    t = tex[0][: (w // 4) * (rows // 4) * bb[fmt]].cpu().numpy().tobytes()
    want = D.oracle_bc_decode(t, fmt, w, rows)
    if afmt:
        want = want.copy(); want[..., 3] = D.oracle_bc_decode(alp[0][: (w // 4) * (rows // 4) * 8].cpu().numpy().tobytes(), afmt, w, rows)
    print("%-26s %8.2f us per frame  %7.0f GB/s algorithmic (%.3f of 8000)  same=%s" % (name, us, alg / (us * 1e-6) / 1e9, alg / (us * 1e-6) / 1e9 / 8000, bool(np.array_equal(got, want))))
    del rgba, tex, alp, out
    torch.cuda.empty_cache()
