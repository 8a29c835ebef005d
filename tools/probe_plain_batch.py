"""GPU probe (r06): a batch of plain hap.h frames of this library (no private table) through the decoder -- block scan, group tables
from the scan's records, block-per-lane kernel -- against the generic kernel alone; kernel classes by HIP events.
    python tools/probe_plain_batch.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
from hap_amd import synth
w, h, fmts, chunks = 7680, 4320, [0x01], [24]
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = hap_amd.Context(0)
nb = (w // 4) * (h // 4) * 16
cap = hap_amd.HapMaxEncodedLength([nb], fmts, chunks)
rgba = [synth.rgba_frame(w, h, i, device="cuda") for i in range(nf)]
frames = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
torch.cuda.synchronize()
r, used, res = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, [1], chunks, frames, flags=0)
assert r == 0
want = torch.empty(nb, dtype=torch.uint8, device="cuda")
dec = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(nf)]
torch.cuda.synchronize()
for flags, name in ((0, "default"), (hap_amd.DECODE_NO_FIELD_GUESS, "generic pieces")):
    n0 = ctx.table_fallbacks()
    ctx.decode_frames(frames, used, 0, dec, flags)
    best = None
    for _ in range(4):
        ctx.timer_start(); ctx.decode_frames(frames, used, 0, dec, flags); ms = ctx.timer_stop()
        best = ms if best is None else min(best, ms)
    ctx.set_profiling(True); ctx.collect_profile()
    ctx.decode_frames(frames, used, 0, dec, flags)
    prof = ctx.collect_profile(); ctx.set_profiling(False)
    ok = True
    for i in range(0, nf, max(1, nf // 4)):
        ctx.compress_rgba(rgba[i], w, h, w * 4, fmts[0], want)
        ok = ok and bool(torch.equal(dec[i], want))
    print("%2d frames %-16s call %.3f ms same=%s fallbacks %d  %s" % (nf, name, best, ok, ctx.table_fallbacks() - n0,
          " ".join("%s %.3f" % (k, v[1]) for k, v in prof.items() if v[0])))
