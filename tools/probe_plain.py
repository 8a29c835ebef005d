"""GPU probe: table-less frames of this library (what plain hap.h HapEncode writes) through the decoder: block scan with
8 KiB marks + generic kernel; kernel classes by HIP events.   python tools/probe_plain.py [C4|C1|C5] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
from hap_amd import synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 1
w, h, fmts, chunks = {"C4": (7680, 4320, [0x01], [24]), "C1": (1920, 1080, [0x83F0], [1]), "C5": (16384, 16384, [0x01, 0x8DBB], [64, 64])}[cfg]
bb = {0x01: 16, 0x83F3: 16, 0x8DBB: 8, 0x83F0: 8}
ctx = hap_amd.Context(0)
sizes = [(w // 4) * (h // 4) * bb[f] for f in fmts]
cap = hap_amd.HapMaxEncodedLength(sizes, fmts, chunks)
rgba = [synth.rgba_frame(w, h, i, device="cuda") for i in range(nf)]
frames = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
torch.cuda.synchronize()
for flags, name in ((0, "plain"), (hap_amd.ENCODE_FRAGMENT_INDEX, "with table")):
    r, used, res = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, [1] * len(fmts), chunks, frames, flags=flags)
    assert r == 0
    for t in range(len(fmts)):
        want = [torch.empty(sizes[t], dtype=torch.uint8, device="cuda") for _ in range(nf)]
        for i in range(nf):
            ctx.compress_rgba(rgba[i], w, h, w * 4, fmts[t], want[i])
        dec = [torch.zeros(sizes[t], dtype=torch.uint8, device="cuda") for _ in range(nf)]
        torch.cuda.synchronize()
        ctx.decode_frames(frames, used, t, dec)
        best = None
        for _ in range(5):
            ctx.timer_start(); ctx.decode_frames(frames, used, t, dec); ms = ctx.timer_stop()
            best = ms if best is None else min(best, ms)
        ctx.set_profiling(True); ctx.collect_profile()
        ctx.decode_frames(frames, used, t, dec)
        prof = ctx.collect_profile(); ctx.set_profiling(False)
        ok = all(bool(torch.equal(dec[i], want[i])) for i in range(nf))
        print("%-10s tex%d %d frame(s): decode call %.3f ms  same=%s  kernels: %s" % (
            name, t, nf, best, ok, " ".join("%s %.3f" % (k, v[1]) for k, v in prof.items() if v[0])))
