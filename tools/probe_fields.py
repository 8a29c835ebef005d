"""GPU probe for the field-stream path (fragment table version 3): parity of the new decoder against the block
encoder's output and the CPU checker, sizes with / without half-tile splitting, kernel times of both decoders.
    python tools/probe_fields.py [C4|C3|C5] [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import hap_amd
from hap_amd import synth
import _libs as L

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w, h, fmts, chunks = {"C4": (7680, 4320, [0x01], [24]), "C3": (3840, 2160, [0x83F3], [8]),
                      "C5": (16384, 16384, [0x01, 0x8DBB], [64, 64]), "S": (512, 256, [0x01], [4]),
                      "C2": (3840, 2160, [0x83F0], [1]), "A8": (7680, 4320, [0x8DBB], [24]), "A2": (2048, 512, [0x8DBB], [4]), "S1": (512, 256, [0x83F0], [2])}[cfg]
bb = {0x01: 16, 0x83F3: 16, 0x8DBB: 8, 0x83F0: 8}
ctx = hap_amd.Context(0)
sizes = [(w // 4) * (h // 4) * bb[f] for f in fmts]
cap = hap_amd.HapMaxEncodedLength(sizes, fmts, chunks)
rgba = [synth.rgba_frame(w, h, i, device="cuda") for i in range(nf)]
frames = [torch.zeros(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
tex = [[torch.zeros(s, dtype=torch.uint8, device="cuda") for _ in range(nf)] for s in sizes]
torch.cuda.synchronize()
for t, f in enumerate(fmts):
    for i in range(nf):
        assert ctx.compress_rgba(rgba[i], w, h, w * 4, f, tex[t][i]) == (0, sizes[t])
ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, [1] * len(fmts), chunks, frames, flags=hap_amd.ENCODE_FRAGMENT_INDEX)
ctx.set_profiling(True); ctx.collect_profile()
r, used, res = ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, [1] * len(fmts), chunks, frames, flags=hap_amd.ENCODE_FRAGMENT_INDEX)
prof = ctx.collect_profile(); ctx.set_profiling(False)
print("encode", r, res[:2], "ratio %.4f" % (sum(used) / nf / sum(sizes)), "compress_ms %.3f bc_ms %.3f pack_ms %.3f gather_ms %.3f" % (
    prof.get("snappy_compress", (0, 0))[1], prof.get("block_encode", (0, 0))[1], prof.get("frame_pack", (0, 0))[1], prof.get("frame_gather", (0, 0))[1]))
ok = True
for flags, name in ((0, "fields"), (hap_amd.DECODE_IGNORE_HALF_TILES, "generic-fragments"), (hap_amd.DECODE_IGNORE_FRAGMENT_INDEX, "streams")):
    for t in range(len(fmts)):
        dec = [torch.zeros(sizes[t], dtype=torch.uint8, device="cuda") for _ in range(nf)]
        torch.cuda.synchronize()
        ctx.decode_frames(frames, used, t, dec, flags=flags)        # warm-up (module load, scratch growth)
        ctx.set_profiling(True); ctx.collect_profile()
        r, dused, dfm, dres = ctx.decode_frames(frames, used, t, dec, flags=flags)
        prof = ctx.collect_profile(); ctx.set_profiling(False)
        same = [bool(torch.equal(dec[i], tex[t][i])) for i in range(nf)]
        ms = prof.get("snappy_decode", (0, 0.0))[1]
        print("%-18s tex%d r=%d res=%s same=%s decode_ms=%.3f (%.0f GB/s out)" % (name, t, r, dres[:2], same[:4], ms,
              nf * sizes[t] / (ms * 1e-3) / 1e9 if ms else 0))
        if not all(same) and flags == 0:
            ok = False
            a = dec[0].cpu().numpy(); b = tex[t][0].cpu().numpy()
            bad = np.nonzero(a != b)[0]
            print("  first mismatches at", bad[:8], "count", bad.size, "frag", bad[0] // 8192 if bad.size else None)
            if bad.size:
                p = int(bad[0]) & ~15
                print("   got ", a[p:p + 32].tobytes().hex())
                print("   want", b[p:p + 32].tobytes().hex())
# the CPU checker decodes the frame with the version-3 table unchanged
frame = frames[0][: used[0]].cpu().numpy()
api = L.ref_api() or L.oracle_api()
for t in range(len(fmts)):
    rc, out, fmt = api.decode_np(frame, t, sizes[t])
    print("checker decode tex%d" % t, rc, bool(rc == 0 and np.array_equal(out, tex[t][0].cpu().numpy())))
print("PARITY", ok)
