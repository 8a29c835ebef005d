"""Prints digests of encoded frames (deterministic encoder): used to confirm that a kernel rewrite that is
meant to be output-neutral really leaves every byte of the compressed streams unchanged."""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import hap_amd
from hap_amd import synth
import _data as D

ctx = hap_amd.Context(0)
h = hashlib.sha256()
for (w, hgt, fmts, chunks) in [(1920, 1080, [0x83F0], [1]), (1920, 1080, [0x83F3], [8]), (3840, 2160, [0x01], [24]),
                               (2048, 2048, [0x01, 0x8DBB], [16, 16])]:
    for lg in (10, 13, 16):
        ctx.set_fragment_log2(lg)
        for fr in range(2):
            img = synth.rgba_frame(w, hgt, fr, device="cuda")
            tb = [(w // 4) * (hgt // 4) * (8 if f in (0x83F0, 0x8DBB) else 16) for f in fmts]
            out = torch.zeros(hap_amd.HapMaxEncodedLength(tb, fmts, chunks) + 65536, dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            r, used, res = ctx.encode_frames_rgba([img], w, hgt, w * 4, fmts, [1] * len(fmts), chunks, [out], flags=1)
            assert r == 0
            b = out[: used[0]].cpu().numpy().tobytes()
            d = hashlib.sha256(b).hexdigest()[:16]
            h.update(b)
            print(w, hgt, [hex(f) for f in fmts], "F=2^%d" % lg, "frame", fr, used[0], d)
# byte streams with odd sizes (byte-granular path) and raw textures
ctx.set_fragment_log2(13)
for n, kind, fmt, ch in [(100001, "mixed", 0x83F3, 3), (65536 * 3 + 2, "runs", 0x8E8C, 1), (16 * 7919, "mixed", 0x8DBB, 7), (999, "zero", 0x83F0, 2)]:
    tex = D.stream_bytes(n, kind, seed=n)
    out = np.zeros(hap_amd.HapMaxEncodedLength([n], [fmt], [ch]) + 65536, dtype=np.uint8)
    r, used, res = ctx.encode_frames([[tex]], [fmt], [1], [ch], [out], flags=1)
    assert r == 0
    b = out[: used[0]].tobytes(); h.update(b)
    print("bytes", n, kind, hex(fmt), ch, used[0], hashlib.sha256(b).hexdigest()[:16])
print("TOTAL", h.hexdigest())
