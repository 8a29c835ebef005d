"""CPU study for VERDICT r04 item 4 ("frames from other encoders: decide it with data"): how deep are the copy
dependencies of libsnappy-made Hap chunks when every 64 KiB block is cut at 4 KiB windows of compressed bytes (the
records the block scan already makes), literals are placed first and copies are resolved in rounds across windows?

Model: one wavefront per window.  Inside its window a wavefront knows every element's output position (prefix sums of the
parse) and may run elements in any order; a copy whose source bytes were produced by ANOTHER window can only run in a
round after the one that produced them.  depth(byte) = 0 for literal bytes; for a copied byte depth(source) + 1 if the
source lies in another window, depth(source) otherwise.  A block needs max depth + 1 rounds; a window is finished in the
round of its deepest byte.

    python tools/foreign_depth.py [C4|C5y|C5a|C3|C2]
No GPU.  Needs libsnappy (tests/_libs.snappy_lib) -- the streams are the reference's, not this library's.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _data as D
import _libs as L

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
w, h, fmt, chunk_bytes = {"C4": (7680, 720, L.FMT_YCOCG, 7680 * 4320 // 24), "C5y": (16384, 256, L.FMT_YCOCG, 16384 * 16384 // 64),
                          "C5a": (16384, 512, L.FMT_RGTC1, 16384 * 16384 // 2 // 64), "C3": (3840, 2160, L.FMT_DXT5, 3840 * 2160 // 8),
                          "C2": (3840, 2160, L.FMT_DXT1, 3840 * 2160 // 2)}[cfg]
from hap_amd import synth
img = synth.rgba_frame(w, h, 0, device="cpu").numpy()
tex = np.frombuffer(D.oracle_bc_encode(img, fmt), dtype=np.uint8)
chunk = np.ascontiguousarray(tex[: min(len(tex), chunk_bytes)])
snappy = L.snappy_lib()
if snappy is None:
    raise SystemExit("libsnappy not found")
cap = 32 + len(chunk) + len(chunk) // 6
out = np.zeros(cap, dtype=np.uint8)
n_out = C.c_size_t(cap)
assert snappy.snappy_compress(chunk.ctypes.data_as(C.c_char_p), C.c_size_t(len(chunk)), out.ctypes.data_as(C.c_char_p), C.byref(n_out)) == 0
s = out[: n_out.value].tobytes()
print("%s: chunk of %d bytes -> %d (ratio %.4f)" % (cfg, len(chunk), len(s), len(s) / len(chunk)))
# varint
q = 0
while s[q] & 0x80:
    q += 1
q += 1
WINDOW = 4096
total = len(chunk)
depth = np.zeros(total, dtype=np.int16)
wid = np.zeros(total, dtype=np.int32)
p = 0
elements = lits = 0
cross = 0
while q < len(s):
    tag = s[q]
    kind = tag & 3
    win = q // WINDOW
    if kind == 0:
        ln = (tag >> 2) + 1
        hd = 1
        if ln > 60:
            ex = ln - 60
            ln = int.from_bytes(s[q + 1: q + 1 + ex], "little") + 1
            hd = 1 + ex
        depth[p: p + ln] = 0
        wid[p: p + ln] = win
        q += hd + ln
        lits += 1
    else:
        if kind == 1:
            ln = 4 + ((tag >> 2) & 7)
            off = ((tag >> 5) << 8) | s[q + 1]
            q += 2
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = s[q + 1] | (s[q + 2] << 8)
            q += 3
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(s[q + 1: q + 5], "little")
            q += 5
        src = p - off
        if off >= ln:
            d = depth[src: src + ln] + (wid[src: src + ln] != win)
            depth[p: p + ln] = d
        else:                                   # overlapping: the first `off` bytes repeat
            d = depth[src: src + off] + (wid[src: src + off] != win)
            reps = -(-ln // off)
            depth[p: p + ln] = np.tile(d, reps)[:ln]
        cross += int((wid[src: src + min(ln, off)] != win).any())
        wid[p: p + ln] = win
    p += ln
    elements += 1
assert p == total
print("elements %d (%.2f bytes each), literals %d, copies with a source in another window %d (%.1f %%)" % (
    elements, total / elements, lits, cross, 100.0 * cross / max(1, elements - lits)))
hist = np.bincount(depth)
cum = np.cumsum(hist) / total
print("bytes by dependency depth (rounds after the literals):")
for d in range(min(len(hist), 40)):
    print("  depth %2d: %6.2f %%   cumulative %6.2f %%" % (d, 100.0 * hist[d] / total, 100.0 * cum[d]))
print("max depth %d; bytes resolved within 8 rounds: %.2f %%" % (int(depth.max()), 100.0 * cum[min(8, len(cum) - 1)]))
# per 64 KiB block: rounds needed; per window: its deepest byte
blocks = [int(depth[b: b + 65536].max()) for b in range(0, total, 65536)]
print("rounds per 64 KiB block (max depth + 1): mean %.1f, median %d, max %d" % (np.mean(blocks) + 1, int(np.median(blocks)) + 1, max(blocks) + 1))
wmax = {}
for win in np.unique(wid):
    wmax[int(win)] = int(depth[wid == win].max())
vals = np.array(list(wmax.values()))
print("windows: %d; finished by round 1 / 2 / 4 / 8: %.1f / %.1f / %.1f / %.1f %%; mean last round %.1f" % (
    len(vals), 100.0 * (vals <= 0).mean(), 100.0 * (vals <= 1).mean(), 100.0 * (vals <= 3).mean(), 100.0 * (vals <= 7).mean(), vals.mean() + 1))
