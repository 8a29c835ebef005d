"""Dev diagnostic: encode one C4 frame, walk chunk 0's Snappy elements on the CPU and report the first copy whose source
bytes (in the ORIGINAL texture) differ from its destination bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, hap_amd
from hap_amd import synth
w, h, fmt, chunks = 7680, 4320, 0x01, 24
ctx = hap_amd.Context(0)
tb = (w // 4) * (h // 4) * 16
rgba = synth.rgba_frame(w, h, 0, device="cuda")
tex = torch.zeros(tb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
assert ctx.compress_rgba(rgba, w, h, w * 4, fmt, tex) == (0, tb)
cap = hap_amd.HapMaxEncodedLength([tb], [fmt], [chunks])
out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
r, used, res = ctx.encode_frames_rgba([rgba], w, h, w * 4, [fmt], [1], [chunks], [out], flags=1)
fr = out[: used[0]].cpu().numpy().tobytes(); T = tex.cpu().numpy().tobytes()
def sec(b, at):
    n = int.from_bytes(b[at:at+3], 'little'); t = b[at+3]; hh = 4
    if n == 0: n = int.from_bytes(b[at+4:at+8], 'little'); hh = 8
    return hh, n, t
hh, n, t = sec(fr, 0); at = hh; h2, n2, t2 = sec(fr, at); payload = at + h2 + n2
s = fr[payload:]
i = 0
while s[i] & 0x80: i += 1
i += 1
op = 0; bad = 0
while op < 1382400 and bad < 5:
    tag = s[i]; k = tag & 3; start = i
    if k == 0:
        ln = (tag >> 2) + 1
        if ln > 60:
            e = ln - 60; ln = int.from_bytes(s[i+1:i+1+e], 'little') + 1; i += 1 + e
        else: i += 1
        if s[i:i+ln] != T[op:op+ln]:
            print("LITERAL mismatch at out", op, "len", ln); bad += 1
        i += ln; op += ln
    else:
        if k == 1: ln = 4 + ((tag >> 2) & 7); off = ((tag >> 5) << 8) | s[i+1]; i += 2
        elif k == 2: ln = (tag >> 2) + 1; off = int.from_bytes(s[i+1:i+3], 'little'); i += 3
        else: ln = (tag >> 2) + 1; off = int.from_bytes(s[i+1:i+5], 'little'); i += 5
        src = bytes(T[op - off + (j % off if off < ln else j)] for j in range(ln)) if off < ln else T[op-off:op-off+ln]
        if src != T[op:op+ln]:
            print("COPY mismatch at out %d (frag %d, block %d, byte-in-block %d) len %d off %d (blocks %g): src %s dst %s" % (
                op, op // 8192, (op % 8192) // 16, op % 16, ln, off, off / 16, src.hex(), T[op:op+ln].hex()))
            bad += 1
        op += ln
print("walked to", op, "bad", bad)
