"""Offline look at a frame written with the fragment table version 2: elements per half-tile (the parse lanes'
trip counts), element kinds and sizes.  python tools/analyze_field_stream.py gpurun_out/c4_frame0.hap"""
import sys, collections
import numpy as np
data = open(sys.argv[1], "rb").read()
def sec(at):
    ln = int.from_bytes(data[at:at+3], "little"); t = data[at+3]; h = 4
    if ln == 0:
        ln = int.from_bytes(data[at+4:at+8], "little"); h = 8
    return h, ln, t
h, ln, t = sec(0); assert t >> 4 == 0xC, hex(t)
at = h
h2, iln, t2 = sec(at); assert t2 == 1
p = at + h2; end = p + iln
tabs = {}
while p < end:
    hh, l, ty = sec(p); tabs[ty] = (p + hh, l); p += hh + l
payload = end
n = tabs[2][1]
sizes = [int.from_bytes(data[tabs[3][0]+4*i:tabs[3][0]+4*i+4], "little") for i in range(n)]
fo, fl = tabs[0x46]
ver, log2, b2, win = data[fo:fo+4]
N = (fl - 4) // 68
fpc = N // n
fsz = np.frombuffer(data, dtype="<u4", count=N, offset=fo+4)
tsz = np.frombuffer(data, dtype=np.uint8, count=64*N, offset=fo+4+4*N).reshape(N, 64)
print("version", ver, "fragments", N, "per chunk", fpc, "fields", b2 >> 4, "window", win)
counts = []; kinds = collections.Counter(); lens = collections.Counter(); maxper = []
cpos = payload
for c in range(n):
    at = cpos
    # varint
    while data[at] & 0x80: at += 1
    at += 1
    for k in range(fpc):
        f = c * fpc + k
        assert tsz[f].sum() == fsz[f], (f, tsz[f].sum(), fsz[f])
        q = at
        per = []
        for hti in range(64):
            e = q + int(tsz[f][hti]); cnt = 0
            while q < e:
                tag = data[q]; kd = tag & 3
                if kd == 0:
                    l = (tag >> 2) + 1; hd = 1
                    if l == 61: l = data[q+1] + 1; hd = 2
                    q += hd + l
                elif kd == 1: l = 4 + ((tag >> 2) & 7); q += 2
                else: l = (tag >> 2) + 1; q += 3
                kinds[kd] += 1; lens[(kd, l)] += 1; cnt += 1
            assert q == e
            per.append(cnt)
        counts.extend(per); maxper.append(max(per))
        at += int(fsz[f])
    cpos += sizes[c]
    if c >= 3: break
counts = np.array(counts); maxper = np.array(maxper)
print("half-tiles", counts.size, "mean elements", counts.mean(), "p50/p90/p99/max", np.percentile(counts, [50, 90, 99, 100]))
print("per-fragment max: mean", maxper.mean(), "p50/p90/max", np.percentile(maxper, [50, 90, 100]))
print("sum of per-fragment (max) vs sum of means:", maxper.sum(), counts.sum() / 64)
print("kinds", dict(kinds))
top = sorted(lens.items(), key=lambda kv: -kv[1])[:16]
print("top (kind,len):", top)
hist = np.bincount(counts, minlength=33)
print("hist elements/half-tile:", hist.tolist())
