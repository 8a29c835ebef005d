"""Offline look at a frame written with the fragment table version 3: elements per group (the parse lanes' trip
counts) and per 128-byte half-tile, element kinds and sizes.  python tools/analyze_field_stream.py gpurun_out/c4_frame0.hap"""
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
N = (fl - 4) // 100
fpc = N // n
fsz = np.frombuffer(data, dtype="<u4", count=N, offset=fo+4)
def groups(f):
    bits = int.from_bytes(data[fo+4+4*N+96*f: fo+4+4*N+96*(f+1)], "little")
    return [(bits >> (12 * g)) & 0xFFF for g in range(64)]
print("version", ver, "fragments", N, "per chunk", fpc, "fields", b2 >> 4, "window", win)
counts = []; kinds = collections.Counter(); lens = collections.Counter(); maxper = []; group_trips = []
cpos = payload
for c in range(n):
    at = cpos
    # varint
    while data[at] & 0x80: at += 1
    at += 1
    for k in range(fpc):
        f = c * fpc + k
        gs = groups(f)
        assert sum(gs) == fsz[f], (f, sum(gs), fsz[f])
        q = at
        per = collections.Counter(); produced = 0; trips = []
        for g in range(64):
            e = q + gs[g]; cnt = 0
            while q < e:
                tag = data[q]; kd = tag & 3
                if kd == 0:
                    l = (tag >> 2) + 1; hd = 1
                    if l == 61: l = data[q+1] + 1; hd = 2
                    q += hd + l
                elif kd == 1: l = 4 + ((tag >> 2) & 7); q += 2
                else: l = (tag >> 2) + 1; q += 3
                kinds[kd] += 1; lens[(kd, l)] += 1; cnt += 1
                per[produced >> 7] += 1; produced += l
            assert q == e
            trips.append(cnt)
        counts.extend(per[h] for h in range((produced + 127) >> 7)); maxper.append(max(per.values())); group_trips.append(max(trips))
        at += int(fsz[f])
    cpos += sizes[c]
    if c >= 3: break
counts = np.array(counts); maxper = np.array(maxper)
print("half-tiles", counts.size, "mean elements", counts.mean(), "p50/p90/p99/max", np.percentile(counts, [50, 90, 99, 100]))
print("busiest half-tile per fragment: mean", maxper.mean(), "p50/p90/max", np.percentile(maxper, [50, 90, 100]))
print("elements per group (the decoder's trip count) per fragment: mean", np.mean(group_trips), "max", max(group_trips))
print("sum of per-fragment (max) vs sum of means:", maxper.sum(), counts.sum() / 64)
print("kinds", dict(kinds))
top = sorted(lens.items(), key=lambda kv: -kv[1])[:16]
print("top (kind,len):", top)
hist = np.bincount(counts, minlength=33)
print("hist elements/half-tile:", hist.tolist())
