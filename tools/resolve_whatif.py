"""What-if study on top of tools/resolve_stats.py's model: rounds needed under different DPP hop sets and with copies
re-pointed to the farthest equivalent distance (4 or 2 blocks) when every field of the element also matches there.
    python tools/resolve_whatif.py [C4|C5y] [fragments]"""
import os, sys, ctypes as C, collections, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _libs as L
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nfrag = int(sys.argv[2]) if len(sys.argv) > 2 else 200
layout = 4
tex = np.fromfile("/tmp/resolve_stats_%s.tex" % cfg, dtype=np.uint8)
lib = L.oracle_lib(); lib.ofs_compress_fragment.restype = C.c_uint
FO = (0, 2, 8, 12); FS = (2, 6, 4, 4)
pos2field = {u * 16 + FO[k]: u * 4 + k for u in range(8) for k in range(4)}
out = np.zeros(8192 + 512, dtype=np.uint8); gt = np.zeros(196, dtype=np.uint8)
rng = np.random.RandomState(1)
frags = rng.choice(len(tex) // 8192, size=min(nfrag, len(tex) // 8192), replace=False)
lanes1 = np.arange(64); row = lanes1 & 15

def rounds_after(srcl, hopset):
    res = srcl < 0
    ptr = np.where(res, lanes1, srcl)
    for hh in hopset:
        take = (~res) & (ptr == lanes1 - hh) & (row >= hh)
        srcs = np.where(take, lanes1 - hh, lanes1)
        nptr = np.where(take, ptr[srcs], ptr); nres = np.where(take, res[srcs], res)
        ptr, res = np.where(nres, lanes1, nptr), nres
    r = 0
    while not res.all():
        nres = res | res[ptr]
        nptr = np.where(res, ptr, ptr[ptr])
        res, ptr = nres, np.where(nres, lanes1, nptr)
        r += 1
    COLROUNDS[0] += r
    return r
COLROUNDS = [0]

hopsets = [(1, 2, 4, 8), (1, 2, 4), (4, 8), (1, 4), (4,), (2, 4, 8), ()]
acc = {(pol, hs): [] for pol in ("asis", "far") for hs in hopsets}
upgraded = tot_copy_el = 0
colr = {}
for f in frags:
    src = tex[f * 8192:(f + 1) * 8192]
    n = lib.ofs_compress_fragment(src.ctypes.data_as(C.c_void_p), 8192, layout, 0, out.ctypes.data_as(C.c_void_p), gt.ctypes.data_as(C.c_void_p))
    s = out[:n].tobytes()
    blocks = src.reshape(512, 16)
    def feq(fi, d):       # field fi equals the same field d blocks back
        b = fi >> 2; k = fi & 3
        if b < d: return False
        return bytes(blocks[b, FO[k]:FO[k] + FS[k]]) == bytes(blocks[b - d, FO[k]:FO[k] + FS[k]])
    dist = {"asis": np.zeros(2048, dtype=np.int32), "far": np.zeros(2048, dtype=np.int32)}
    q = 0; p = 0
    while q < n:
        tag = s[q]; kd = tag & 3
        if kd == 0:
            ln = (tag >> 2) + 1; hd = 1
            if ln == 61: ln = s[q + 1] + 1; hd = 2
            off = 0; q += hd + ln
        elif kd == 1:
            ln = 4 + ((tag >> 2) & 7); off = ((tag >> 5) << 8) | s[q + 1]; q += 2
        else:
            ln = (tag >> 2) + 1; off = s[q + 1] | (s[q + 2] << 8); q += 3
        fl = []
        a = p
        while a < p + ln:
            fi = (a >> 7) * 32 + pos2field[a & 127]
            fl.append(fi); k = fi & 3
            a += FS[k]
        far = off
        if off in (16, 32):
            tot_copy_el += 1
            for d in (4, 2):
                if d * 16 > off and all(feq(fi, d) for fi in fl):
                    far = d * 16; upgraded += 1
                    break
        elif off:
            tot_copy_el += 1
        for fi in fl:
            dist["asis"][fi] = off; dist["far"][fi] = far
        p += ln
    for pol in ("asis", "far"):
        d = dist[pol].reshape(8, 64, 4); du = d // 16
        instep = (d > 0) & (du <= lanes1.reshape(1, 64, 1))
        srcl = np.where(instep, lanes1.reshape(1, 64, 1) - du, -1)
        for hs in hopsets:
            COLROUNDS[0] = 0
            acc[(pol, hs)].append(max(rounds_after(srcl[s_, :, k], hs) for s_ in range(8) for k in range(4)))
            colr.setdefault((pol, hs), []).append(COLROUNDS[0])
print(cfg, "fragments", len(frags), "copy elements re-pointed: %.3f" % (upgraded / max(1, tot_copy_el)))
for pol in ("asis", "far"):
    for hs in hopsets:
        a = np.array(acc[(pol, hs)])
        valu = 3 * len(hs) * 32 + (a.mean() + 1) * 19
        ldsc = (a.mean() + 1) * 32 * 6.1
        print("%-5s hops %-14s rounds mean %.2f max %d   -> resolve VALU %.0f, LDS cycles %.0f; column-rounds needed %.1f of %.0f" % (pol, hs, a.mean(), a.max(), valu, ldsc, np.mean(colr[(pol, hs)]), 32 * (a.mean() + 1)))
