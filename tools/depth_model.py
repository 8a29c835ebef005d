"""Round 6, CPU only: classes of the elements of oracle-made field streams (literal / copy from an earlier step / run = overlapping
copy one block back / single-block and multi-block copies by distance) and the depth of the copy chains inside the decoder's
64-block steps, under Snappy's semantics and with overlapping elements collapsed to their root.  Needs the texture cache that
tools/resolve_stats.py writes (/tmp/resolve_stats_<cfg>.tex).
    python tools/depth_model.py [C4|C5y|C5a|C2|C3] [fragments]"""
import os, sys, ctypes as C, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _libs as L
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nfrag = int(sys.argv[2]) if len(sys.argv) > 2 else 200
layout = {"C4": 4, "C5y": 4, "C5a": 6, "C2": 2, "C3": 4}[cfg]
tex = np.fromfile("/tmp/resolve_stats_%s.tex" % cfg, dtype=np.uint8)
lib = L.oracle_lib(); lib.ofs_compress_fragment.restype = C.c_uint
FO = {4: (0, 2, 8, 12), 2: (0, 4, 8, 12), 6: (0, 2, 8, 10)}[layout]
FS = {4: (2, 6, 4, 4), 2: (4, 4, 4, 4), 6: (2, 6, 2, 6)}[layout]
B = 16 if layout == 4 else 8
FPB = 4 if layout == 4 else 2          # fields per block
STEPB = 64 * B
out = np.zeros(8192 + 512, dtype=np.uint8); gt = np.zeros(196, dtype=np.uint8)
rng = np.random.RandomState(1)
frags = rng.choice(len(tex) // 8192, size=min(nfrag, len(tex) // 8192), replace=False)
# field index by byte position inside a 16-byte unit
pos2k = {FO[k]: k for k in range(4)}
cls = collections.Counter(); clsb = collections.Counter(); depth_hist = collections.Counter(); depthA_hist = collections.Counter()
nel_tot = 0; tot_c = 0
maxdepth_frag = []; maxdepthA_frag = []
for f in frags:
    src = tex[f * 8192:(f + 1) * 8192]
    n = lib.ofs_compress_fragment(src.ctypes.data_as(C.c_void_p), 8192, layout, 0, out.ctypes.data_as(C.c_void_p), gt.ctypes.data_as(C.c_void_p))
    tot_c += n
    s = out[:n].tobytes()
    q = 0; p = 0
    # per field (byte position of field start -> info)
    src_of = {}      # field start pos -> source field start pos (snappy semantics) or None (literal)
    srcA_of = {}     # analytic: runs (off == B, len > B... ) point at root
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
        nel_tot += 1
        step0 = p - (p % STEPB)
        if off == 0:
            c = "lit"
        else:
            same = (p - off) >= step0 or (p + ln - 1 - off) >= step0
            multi = (p // B) != ((p + ln - 1) // B)
            if not same: c = "copy_earlier"
            elif not multi: c = "same_single_d%d" % min(off // B, 5)
            elif off < ln: c = "same_run_d%d" % (off // B)
            else: c = "same_multi_nonoverlap_d%d" % min(off // B, 5)
        cls[c] += 1; clsb[c] += ln
        a = p
        while a < p + ln:
            k = pos2k[a & 15]
            if off == 0:
                src_of[a] = None; srcA_of[a] = None
            else:
                sp = a - off
                src_of[a] = sp if sp >= step0 else None
                # analytic: overlapped element: source = first occurrence of column before element start
                if off < ln:
                    j = (a - p) // off
                    spa = a - off * (j + 1)
                else:
                    spa = sp
                srcA_of[a] = spa if spa >= step0 else None
            a += FS[k]
        p += ln
    assert p == 8192
    def depths(m):
        d = {}
        mx = 0
        for a in sorted(m):
            sp = m[a]
            d[a] = 0 if sp is None else d[sp] + 1
            mx = max(mx, d[a])
        return d, mx
    d, mx = depths(src_of); dA, mxA = depths(srcA_of)
    for v in d.values(): depth_hist[min(v, 9)] += 1
    for v in dA.values(): depthA_hist[min(v, 9)] += 1
    maxdepth_frag.append(mx); maxdepthA_frag.append(mxA)
print(cfg, "ratio %.4f elements/frag %.1f" % (tot_c / (8192 * len(frags)), nel_tot / len(frags)))
tot = sum(cls.values()); totb = sum(clsb.values())
for c in sorted(cls):
    print("  %-28s elements %.3f bytes %.3f  mean len %.1f" % (c, cls[c] / tot, clsb[c] / totb, clsb[c] / cls[c]))
t = sum(depth_hist.values())
print("field depth (snappy):  ", {k: round(v / t, 3) for k, v in sorted(depth_hist.items())}, "max per frag mean %.1f" % np.mean(maxdepth_frag))
print("field depth (analytic):", {k: round(v / t, 3) for k, v in sorted(depthA_hist.items())}, "max per frag mean %.1f max %d" % (np.mean(maxdepthA_frag), max(maxdepthA_frag)))
