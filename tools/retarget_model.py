"""Round 6, CPU only: re-point the copy offsets of oracle-made field streams by candidate rules (powers of
two of the base distance, or all the way to the head of the chain), check that the stream still decodes to the same bytes, and
count the pointer-doubling rounds a decoder WITHOUT hops would need -- under Snappy's semantics (S) and with overlapping elements
collapsed to their root at parse time (A).  The answer (LABNOTES.md, Part R6): 3.5 rounds at best, one offset per element cannot
serve field columns whose chains have different heads.  Needs /tmp/resolve_stats_<cfg>.tex (tools/resolve_stats.py).
    python tools/retarget_model.py [C4|C5y|C5a] [fragments]"""
import os, sys, ctypes as C, collections, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _libs as L
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nfrag = int(sys.argv[2]) if len(sys.argv) > 2 else 150
layout = {"C4": 4, "C5y": 4, "C5a": 6, "C2": 2, "C3": 4}[cfg]
tex = np.fromfile("/tmp/resolve_stats_%s.tex" % cfg, dtype=np.uint8)
lib = L.oracle_lib(); lib.ofs_compress_fragment.restype = C.c_uint
FO = {4: (0, 2, 8, 12), 2: (0, 4, 8, 12), 6: (0, 2, 8, 10)}[layout]
FS = {4: (2, 6, 4, 4), 2: (4, 4, 4, 4), 6: (2, 6, 2, 6)}[layout]
B = 16 if layout == 4 else 8
STEPB = 64 * B
out = np.zeros(8192 + 512, dtype=np.uint8); gt = np.zeros(196, dtype=np.uint8)
rng = np.random.RandomState(1)
frags = rng.choice(len(tex) // 8192, size=min(nfrag, len(tex) // 8192), replace=False)
pos2k = {FO[k]: k for k in range(4)}

def parse(s, n):
    q = 0; p = 0; els = []
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
        els.append((p, ln, off)); p += ln
    return els

def fields_of(p, ln):
    a = p; r = []
    while a < p + ln:
        k = pos2k[a & 15]; r.append((a, FS[k])); a += FS[k]
    return r

def chain(src, a, sz, dB, cap):
    """number of consecutive equalities of the field at a going back in steps of dB bytes (>= 1 for a copy at base dB)"""
    n = 0; v = src[a:a + sz]
    while n < cap and a - (n + 1) * dB >= 0 and src[a - (n + 1) * dB: a - (n + 1) * dB + sz] == v:
        n += 1
    return n

def retarget(rule, src, p, ln, off):
    if off == 0 or off > 4 * B or off % B:
        return off
    fl = fields_of(p, ln)
    cap = 64
    n = min(chain(src, a, sz, off, cap) for a, sz in fl)
    assert n >= 1, (p, ln, off)
    step0 = p - p % STEPB
    if rule == "head":       # to the chain's head; if the head lies before the step, the nearest multiple whose sources all lie before the step
        m = n
        need = -(-(p + ln - step0) // off)        # multiples so that the last source byte lies before the step
        if need <= n: m = need
    elif rule == "pow2":
        m = 1 << int(math.floor(math.log2(n)))
        need = -(-(p + ln - step0) // off)
        need2 = 1 << int(math.ceil(math.log2(need))) if need > 1 else 1
        if need2 <= n: m = need2
    elif rule == "pow2plain":
        m = 1 << int(math.floor(math.log2(n)))
    else:
        m = 1
    no = off * m
    if ln < 12 and off < 2048 and no >= 2048:       # would change the element's size class: keep within copy-1 reach
        m = 2047 // off; no = off * m
    return no

def depth_stats(els, sem):
    src_of = {}
    for p, ln, off in els:
        step0 = p - p % STEPB
        over = sem == "A" and off == B and (p % B) + ln > B and (p - step0) // B >= 1
        k0pos = p % B
        s_blk = p // B
        for a, sz in fields_of(p, ln):
            if off == 0: src_of[a] = None; continue
            if over:
                # root: block s-1 for columns at or beyond the start's column, block s for the columns before it
                col = a % B
                rb = s_blk - 1 if col >= k0pos else s_blk
                sp = rb * B + col
                if sp == a: sp = a - off
            else:
                sp = a - off
            src_of[a] = sp if sp >= step0 else None
    d = {}; mx = 0
    for a in sorted(src_of):
        sp = src_of[a]
        d[a] = 0 if sp is None else d[sp] + 1
        mx = max(mx, d[a])
    return mx, d

res = {}
for rule in ("none", "pow2plain", "pow2", "head"):
    for sem in ("S", "A"):
        res[(rule, sem)] = collections.Counter()
deep = collections.Counter()
for f in frags:
    src = tex[f * 8192:(f + 1) * 8192].tobytes()
    n = lib.ofs_compress_fragment(tex[f * 8192:].ctypes.data_as(C.c_void_p), 8192, layout, 0, out.ctypes.data_as(C.c_void_p), gt.ctypes.data_as(C.c_void_p))
    els = parse(out[:n].tobytes(), n)
    for rule in ("none", "pow2plain", "pow2", "head"):
        els2 = [(p, ln, retarget(rule, src, p, ln, off)) for p, ln, off in els]
        # check validity: decoding must reproduce src
        buf = bytearray(8192)
        for p, ln, off in els2:
            if off == 0: buf[p:p + ln] = src[p:p + ln]
            else:
                for i in range(ln): buf[p + i] = buf[p + i - off]
        assert bytes(buf) == src, rule
        for sem in ("S", "A"):
            mx, d = depth_stats(els2, sem)
            rounds = 0 if mx == 0 else int(math.floor(math.log2(mx))) + 1
            res[(rule, sem)][rounds] += 1
print(cfg, "fragments", len(frags))
for key in res:
    c = res[key]; t = sum(c.values())
    print("  %-10s %s  rounds: %s  mean %.2f" % (key[0], key[1], dict(sorted(c.items())), sum(k * v for k, v in c.items()) / t))

# diagnose: deepest chain under head/A in a few fragments
print("---- diagnose")
for f in frags[:6]:
    src = tex[f * 8192:(f + 1) * 8192].tobytes()
    n = lib.ofs_compress_fragment(tex[f * 8192:].ctypes.data_as(C.c_void_p), 8192, layout, 0, out.ctypes.data_as(C.c_void_p), gt.ctypes.data_as(C.c_void_p))
    els = parse(out[:n].tobytes(), n)
    els2 = [(p, ln, retarget("head", src, p, ln, off)) for p, ln, off in els]
    orig = {p: off for p, ln, off in els}
    mx, d = depth_stats(els2, "A")
    # element of each field
    owner = {}
    for p, ln, off in els2:
        for a, sz in fields_of(p, ln): owner[a] = (p, ln, off)
    a = max(d, key=lambda x: d[x])
    print("frag", f, "max depth", mx, "at", a, "col", a % B)
    # rebuild chain
    src_of = {}
    for p, ln, off in els2:
        step0 = p - p % STEPB
        over = off == B and (p % B) + ln > B and (p - step0) // B >= 1
        for aa, sz in fields_of(p, ln):
            if off == 0: src_of[aa] = None; continue
            if over:
                col = aa % B; rb = p // B - 1 if col >= p % B else p // B; sp = rb * B + col
                if sp == aa: sp = aa - off
            else: sp = aa - off
            src_of[aa] = sp if sp >= step0 else None
    while a is not None:
        p, ln, off = owner[a]
        print("    field %5d (blk %3d col %2d) element p=%5d len=%2d off=%4d (was %4d) depth %d" % (a, (a % STEPB) // B, a % B, p, ln, off, orig[p], d[a]))
        a = src_of[a]
