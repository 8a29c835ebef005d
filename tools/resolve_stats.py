"""CPU model of the field-stream decoder's chain resolution (snappy_decode_fields.hip) on textures made by the oracle:
how many field columns have sources inside their own 64-block step, what the DPP hops leave pending, how many
pointer-doubling rounds a fragment needs, elements per fragment.  No GPU.
    python tools/resolve_stats.py [C4|C5a|C5y] [fragments]
"""
import os, sys, ctypes as C, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _libs as L
import _data as D

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
nfrag = int(sys.argv[2]) if len(sys.argv) > 2 else 600
w, h, fmt, layout = {"C4": (7680, 1024, L.FMT_YCOCG, 4), "C5y": (16384, 512, L.FMT_YCOCG, 4),
                     "C5a": (16384, 512, L.FMT_RGTC1, 6), "C2": (3840, 2160, L.FMT_DXT1, 2), "C3": (3840, 2160, L.FMT_DXT5, 4)}[cfg]
cache = "/tmp/resolve_stats_%s.tex" % cfg
if os.path.exists(cache):
    tex = np.fromfile(cache, dtype=np.uint8)
else:
    # (a band of the frame: the synthetic content is position-hashed, any band has the same statistics)
    from hap_amd import synth
    img = synth.rgba_frame(w, h, 0, device="cpu").numpy()
    tex = np.frombuffer(D.oracle_bc_encode(img, fmt), dtype=np.uint8).copy()
    tex.tofile(cache)
lib = L.oracle_lib()
lib.ofs_compress_fragment.restype = C.c_uint
FO = {4: (0, 2, 8, 12), 2: (0, 4, 8, 12), 6: (0, 2, 8, 10)}[layout]
BLOCK = 16 if layout == 4 else 8
UNITB = 16
pos2field = {}
for u in range(8):
    for k in range(4):
        pos2field[u * 16 + FO[k]] = u * 4 + k
out = np.zeros(8192 + 512, dtype=np.uint8); gt = np.zeros(196, dtype=np.uint8)
rng = np.random.RandomState(1)
frags = rng.choice(len(tex) // 8192, size=min(nfrag, len(tex) // 8192), replace=False)
st = collections.Counter(); rounds_hist = collections.Counter(); rounds_nohop = collections.Counter()
pend_by_k = collections.Counter(); tot_by_k = collections.Counter(); el_hist = []
total_c = 0
for f in frags:
    src = tex[f * 8192:(f + 1) * 8192]
    n = lib.ofs_compress_fragment(src.ctypes.data_as(C.c_void_p), 8192, layout, 0, out.ctypes.data_as(C.c_void_p), gt.ctypes.data_as(C.c_void_p))
    total_c += n
    s = out[:n].tobytes()
    # per field: distance in bytes (0 = literal)
    dist = np.zeros(2048, dtype=np.int32)
    q = 0; p = 0; nel = 0
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
        # fields covered by [p, p + ln)
        a = p
        while a < p + ln:
            hp = a & 127
            fi = (a >> 7) * 32 + pos2field[hp]
            dist[fi] = off
            k = fi & 3
            a += (FO[k + 1] if k < 3 else 16) - FO[k]
        p += ln; nel += 1
    assert p == 8192
    el_hist.append(nel)
    # decoder view: lane = 16-byte unit of the step (layout 4) -- for 8-byte layouts model the planned 16-byte-unit decoder:
    # distances that are not whole units cannot be expressed there; count them
    d = dist.reshape(8, 64, 4)              # [step, lane, field]
    if BLOCK == 8:
        st["odd_distance_fields"] += int(((d % 16) != 0).sum())
    du = d // UNITB                         # distance in units (floor)
    lanes = np.arange(64).reshape(1, 64, 1)
    copy = d > 0
    instep = copy & (du <= lanes) & ((d % 16) == 0)
    srcl = np.where(instep, lanes - du, -1)             # source lane or -1
    for k in range(4):
        pend_by_k[k] += int(instep[:, :, k].sum()); tot_by_k[k] += 512
    st["fields"] += 2048; st["copy_fields"] += int(copy.sum()); st["instep_fields"] += int(instep.sum())
    cols = instep.any(axis=1)             # [step, field]
    st["cols"] += 32; st["cols_pending"] += int(cols.sum())
    # pointer state: ptr[l] = lane it waits for (or l itself when resolved)
    def hops(ptr):
        ptr = ptr.copy()
        row = np.arange(64) & 15
        for hh in (1, 2, 4, 8):
            lanes1 = np.arange(64)
            want = lanes1 - hh
            take = (ptr == want) & (row >= hh) & (ptr != lanes1)
            src = np.where(take, lanes1 - hh, lanes1)
            newp = ptr[src]
            # taking a resolved source (ptr[src] == src) means: my root is src
            ptr = np.where(take, np.where(newp == src, src, newp), ptr)
            # a lane whose pointer now names a resolved lane is resolved in the descriptor sense (it holds the root's value)
        return ptr
    def nrounds(ptr, res):
        # res[l]: lane l holds a final descriptor.  ptr[l] for unresolved: lane to fetch from
        r = 0
        while not res.all():
            nres = res | res[ptr]
            nptr = np.where(res, ptr, ptr[ptr])
            # descriptor semantics: after a fetch, lane holds the fetched lane's descriptor: resolved if that was resolved
            res, ptr = nres, np.where(nres, np.arange(64), nptr)
            r += 1
        return r
    maxr = 0; maxr0 = 0; colp_after = 0
    for s_ in range(8):
        for k in range(4):
            sl = srcl[s_, :, k]
            lanes1 = np.arange(64)
            res0 = sl < 0
            ptr0 = np.where(res0, lanes1, sl)
            maxr0 = max(maxr0, nrounds(ptr0, res0))
            # hops, in descriptor semantics: simulate exactly as the kernel (value copied from the lane below when I point at it)
            ptr = ptr0.copy(); res = res0.copy(); row = lanes1 & 15
            for hh in (1, 2, 4, 8):
                take = (~res) & (ptr == lanes1 - hh) & (row >= hh)
                srcs = np.where(take, lanes1 - hh, lanes1)
                nptr = np.where(take, ptr[srcs], ptr); nres = np.where(take, res[srcs], res)
                # (a resolved source's descriptor names itself: the taker becomes resolved)
                ptr, res = np.where(nres, lanes1, nptr), nres
            if not res.all():
                colp_after += 1
            maxr = max(maxr, nrounds(ptr, res))
    st["cols_pending_after_hops"] += colp_after
    rounds_hist[maxr] += 1; rounds_nohop[maxr0] += 1
nf = len(frags)
print(cfg, "fragments", nf, "ratio(elements only) %.4f" % (total_c / (nf * 8192.0)))
el = np.array(el_hist); print("elements per fragment: mean %.0f  p50 %d p90 %d max %d;  trips ceil(N/64): mean %.2f" % (el.mean(), np.percentile(el, 50), np.percentile(el, 90), el.max(), np.ceil(el / 64).mean()))
print("fields: copy %.3f, in-step copy %.3f" % (st["copy_fields"] / st["fields"], st["instep_fields"] / st["fields"]))
print("in-step pending share by field k:", {k: round(pend_by_k[k] / tot_by_k[k], 3) for k in range(4)})
print("columns with any in-step source: %.3f; still pending after the hops: %.3f" % (st["cols_pending"] / st["cols"], st["cols_pending_after_hops"] / st["cols"]))
print("rounds per fragment with hops:", sorted(rounds_hist.items()), "mean %.2f" % (sum(k * v for k, v in rounds_hist.items()) / nf))
print("rounds per fragment without hops:", sorted(rounds_nohop.items()), "mean %.2f" % (sum(k * v for k, v in rounds_nohop.items()) / nf))
if BLOCK == 8:
    print("fields at odd block distances: %.4f of all fields" % (st["odd_distance_fields"] / st["fields"]))

# ---- what-if: elements may only start and end on 8-byte positions (items = halves of a unit) ----
def cost_stream(kinds, dists, sizes):
    """kinds/dists per item in order, sizes per item (bytes); elements = runs of equal (kind, dist) inside a half-tile,
    copies at most 64 bytes.  Returns stream bytes."""
    total = 0; i = 0; n = len(kinds); pos = 0
    while i < n:
        j = i; ln = 0
        half = pos >> 7
        while j < n and kinds[j] == kinds[i] and dists[j] == dists[i] and ((pos + ln) >> 7) == half and (kinds[i] == 0 or ln + sizes[j] <= 64):
            ln += sizes[j]; j += 1
        if kinds[i] == 0:
            total += (1 if ln <= 60 else 2) + ln
        else:
            total += 2 if (ln < 12 and dists[i] < 2048) else 3
        pos += ln; i = j
    return total

tot_f = tot_h = 0
for f in frags[:200]:
    src = tex[f * 8192:(f + 1) * 8192]
    n = lib.ofs_compress_fragment(src.ctypes.data_as(C.c_void_p), 8192, layout, 0, out.ctypes.data_as(C.c_void_p), gt.ctypes.data_as(C.c_void_p))
    s = out[:n].tobytes()
    dist = np.zeros(2048, dtype=np.int32)
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
        a = p
        while a < p + ln:
            fi = (a >> 7) * 32 + pos2field[a & 127]
            dist[fi] = off
            k = fi & 3
            a += (FO[k + 1] if k < 3 else 16) - FO[k]
        p += ln
    fs = [((FO[k + 1] if k < 3 else 16) - FO[k]) for k in range(4)]
    kinds = [1 if d else 0 for d in dist]; sizes = [fs[i & 3] for i in range(2048)]
    tot_f += cost_stream(kinds, list(dist), sizes)
    d2 = dist.reshape(1024, 2)
    same = (d2[:, 0] == d2[:, 1]) & (d2[:, 0] > 0)
    hk = [1 if x else 0 for x in same]; hd_ = [int(d2[i, 0]) if same[i] else 0 for i in range(1024)]
    tot_h += cost_stream(hk, hd_, [8] * 1024)
print("what-if 8-byte items: re-costed field stream %.4f -> halves %.4f (of the texture; elements only)" % (tot_f / (200 * 8192.0), tot_h / (200 * 8192.0)))
