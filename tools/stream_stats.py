"""Dev tool: element / window statistics of the fragment streams in a frame written by tools/dump_frame.py
(simulates the decoder's 64-input-byte windows to see how full its 64-unit production steps are)."""
import sys, struct, collections
data = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/frame_c4.bin", "rb").read()
# 8-byte top header, complex section: instruction container
p = 8 if data[0:3] == b"\0\0\0" else 4
assert data[p + 3] == 0x01
ilen = int.from_bytes(data[p:p + 3], "little"); q = p + 4; end = q + ilen
tables = {}
while q < end:
    l = int.from_bytes(data[q:q + 3], "little"); t = data[q + 3]; tables[t] = data[q + 4:q + 4 + l]; q += 4 + l
n = len(tables[2]); sizes = struct.unpack("<%dI" % n, tables[3])
ft = tables[0x46]; frag_log2, gran_log2 = ft[1], ft[2]; fs = struct.unpack("<%dI" % ((len(ft) - 4) // 4), ft[4:])
fpc = len(fs) // n
G = 1 << gran_log2
print("chunks", n, "frag_log2", frag_log2, "gran", G, "fragments", len(fs))
payload = end
kinds = collections.Counter(); lens = collections.Counter(); tot_windows = tot_steps1 = tot_steps2 = tot_out = tot_tokens = longlit = 0
import random
random.seed(1)
pos = payload
frag_i = 0
sample = set(random.sample(range(len(fs)), 300))
for c in range(n):
    cp = pos
    # varint
    v = 0; sh = 0
    while True:
        b = data[cp]; cp += 1; v |= (b & 0x7F) << sh; sh += 7
        if not b & 0x80: break
    for k in range(fpc):
        L = fs[frag_i]
        if frag_i in sample and L:
            s = data[cp:cp + L]
            # parse elements
            i = 0; toks = []
            while i < L:
                tag = s[i]; kd = tag & 3
                if kd == 0:
                    ln = (tag >> 2) + 1; hd = 1
                    if ln > 60:
                        ex = ln - 60; ln = int.from_bytes(s[i + 1:i + 1 + ex], "little") + 1; hd = 1 + ex
                    toks.append((i, hd + ln, ln, 0 if hd == 1 else 9)); i += hd + ln
                elif kd == 1:
                    toks.append((i, 2, 4 + ((tag >> 2) & 7), 1)); i += 2
                elif kd == 2:
                    toks.append((i, 3, (tag >> 2) + 1, 2)); i += 3
                else:
                    toks.append((i, 5, (tag >> 2) + 1, 3)); i += 5
            for t in toks: kinds[t[3]] += 1; lens[t[2]] += 1
            tot_tokens += len(toks)
            # windows: start at ip, take tokens starting within [ip, ip+64), stop at long literal
            j = 0; ip = 0; per_window = []
            while j < len(toks):
                if toks[j][3] == 9:
                    longlit += 1; ip = toks[j][0] + toks[j][1]; j += 1; per_window.append(None); continue
                out = 0
                while j < len(toks) and toks[j][0] < ip + 64 and toks[j][3] != 9 and out + toks[j][2] <= 1024:
                    out += toks[j][2]; j += 1
                ip = toks[j][0] if j < len(toks) else L
                per_window.append(out)
            wins = [w for w in per_window if w is not None]
            tot_windows += len(wins); tot_out += sum(wins)
            tot_steps1 += sum(-(-(w // G) // 64) for w in wins)
            # two windows per pass
            k2 = 0; a = 0
            seq = per_window
            idx = 0
            while idx < len(seq):
                if seq[idx] is None: idx += 1; continue
                o = seq[idx]; idx += 1
                if idx < len(seq) and seq[idx] is not None and o + seq[idx] <= 1024:
                    o += seq[idx]; idx += 1
                tot_steps2 += -(-(o // G) // 64)
        cp += L; frag_i += 1
    pos += sizes[c]
print("sampled fragments", len(sample), "tokens/frag %.1f" % (tot_tokens / len(sample)), "windows/frag %.1f" % (tot_windows / len(sample)),
      "long literals/frag %.2f" % (longlit / len(sample)), "out bytes/window %.1f" % (tot_out / max(1, tot_windows)))
print("steps/frag: one window per pass %.1f, two per pass %.1f, ideal %.1f" % (tot_steps1 / len(sample), tot_steps2 / len(sample), (1 << frag_log2) / G / 64))
print("kinds", dict(kinds)); print("top lens", lens.most_common(12))
