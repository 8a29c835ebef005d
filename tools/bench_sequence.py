"""Disk -> GPU decode of a Hap sequence file: writes N 8K Hap Q frames made by the GPU encoder
to a sequence file, then times HapGpuDecodeSequence (double-buffered read-ahead) against reading everything
first and decoding afterwards.  usage: bench_sequence.py [frames] [directory]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
from hap_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
where = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
w, h, fmt, chunks = 7680, 4320, 0x01, 24
ctx = hap_amd.Context(0)
cap = hap_amd.HapMaxEncodedLength([w * h], [fmt], [chunks])
path = os.path.join(where, "bench_8k.hapseq")
out = torch.empty(cap, dtype=torch.uint8, device="cuda")
total = 0
with hap_amd.SequenceWriter(path, w, h) as wr:
    for i in range(n):
        img = synth.rgba_frame(w, h, i % 8, device="cuda")
        torch.cuda.synchronize()
        r, used, res = ctx.encode_frames_rgba([img], w, h, w * 4, [fmt], [1], [chunks], [out], flags=1)
        assert r == 0
        frame = out[: used[0]].cpu().numpy()
        wr.append(frame); total += used[0]
dec = [torch.empty(w * h, dtype=torch.uint8, device="cuda") for _ in range(n)]
torch.cuda.synchronize()
rd = hap_amd.SequenceReader(path)
res = {"frames": n, "file_MB": round(total / 1e6, 1), "where": where}
for batch in (4, 8, 16, 30):
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        r, used, fmts, rs = ctx.decode_sequence(rd, 0, n, 0, dec, batch=batch)
        best = min(best, time.perf_counter() - t0)
        assert r == 0
    res["pipelined_batch%d" % batch] = {"ms": round(best * 1e3, 2), "fps": round(n / best, 1),
                                        "file_GBps": round(total / best / 1e9, 2), "rgba_GBps": round(n * w * h * 4 / best / 1e9, 1)}
# serial: read everything (pageable), then one decode call
t0 = time.perf_counter()
r, frames = rd.read(0, n)
t1 = time.perf_counter()
r2, used, fmts, rs = ctx.decode_frames(frames, [len(f) for f in frames], 0, dec)
t2 = time.perf_counter()
res["read_then_decode"] = {"read_ms": round((t1 - t0) * 1e3, 2), "decode_ms": round((t2 - t1) * 1e3, 2), "fps": round(n / (t2 - t0), 1)}
print(json.dumps(res))
os.remove(path)
