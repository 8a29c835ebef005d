"""GPU probe: do the hot path's kernels overlap when two contexts (two HIP streams) work on halves of a batch at the same
time?  Block encode is HBM-bound with the vector units about half busy, the Snappy kernels are bound by vector issue:
two streams could fill each other's gaps.  One context over 60 frames against two contexts (two host threads) over 30
each, encode only and encode + decode.      python tools/probe_two_streams.py [frames] [reps]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
import bench as B
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 60
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 2
dev = torch.device("cuda:0")
flags = hap_amd.ENCODE_FRAGMENT_INDEX
whole = B.Stream(hap_amd, hap_amd.Context(0), dev, "C4", list(range(nf)), flags)
halves = [B.Stream(hap_amd, hap_amd.Context(0), dev, "C4", list(range(k, nf, parts)), flags) for k in range(parts)]

def run(streams, what):
    def work(s):
        for _ in range(reps):
            if what == "encode":
                s.used = s.encode()
            elif what == "decode":
                s.decode(s.used)
            else:
                s.step()
    for s in streams:
        s.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(s,)) for s in streams]
    [t.start() for t in th]; [t.join() for t in th]
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for what in ("encode", "decode", "step"):
    a = run([whole], what); b = run(halves, what); a2 = run([whole], what); b2 = run(halves, what)
    print("%-7s one context x %d frames: %.3f / %.3f ms    %d contexts x %d: %.3f / %.3f ms" % (what, nf, a, a2, parts, nf // parts, b, b2))
