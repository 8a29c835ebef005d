"""Does running two encode pipelines on two streams at once raise the throughput?  (block encode is VALU-only, the
compressor leans on the scalar unit: their instructions could share a SIMD's cycles.)
    python tools/probe_overlap.py [frames per call] [seconds]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
from hap_amd import synth
w, h, fmt, chunks = 7680, 4320, 0x01, 24
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 15
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
tb = (w // 4) * (h // 4) * 16
cap = hap_amd.HapMaxEncodedLength([tb], [fmt], [chunks])
def make(ctx):
    rgba = hap_amd.BufferList([synth.rgba_frame(w, h, i % 6, device="cuda") for i in range(nf)])
    frames = hap_amd.BufferList([torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)])
    return rgba, frames
ctxs = [hap_amd.Context(0), hap_amd.Context(0)]
data = [make(c) for c in ctxs]
torch.cuda.synchronize()
def loop(i, stop, count, delay=0.0):
    ctx = ctxs[i]; rgba, frames = data[i]
    time.sleep(delay)
    while not stop.is_set():
        r, used, _ = ctx.encode_frames_rgba(rgba, w, h, w * 4, [fmt], [1], [chunks], frames, flags=1)
        assert r == 0
        count[i] += nf
for i in (0, 1):
    ctxs[i].encode_frames_rgba(data[i][0], w, h, w * 4, [fmt], [1], [chunks], data[i][1], flags=1)
for threads in (1, 2):
    stop = threading.Event(); count = [0, 0]
    ts = [threading.Thread(target=loop, args=(i, stop, count, 0.0005 * i)) for i in range(threads)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    time.sleep(secs)
    stop.set()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print("%d stream(s): %.0f frames/s encode (%.3f ms per frame)" % (threads, sum(count) / dt, dt / max(1, sum(count)) * 1e3))
