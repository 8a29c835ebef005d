"""GPU probe: device memory and host RSS across many batched calls (nothing may grow once the arenas exist)."""
import os, sys, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd, bench as B
dev = torch.device("cuda:0")
ctx = hap_amd.Context(0)
s = B.Stream(hap_amd, ctx, dev, "C3", list(range(16)), hap_amd.ENCODE_FRAGMENT_INDEX)
pics = hap_amd.BufferList([torch.empty(s.rgba_bytes, dtype=torch.uint8, device=dev) for _ in range(s.nf)])
def snap():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) >> 20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss >> 10
for _ in range(20):
    s.step(); ctx.decode_frames_rgba(s.frames, s.used, 1, pics, s.w, s.h)
a = snap()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2000):
    s.step(); ctx.decode_frames_rgba(s.frames, s.used, 1, pics, s.w, s.h)
b = snap()
print("device MiB in use %d -> %d, host max RSS MiB %d -> %d, bit_exact %s" % (a[0], b[0], a[1], b[1], s.bit_exact()))
