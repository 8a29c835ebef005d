"""Dev probe: the library's own host->device copy path (hipMemcpyAsync on its stream)."""
import ctypes as C, time, os, sys
import torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hap_amd
from hap_amd._lib import lib
lib.hapgpu_rt_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
lib.hapgpu_rt_device_scratch.restype = C.c_void_p
lib.hapgpu_rt_device_scratch.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
lib.hapgpu_rt_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
lib.hapgpu_rt_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
lib.hapgpu_rt_sync.argtypes = [C.c_void_p]
rt = C.c_void_p()
assert lib.hapgpu_rt_create(0, C.byref(rt)) == 0
n = 132710400
dev = lib.hapgpu_rt_device_scratch(rt, 0, n)
pag = np.full(n, 7, dtype=np.uint8)
pin = torch.empty(n, dtype=torch.uint8).pin_memory(); pin.fill_(7)
for name, ptr in (("pageable", pag.ctypes.data), ("pinned", pin.data_ptr())):
    for rep in range(3):
        t0 = time.perf_counter(); lib.hapgpu_rt_h2d(rt, dev, ptr, n); lib.hapgpu_rt_sync(rt); dt = time.perf_counter() - t0
    print(name, "h2d %.1f GB/s" % (n / dt / 1e9))
    for rep in range(3):
        t0 = time.perf_counter(); lib.hapgpu_rt_d2h(rt, ptr, dev, n); lib.hapgpu_rt_sync(rt); dt = time.perf_counter() - t0
    print(name, "d2h %.1f GB/s" % (n / dt / 1e9))
