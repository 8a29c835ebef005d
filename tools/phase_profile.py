"""Dev tool: per-phase shader-clock cycles of the compress kernel (library built with -DHAP_PHASE_PROFILE)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import hap_amd
from hap_amd import synth
from hap_amd._lib import lib
w, h, fmts, chunks, nf = 7680, 4320, [0x01], [24], 12
ctx = hap_amd.Context(0)
rgba = [synth.rgba_frame(w, h, i, device="cuda") for i in range(nf)]
cap = hap_amd.HapMaxEncodedLength([w * h], fmts, chunks)
frames = [torch.empty(cap, dtype=torch.uint8, device="cuda") for _ in range(nf)]
torch.cuda.synchronize()
ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, [1], chunks, frames, flags=1)
out = (C.c_ulonglong * 8)()
lib.hapgpu_debug_phase_cycles.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
print("rc", lib.hapgpu_debug_phase_cycles(out, 1))
ctx.encode_frames_rgba(rgba, w, h, w * 4, fmts, [1], chunks, frames, flags=1)
lib.hapgpu_debug_phase_cycles(out, 1)
names = ["prologue", "analysis", "barrier1 wait", "emit+insert", "barrier2 wait"]
tot = sum(out[:5]); print("waves", out[5], "cycles per wave", tot // max(1, out[5]))
for n, v in zip(names, out[:5]):
    print("%-14s %14d  %5.1f %%" % (n, v, 100.0 * v / max(1, tot)))
