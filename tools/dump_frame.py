"""Dev tool: encodes one C4 frame on the GPU and writes it to gpurun_out/frame_c4.bin for offline stream statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, hap_amd
from hap_amd import synth
w, h = 7680, 4320
ctx = hap_amd.Context(0)
img = synth.rgba_frame(w, h, 3, device="cuda")
out = torch.zeros(hap_amd.HapMaxEncodedLength([w * h], [1], [24]), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
r, used, res = ctx.encode_frames_rgba([img], w, h, w * 4, [1], [1], [24], [out], flags=1)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "frame_c4.bin"), "wb").write(out[: used[0]].cpu().numpy().tobytes())
print("wrote", used[0])
