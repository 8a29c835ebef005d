import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hap_amd
from hap_amd import synth
w, h = 7680, 4320
ctx = hap_amd.Context(0)
cap = hap_amd.HapMaxEncodedLength([w*h], [1], [24])
rgba = [synth.rgba_frame(w, h, 0, device="cuda")]
fr = [torch.zeros(cap, dtype=torch.uint8, device="cuda")]
torch.cuda.synchronize()
r, used, res = ctx.encode_frames_rgba(rgba, w, h, w*4, [1], [1], [24], fr, flags=1)
open("gpurun_out/c4_frame0_%s.hap" % os.environ.get("TAG", "cur"), "wb").write(fr[0][:used[0]].cpu().numpy().tobytes())
print(r, used)
