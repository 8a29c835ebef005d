import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, hap_amd
from hap_amd import synth
w, h = 7680, 4320
ctx = hap_amd.Context(0)
tb = (w // 4) * (h // 4) * 16
for frame in (0, 7):
    rgba = synth.rgba_frame(w, h, frame, device="cuda")
    tex = torch.empty(tb, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.compress_rgba(rgba, w, h, w * 4, 0x01, tex)
    for fmt, name in ((0x01, "YCoCg"), (0x8E8C, "BC7")):
        cap = hap_amd.HapMaxEncodedLength([tb], [fmt], [24])
        out = torch.empty(cap, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for flags in (0, hap_amd.ENCODE_FRAGMENT_INDEX):
            r, used, res = ctx.encode_frames([[tex]], [fmt], [1], [24], [out], flags=flags)
            print("frame", frame, name, "flags", flags, "ratio %.4f" % (used[0] / tb))
