import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, torch, hap_amd
from hap_amd import synth
import _data as D, _libs as L
ctx = hap_amd.Context(0)
for (w, h) in ((2048, 512), (7680, 4320), (16384, 4096)):
    rgba = synth.rgba_frame(w, h, 0, device="cuda")
    n = (w // 4) * (h // 4) * 8
    tex = torch.zeros(n, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    ctx.compress_rgba(rgba, w, h, w * 4, 0x8DBB, tex)
    cap = hap_amd.HapMaxEncodedLength([n], [0x8DBB], [8])
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
    r, used, res = ctx.encode_frames([[tex]], [0x8DBB], [1], [8], [out], flags=1)
    ref = L.ref_api() or L.oracle_api()
    theirs = len(ref.encode_np([tex.cpu().numpy()], [0x8DBB], [1], [8])[1]) if w <= 7680 else 0
    print(w, h, "ours %.4f" % (used[0] / n), "libsnappy %.4f" % (theirs / n))
