#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/hq.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np, hap_amd, _libs as L
from hap_amd import synth
w, h, fmt, chunks = 7680, 4320, L.FMT_YCOCG, 24
ctx = hap_amd.Context(0)
nb = (w // 4) * (h // 4) * 16
api = L.ref_api() or L.oracle_api()
rgba = synth.rgba_frame(w, h, 0, device="cuda"); t = torch.zeros(nb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
assert ctx.compress_rgba(rgba, w, h, w * 4, fmt, t) == (0, nb)
r, frame = api.encode_np([t.cpu().numpy()], [fmt], [1], [chunks]); assert r == 0
fr = torch.from_numpy(frame).cuda(); out = torch.zeros(nb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
ctx.decode_frames([fr], [fr.numel()], 0, [out])
print("resolved", ctx.resolved_blocks())
PY
BRK_PRINT=1 HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_brkt.so python /tmp/hq.py 2>&1 | grep "merge\|resolved"
