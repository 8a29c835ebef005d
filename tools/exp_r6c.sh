#!/bin/bash
cd $GRAFT_REPO_ROOT
cat > /tmp/pl.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, hap_amd
from hap_amd import synth
w, h = 7680, 4320
ctx = hap_amd.Context(0)
nb = (w // 4) * (h // 4) * 16
cap = hap_amd.HapMaxEncodedLength([nb], [1], [24])
rgba = [synth.rgba_frame(w, h, 0, device="cuda")]
frames = [torch.zeros(cap, dtype=torch.uint8, device="cuda")]
torch.cuda.synchronize()
r, used, res = ctx.encode_frames_rgba(rgba, w, h, w * 4, [1], [1], [24], frames, flags=0)
dec = [torch.zeros(nb, dtype=torch.uint8, device="cuda")]
torch.cuda.synchronize()
ctx.decode_frames(frames, used, 0, dec)
print("resolved", ctx.resolved_blocks())
PY
BRK_PRINT=1 HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_brkt.so python /tmp/pl.py 2>&1 | grep "merge\|resolved"
