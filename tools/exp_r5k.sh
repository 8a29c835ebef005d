#!/bin/bash
# round 5: where a decode call of fine-chunk frames spends its host time (HAPB_TRACE build)
cd $GRAFT_REPO_ROOT
HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_trace.so python - <<'PY' 2>&1 | tail -40
import torch, hap_amd, bench as B, sys
dev = torch.device("cuda:0")
ctx = hap_amd.Context(0)
w, h, fmts = 7680, 4320, [0x01]
tb = [(w // 4) * (h // 4) * 16]
fine = [hap_amd.fine_chunk_count(tb[0], 0x01)]
B.CONFIGS["C4fine"] = (w, h, fmts, fine, 60)
s = B.Stream(hap_amd, ctx, dev, "C4fine", list(range(60)), hap_amd.ENCODE_FINE_CHUNKS)
s.step(); s.step()
sys.stderr.flush()
print("---- one more decode call, traced above/below", flush=True)
s.decode(s.used)
ctx.timer_start(); s.decode(s.used); print("decode call ms", ctx.timer_stop())
PY
