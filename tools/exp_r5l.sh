#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q -k "fine or field or stream or table" > gpurun_out/r5l_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5l_pytest.log
python - <<'PY'
import torch, hap_amd, bench as B
ctx = hap_amd.Context(0)
for nf in (60, 30):
    r = B.fine_chunks_option(hap_amd, ctx, torch.device("cuda:0"), "C4", nf, lambda: (torch.cuda.synchronize(), ctx.synchronize()))
    print(nf, "decode_ms", r["decode_ms"], r["kernels_ms"], r["bit_exact"], "fallbacks", ctx.table_fallbacks())
PY
timeout 200 python tools/stress.py 91 60 2>&1 | tail -2
