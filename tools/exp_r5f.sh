#!/bin/bash
# round 5: the group-table pre-pass against wavefronts per CU (dynamic LDS it never touches), measurement build
cd $GRAFT_REPO_ROOT
for lds in 0 10000 20000 40000 65536; do
HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_ab.so HAP_AMD_GUESS_LDS=$lds python - <<'PY'
import os, torch, hap_amd, bench as B
ctx = hap_amd.Context(0)
r = B.fine_chunks_option(hap_amd, ctx, torch.device("cuda:0"), "C4", 30, lambda: (torch.cuda.synchronize(), ctx.synchronize()))
print("lds", os.environ["HAP_AMD_GUESS_LDS"], "decode_ms", r["decode_ms"], r["kernels_ms"], r["bit_exact"])
PY
done
