#!/bin/bash
cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, hap_amd, bench as B
dev = torch.device("cuda:0")
ctx = hap_amd.Context(0)
f = lambda: (torch.cuda.synchronize(), ctx.synchronize())
for cfg, nf in (("C4", 60), ("C5", 4), ("C4", 60), ("C5", 4)):
    s = B.Stream(hap_amd, ctx, dev, cfg, list(range(nf)), hap_amd.ENCODE_FRAGMENT_INDEX)
    e, prof = s.timed(12, 2, f)
    k, _ = s.kernel_table(prof, 12, cfg)
    print(cfg, "decode", k["snappy_decode"]["ms_avg"], "GBps", k["snappy_decode"]["algorithmic_GBps"], "step", e / 12 * 1e3)
    del s; torch.cuda.empty_cache()
PY
