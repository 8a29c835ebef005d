#!/bin/bash
# round 5: one or two encode contexts in the pipelined step, 60 and 8 frames
cd $GRAFT_REPO_ROOT
python - <<'PY'
import torch, hap_amd, bench as B
dev = torch.device("cuda:0")
ctx, dec, enc2 = hap_amd.Context(0), hap_amd.Context(0), hap_amd.Context(0)
f = lambda: (torch.cuda.synchronize(), ctx.synchronize(), dec.synchronize(), enc2.synchronize())
for nf, steps in ((60, 30), (8, 60), (16, 40), (4, 60)):
    for name, e2 in (("one encode context", None), ("two encode contexts", enc2)):
        s = B.Stream(hap_amd, ctx, dev, "C4", list(range(nf)), hap_amd.ENCODE_FRAGMENT_INDEX, ctx_dec=dec, ctx_enc2=e2)
        best = min(s.timed(steps, 2, f, pipelined=True)[0] for _ in range(3))
        print("%2d frames, %-20s %.4f ms per step  bit_exact %s" % (nf, name, best / steps * 1e3, s.bit_exact()))
        del s
        torch.cuda.empty_cache()
PY
