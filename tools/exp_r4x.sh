python - <<'PY'
import os, torch, hap_amd, bench as B
B.CONFIGS["C4one"] = (7680, 4320, [0x01], [1], 60)
B.CONFIGS["C4many"] = (7680, 4320, [0x01], [400], 60)
for cfg in ("C4one", "C4many"):
    for env in ({}, {"HAP_AMD_NO_PLACING": "1"}):
        os.environ.pop("HAP_AMD_NO_PLACING", None); os.environ.update(env)
        ctx = hap_amd.Context(0)
        s = B.Stream(hap_amd, ctx, torch.device("cuda:0"), cfg, list(range(16)), hap_amd.ENCODE_FRAGMENT_INDEX)
        s.step()
        best = 1e9
        for _ in range(4):
            ctx.timer_start(); s.used = s.encode(); best = min(best, ctx.timer_stop())
        s.decode(s.used)
        print(cfg, env, "encode %.3f ms retries %d bit_exact %s chunks %s" % (best, ctx.placement_retries(), s.bit_exact(), s.chunks))
        del s, ctx
PY
