export HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_trace.so
python - <<'PY' 2>&1 | tail -40
import sys, torch, hap_amd, bench as B
dev = torch.device("cuda:0")
for nf in (8, 60):
    ctx = hap_amd.Context(0)
    s = B.Stream(hap_amd, ctx, dev, "C4", list(range(nf)), hap_amd.ENCODE_FRAGMENT_INDEX)
    s.step(); s.step()
    sys.stderr.write("---- %d frames\n" % nf); sys.stderr.flush()
    s.decode(s.used)
    ctx.timer_start(); s.decode(s.used); print(nf, "decode call ms", ctx.timer_stop())
PY
