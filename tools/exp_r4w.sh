for round in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export HAP_AMD_NO_GRAPHS=1; else unset HAP_AMD_NO_GRAPHS; fi
  for nf in 60 8; do
    timeout 300 python bench.py --no-extras --frames $nf 2>&1 | tail -1 > /tmp/b.json
    python -c "
import json; d=json.load(open('/tmp/b.json')); print('NO_GRAPHS=$v frames $nf', d['value'], d['ms_per_step'], d['encode_only']['ms'], d['decode_only']['ms'])"
  done
done; done
unset HAP_AMD_NO_GRAPHS
python - <<'PY'
import os, torch, hap_amd, bench as B
for ng in ("", "1"):
    if ng: os.environ["HAP_AMD_NO_GRAPHS"] = "1"
    ctx = hap_amd.Context(0)
    s = B.Stream(hap_amd, ctx, torch.device("cuda:0"), "C1", [0], 0)
    for _ in range(5): s.used = s.encode()
    best = 1e9
    for _ in range(20):
        ctx.timer_start(); s.used = s.encode(); best = min(best, ctx.timer_stop())
    print("C1 one frame encode call, NO_GRAPHS=%r: %.4f ms" % (ng, best))
PY
