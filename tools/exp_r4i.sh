set -x
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for v in 0 1; do
  if [ $v = 1 ]; then export HAP_AMD_NO_PLACING=1; fi
  timeout 300 python bench.py --no-extras 2>&1 | tail -1 > gpurun_out/b_placed_$v.json
  python - <<PY
import json
d=json.loads(open('gpurun_out/b_placed_$v.json').read())
print("NO_PLACING=$v", d.get("value"), d.get("ms_per_step"), d.get("bit_exact"), {k:(v["ms_avg"]) for k,v in d["kernels"].items()})
PY
done
