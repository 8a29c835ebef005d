#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for i in 1 2; do
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r5w_$i.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/r5w_$i.json").read().strip().splitlines()[-1])
print("value", d["value"], "serial", d["serial_step"]["ms_per_step"], {k: v["ms_avg"] for k, v in d["kernels"].items()})
PY
timeout 300 python bench.py --config C5 --frames 4 --steps 12 --warmup 3 --no-cpu-baseline --no-extras --serial > gpurun_out/r5w_c5_$i.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/r5w_c5_$i.json").read().strip().splitlines()[-1])
print("c5 value", d["value"], {k: v["ms_avg"] for k, v in d["kernels"].items()})
PY
done
timeout 200 python tools/stress.py 301 100 2>&1 | tail -1
