#!/bin/bash
# is the first bench process on a fresh box slower than the next ones?  (the driver's run is a first process)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras $EXTRA > gpurun_out/r5x_$i.json 2>/dev/null
python - <<PY
import json
d = json.loads(open("gpurun_out/r5x_$i.json").read().strip().splitlines()[-1])
print("run $i value", d["value"], "ms", d["ms_per_step"], "serial", d["serial_step"]["ms_per_step"], {k: v["ms_avg"] for k, v in d["kernels"].items()})
PY
done
