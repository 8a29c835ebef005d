#!/bin/bash
# two-texture frames with the first texture placed: tests that touch placing, then C5 with and without
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "placed or placing or does_not_shrink or full_size or retry or two_halves" > gpurun_out/r5q_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5q_pytest.log
for v in 0 1; do
  if [ $v = 1 ]; then export HAP_AMD_NO_PLACING=1; else export HAP_AMD_PLACING_MIN_FRAMES=1; fi
  timeout 600 python bench.py --config C5 --frames 4 --steps 8 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/r5q_c5_noplacing$v.json 2> gpurun_out/r5q_c5_noplacing$v.err; echo "rc=$?"
done
python - <<'PY'
import json
for v in (0, 1):
    try:
        d = json.loads(open("gpurun_out/r5q_c5_noplacing%d.json" % v).read().strip().splitlines()[-1])
        print("NO_PLACING=%d" % v, d["value"], d["ms_per_step"], d.get("serial_step"), {k: x["ms_avg"] for k, x in d["kernels"].items()}, d.get("bit_exact"))
    except Exception as e:
        print("v", v, "failed", e)
PY
