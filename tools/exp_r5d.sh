#!/bin/bash
# round 5, fourth GPU call: -m gpu suite (table-less field decode, competing kernel, environment switches), bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5d; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep "placed under competition" $O/pytest.log; tail -3 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5d/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "bit_exact", "serial_step", "roofline", "small_batch", "plain_frames_batched", "fine_chunks_option"):
    print(k, json.dumps(d.get(k)))
c5 = d.get("c5") or {}
for k in ("value", "roofline", "decode_by_layout"):
    print("c5", k, json.dumps(c5.get(k)))
PY
