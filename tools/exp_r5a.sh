#!/bin/bash
# round 5, first GPU call: the whole -m gpu suite, the field-stream probe per config, the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for cfg in "C5 2" "C4 30" "C3 60" "C2 60"; do
  set -- $cfg
  timeout 300 python tools/probe_fields.py $1 $2 > $O/probe_$1.log 2>&1
  grep "^encode\|^fields\|PARITY" $O/probe_$1.log
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5a/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "bit_exact", "bit_exact_checked", "serial_step", "encode_only", "decode_only", "roofline", "small_batch", "plain_frames_batched"):
    print(k, json.dumps(d.get(k)))
c5 = d.get("c5") or {}
for k in ("value", "ms_per_step", "bit_exact", "roofline", "decode_by_layout", "encode_only", "decode_only"):
    print("c5", k, json.dumps(c5.get(k)))
for c in ("c2", "c3"):
    x = d.get(c) or {}
    print(c, x.get("value"), x.get("bit_exact"), json.dumps(x.get("roofline")))
print("kernels", json.dumps(d.get("kernels")))
print("pipelined kernels", json.dumps(d.get("kernels_in_pipelined_region_ms_avg")))
PY
