#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "plain or scan or foreign or reference or checker or fine_chunk or table_less or malformed or corrupt or full_size" > gpurun_out/r5t_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r5t_pytest.log
bash tools/exp_r5s.sh 2>&1 | grep -v "rocprofv3\]" | head -14
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r5t_bench.json 2> gpurun_out/r5t_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5t_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
for k in ("plain_frames_batched", "fine_chunks_option", "decode_of_reference_encoded_frames", "per_call_hap_h"):
    print(k, json.dumps(d.get(k))[:700])
print("c5", json.dumps(d.get("c5", {}).get("decode_of_reference_encoded_frames"))[:400])
PY
