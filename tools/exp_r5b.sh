#!/bin/bash
# round 5, second GPU call: -m gpu suite, bench line (small batch kernel table, fine chunks, plain frames), ring-size A/B for fine chunks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5b/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "bit_exact", "serial_step", "encode_only", "decode_only", "roofline", "small_batch", "plain_frames_batched", "fine_chunks_option", "per_call_hap_h"):
    print(k, json.dumps(d.get(k)))
c5 = d.get("c5") or {}
for k in ("value", "ms_per_step", "bit_exact", "roofline", "decode_by_layout"):
    print("c5", k, json.dumps(c5.get(k)))
for c in ("c2", "c3"):
    x = d.get(c) or {}
    print(c, x.get("value"), x.get("bit_exact"), json.dumps(x.get("roofline")))
PY
for ring in 11 13; do
HAP_AMD_STREAM_RING_LOG2=$ring python - <<'PY'
import os, torch, hap_amd, bench as B
ctx = hap_amd.Context(0)
print("ring", os.environ.get("HAP_AMD_STREAM_RING_LOG2"), B.fine_chunks_option(hap_amd, ctx, torch.device("cuda:0"), "C4", 30, lambda: (torch.cuda.synchronize(), ctx.synchronize())))
PY
done
