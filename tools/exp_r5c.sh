#!/bin/bash
# round 5, third GPU call: regression check of the header read-back, small batches with placing from 8 frames, fine chunks with the 2 KiB ring
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for pm in 12 8 4; do
HAP_AMD_PLACING_MIN_FRAMES=$pm python - <<'PY'
import os, torch, hap_amd, bench as B
ctx = hap_amd.Context(0); ctx2 = hap_amd.Context(0)
f = lambda: (torch.cuda.synchronize(), ctx.synchronize(), ctx2.synchronize())
r = B.small_batch(hap_amd, ctx, ctx2, torch.device("cuda:0"), "C4", hap_amd.ENCODE_FRAGMENT_INDEX, f, 3.42, 3.576, 60)
print("placing_min", os.environ["HAP_AMD_PLACING_MIN_FRAMES"], r["pipelined"], r["serial"]["ms_per_step"], r["serial"]["kernels_ms_per_step"], "retries", ctx.placement_retries(), ctx.placement_timeouts())
PY
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5c/bench.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "bit_exact", "serial_step", "encode_only", "decode_only", "roofline", "small_batch", "plain_frames_batched", "fine_chunks_option", "per_call_hap_h"):
    print(k, json.dumps(d.get(k)))
c5 = d.get("c5") or {}
for k in ("value", "ms_per_step", "bit_exact", "roofline", "decode_by_layout"):
    print("c5", k, json.dumps(c5.get(k)))
for c in ("c2", "c3"):
    x = d.get(c) or {}
    print(c, x.get("value"), x.get("bit_exact"), json.dumps(x.get("roofline")))
PY
