for v in 1 0; do
  export HAP_AMD_COPY_KERNELS=$v
  for nf in 60 8; do
    timeout 300 python bench.py --no-extras --frames $nf 2>&1 | tail -1 > /tmp/b.json
    python - <<PY
import json
d=json.loads(open('/tmp/b.json').read())
print("COPY_KERNELS=$v frames $nf", d.get("value"), d.get("ms_per_step"), d.get("bit_exact"), d["encode_only"]["ms"], d["decode_only"]["ms"])
PY
  done
  timeout 200 python tools/percall_probe.py 2>/dev/null | tail -2
done
unset HAP_AMD_COPY_KERNELS
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
