timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for v in 1 0; do
  export HAP_AMD_COPY_KERNELS=$v
  for nf in 60 8; do
    timeout 300 python bench.py --no-extras --frames $nf 2>&1 | tail -1 > /tmp/b.json
    python -c "
import json; d=json.load(open('/tmp/b.json')); print('mapped=$v frames $nf', d['value'], d['ms_per_step'], d['bit_exact'], d['encode_only']['ms'], d['decode_only']['ms'])"
  done
done
