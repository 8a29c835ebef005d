for round in 1 2; do
for v in cur nopf; do
  HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_$v.so timeout 300 python bench.py --no-extras | tail -1 > /tmp/b.json
  python -c "
import json; d=json.load(open('/tmp/b.json')); print('$v', d['value'], d['ms_per_step'], d['bit_exact'], d['kernels']['encode_fused']['ms_avg'], d['kernels']['snappy_decode']['ms_avg'])"
done; done
