#!/bin/bash
# Dev tool (runs on the GPU box): one bench line per library variant and workload
#   tools/ab_bench.sh "C4 C5" hap_amd/libhap_amd.so hap_amd/variants/libhap_amd_x.so ...
cfgs="$1"; shift
for lib in "$@"; do for cfg in $cfgs; do
  HAP_AMD_LIBRARY=$PWD/$lib python bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --config $cfg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib'.split('/')[-1], '$cfg', d['value'], d['bit_exact'], d['config'].get('snappy_ratio'), {k:round(v['ms_avg'],3) for k,v in d['kernels'].items()})"
done; done
