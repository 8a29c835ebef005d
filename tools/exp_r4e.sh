#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  lib=$PWD/hap_amd/variants/libhap_amd_$v.so
  [ "$v" = cur ] && lib=$PWD/hap_amd/libhap_amd.so
  echo "== $v"
  HAP_AMD_LIBRARY=$lib python bench.py --no-extras --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
k=d['kernels']
print('value %.1f bit_exact %s ratio %.4f | ' % (d['value'], d['bit_exact'], d['config']['snappy_ratio']) + ' '.join('%s %.3f' % (n, k[n]['ms_avg']) for n in k))"
done
