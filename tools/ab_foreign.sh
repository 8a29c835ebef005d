#!/bin/bash
# Dev tool (gpurun box): tools/probe_foreign.py under each library variant built by tools/build_variants.sh
#   tools/ab_foreign.sh 24 name1 name2 ...   ("cur" = hap_amd/libhap_amd.so)
cd $GRAFT_REPO_ROOT
nf=$1; shift
for v in "$@"; do
  lib=$PWD/hap_amd/variants/libhap_amd_$v.so
  [ "$v" = cur ] && lib=$PWD/hap_amd/libhap_amd.so
  echo "== $v"
  HAP_AMD_LIBRARY=$lib python tools/probe_foreign.py $nf 2>&1 | grep "block scan\|MISMATCH"
done
