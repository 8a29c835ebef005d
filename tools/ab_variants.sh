#!/bin/bash
# Dev tool (gpurun box): tools/probe_fields.py under each library variant built by tools/build_variants.sh
#   tools/ab_variants.sh C4 30 name1 name2 ...   ("cur" = hap_amd/libhap_amd.so)
cd $GRAFT_REPO_ROOT
cfg=$1; nf=$2; shift 2
for v in "$@"; do
  lib=$PWD/hap_amd/variants/libhap_amd_$v.so
  [ "$v" = cur ] && lib=$PWD/hap_amd/libhap_amd.so
  echo "== $v"
  HAP_AMD_LIBRARY=$lib python tools/probe_fields.py $cfg $nf 2>&1 | grep "^encode\|^fields\|^generic\|PARITY\|checker"
done
