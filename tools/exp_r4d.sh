#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in cur "$@"; do
  lib=$PWD/hap_amd/variants/libhap_amd_$v.so
  [ "$v" = cur ] && lib=$PWD/hap_amd/libhap_amd.so
  echo "== $v"
  HAP_AMD_LIBRARY=$lib python tools/probe_bcdecode.py 8 2>&1 | grep "GB/s"
done
