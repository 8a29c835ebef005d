#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in cur "$@"; do
  lib=$PWD/hap_amd/variants/libhap_amd_$v.so
  [ "$v" = cur ] && lib=$PWD/hap_amd/libhap_amd.so
  echo "== $v"
  HAP_AMD_LIBRARY=$lib python tools/probe_plain.py C4 1 2>&1 | grep plain
  HAP_AMD_LIBRARY=$lib python tools/probe_plain.py C1 1 2>&1 | grep plain
  HAP_AMD_LIBRARY=$lib python tools/probe_foreign.py 24 2>&1 | grep "block scan"
  HAP_AMD_LIBRARY=$lib python -m pytest tests -m gpu -x -q -k "block_scan or table_less or smaller_files or malformed or foreign or full_size" 2>&1 | tail -1
done
