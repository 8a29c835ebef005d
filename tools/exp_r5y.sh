#!/bin/bash
# one frame of another encoder per call: does a larger ring (fewer far reads, fewer waves per CU -- there are only two per CU) help?
cd $GRAFT_REPO_ROOT
export HAP_AMD_LIBRARY=$GRAFT_REPO_ROOT/hap_amd/variants/libhap_amd_ab.so
for ring in 11 12 13 14 15 16; do
  echo "ring_log2 $ring"
  HAP_AMD_STREAM_RING_LOG2=$ring timeout 200 python tools/probe_foreign.py 4 2>&1 | grep "block scan"
done
