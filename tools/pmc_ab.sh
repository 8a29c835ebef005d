#!/bin/bash
# Dev tool: dynamic instruction counts of the compress / decode kernels for library variants (gpurun box).
#   tools/pmc_ab.sh old cur
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in "$@"; do
  rm -rf /tmp/pmc_$v
  HAP_AMD_LIBRARY=$PWD/hap_amd/variants/libhap_amd_$v.so rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d /tmp/pmc_$v -o r -- python tools/time_kernels.py 12 > /dev/null 2> /tmp/pmc_$v.err
  echo "== $v"
  python tools/summarize_pmc.py $(find /tmp/pmc_$v -name "*counter_collection.csv") | grep -A9 "snappy_compress_wg\|snappy_decode_fragment" | grep -v "^--"
done
