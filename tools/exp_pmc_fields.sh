#!/bin/bash
# Dev tool (gpurun box): SQ counters of the field-stream decoder for library variants
#   tools/exp_pmc_fields.sh C5 2 cur w24 ...
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cfg=$1; nf=$2; shift 2
for v in "$@"; do
  lib=$PWD/hap_amd/variants/libhap_amd_$v.so
  [ "$v" = cur ] && lib=$PWD/hap_amd/libhap_amd.so
  rm -rf /tmp/pmc_$v
  HAP_AMD_LIBRARY=$lib rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d /tmp/pmc_$v -o r -- python tools/probe_fields.py $cfg $nf > /dev/null 2> /tmp/pmc_$v.err
  echo "== $v"
  python tools/summarize_pmc.py $(find /tmp/pmc_$v -name "*counter_collection.csv") | grep -A9 "snappy_decode_fields" | grep -v "^--"
done
