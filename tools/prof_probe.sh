#!/bin/bash
# Dev tool (gpurun box): kernel times + SQ counters of the decode kernels under tools/probe_fields.py.
#   tools/prof_probe.sh C4 30
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/probe
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/probe_fields.py ${1:-C4} ${2:-30}"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_trace -o r -- $CMD > $OUT/probe.txt 2> /tmp/pp_trace.err
cp $(find /tmp/pp_trace -name "*kernel_stats.csv") $OUT/kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pp_pmc1 -o r -- $CMD > /dev/null 2> /tmp/pp_pmc1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d /tmp/pp_pmc2 -o r -- $CMD > /dev/null 2> /tmp/pp_pmc2.err
python tools/summarize_pmc.py $(find /tmp/pp_pmc1 -name "*counter_collection.csv") $(find /tmp/pp_pmc2 -name "*counter_collection.csv") > $OUT/pmc_summary.txt
cat $OUT/probe.txt | tail -12
grep -i "decode" $OUT/kernel_stats.csv | cut -c1-200
grep -A17 "snappy_decode_fields\|snappy_decode_fragment_kernel<4096" $OUT/pmc_summary.txt | head -60
