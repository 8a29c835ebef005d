#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun): kernel trace + stats, then PMC passes
# (counters in their own runs, never combined with other trace domains).
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 2 --warmup 1 --frames ${FRAMES:-12} --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_trace -o r1 -- $CMD > $OUT/bench_trace.json 2> /tmp/trace.err
cp $(find /tmp/prof_trace -name "*kernel_stats.csv") $OUT/kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/prof_pmc1 -o r1 -- $CMD > /dev/null 2> /tmp/pmc1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d /tmp/prof_pmc2 -o r1 -- $CMD > /dev/null 2> /tmp/pmc2.err
python tools/summarize_pmc.py $(find /tmp/prof_pmc1 -name "*counter_collection.csv") $(find /tmp/prof_pmc2 -name "*counter_collection.csv") > $OUT/pmc_summary.txt
tail -3 /tmp/trace.err /tmp/pmc1.err /tmp/pmc2.err > $OUT/errs.txt
