#!/bin/bash
# Round-2 profiling recipe (gpurun box): kernel trace + stats, SQ counters and HBM traffic counters of bench.py's
# own command, per config -- counters in their own runs, never combined with other trace domains.
#   tools/prof_bench.sh C4 60        -> gpurun_out/prof_c4/{kernel_stats.csv, pmc_summary.txt, traffic_summary.txt, bench_*.json}
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
CFG=${1:-C4}; FRAMES=${2:-60}
tag=$(echo $CFG | tr A-Z a-z)
# third argument "sep": the same command with the block encoder as a pass of its own and the fragments gathered
# (HAP_AMD_NO_FUSION / HAP_AMD_NO_PLACING) -> gpurun_out/prof_<cfg>sep
if [ "$3" = sep ]; then export HAP_AMD_NO_FUSION=1 HAP_AMD_NO_PLACING=1; tag=${tag}sep; fi
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $OUT; mkdir -p $OUT
# (--serial: blocking calls on one context, so that every kernel runs by itself and its duration is its own -- the
# region the bench line takes its per-kernel events and roofline from; frames per launch = the bench line's)
# (r06: the bench's own step counts -- r05 profiled 2 steps after 1 warm-up and got launches 8-13 % slower than the
# events of a 20-step region: cold clocks)
STEPS=${STEPS:-20}; WARMUP=${WARMUP:-5}
CMD="python bench.py --config $CFG --steps $STEPS --warmup $WARMUP --frames $FRAMES --no-extras --serial"
echo "${3:+HAP_AMD_NO_FUSION=1 HAP_AMD_NO_PLACING=1 }$CMD" > $OUT/command.txt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_trace -o r -- $CMD > $OUT/bench_trace.json 2> /tmp/pb_trace.err
cp $(find /tmp/pb_trace -name "*kernel_stats.csv") $OUT/kernel_stats.csv
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pb_pmc1 -o r -- $CMD > /dev/null 2> /tmp/pb_pmc1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d /tmp/pb_pmc2 -o r -- $CMD > /dev/null 2> /tmp/pb_pmc2.err
python tools/summarize_pmc.py $(find /tmp/pb_pmc1 -name "*counter_collection.csv") $(find /tmp/pb_pmc2 -name "*counter_collection.csv") > $OUT/pmc_summary.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pb_fetch -o r -- $CMD > /dev/null 2> /tmp/pb_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pb_write -o r -- $CMD > $OUT/bench_traffic.json 2> /tmp/pb_write.err
python tools/summarize_pmc.py $(find /tmp/pb_fetch -name "*counter_collection.csv") $(find /tmp/pb_write -name "*counter_collection.csv") > $OUT/traffic_summary.txt
tail -2 /tmp/pb_trace.err /tmp/pb_pmc1.err /tmp/pb_fetch.err > $OUT/errs.txt 2>&1
grep -c . $OUT/kernel_stats.csv $OUT/pmc_summary.txt $OUT/traffic_summary.txt
