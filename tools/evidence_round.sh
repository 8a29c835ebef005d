#!/bin/bash
# Round evidence in one gpurun call: rocprofv3 passes over bench.py's own command for C4, C5, C2 and C3 (kernel trace + stats,
# SQ counters, FETCH_SIZE / WRITE_SIZE in separate runs: tools/prof_bench.sh), the foreign-frame probe under rocprofv3
# (tools/prof_foreign.sh), the default bench line and the 4K configs.  Everything lands under gpurun_out/evidence/;
# tools/evidence_collect.py copies it into profiles/<round>_* afterwards (run on the development box).
#   tools/evidence_round.sh
cd $GRAFT_REPO_ROOT
E=$GRAFT_REPO_ROOT/gpurun_out/evidence
rm -rf $E; mkdir -p $E
bash tools/prof_bench.sh C4 60 > /dev/null 2>&1; cp -r gpurun_out/prof_c4 $E/c4
bash tools/prof_bench.sh C4 60 sep > /dev/null 2>&1; cp -r gpurun_out/prof_c4sep $E/c4sep
bash tools/prof_bench.sh C5 4 > /dev/null 2>&1; cp -r gpurun_out/prof_c5 $E/c5
bash tools/prof_bench.sh C2 60 > /dev/null 2>&1; cp -r gpurun_out/prof_c2 $E/c2
bash tools/prof_bench.sh C3 60 > /dev/null 2>&1; cp -r gpurun_out/prof_c3 $E/c3
bash tools/prof_foreign.sh 24 > $E/foreign_stdout.txt 2>&1; cp -r gpurun_out/foreign $E/foreign
python bench.py > $E/bench_default.json 2> $E/bench_default.err
tail -c 600 $E/bench_default.json
