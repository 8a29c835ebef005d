#!/bin/bash
# HBM traffic of the pipeline's kernels from the TCC counters (separate --pmc passes, no other
# trace domains): FETCH_SIZE and WRITE_SIZE are in KiB-ish units of the memory-side request
# counters; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md
# section HBM), so both the raw and the doubled value are recorded.
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 2 --warmup 1 --frames ${FRAMES:-30} --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o r1 -- $CMD > /dev/null 2> /tmp/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o r1 -- $CMD > $OUT/bench_traffic.json 2> /tmp/write.err
python tools/summarize_pmc.py $(find /tmp/prof_fetch -name "*counter_collection.csv") $(find /tmp/prof_write -name "*counter_collection.csv") > $OUT/traffic_summary.txt
