#!/bin/bash
# Dev tool (gpurun box): kernel times + SQ counters of the block scan and the stream decode kernel under
# tools/probe_foreign.py.    tools/prof_foreign.sh [frames]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/foreign
rm -rf $OUT; mkdir -p $OUT
CMD="python tools/probe_foreign.py ${1:-24}"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_trace -o r -- $CMD > $OUT/probe.txt 2> /tmp/pf_trace.err
cp $(find /tmp/pf_trace -name "*kernel_stats.csv") $OUT/kernel_stats.csv
cp $(find /tmp/pf_trace -name "*kernel_trace.csv") /tmp/pf_ktrace.csv
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d /tmp/pf_pmc1 -o r -- $CMD > /dev/null 2> /tmp/pf_pmc1.err
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS --output-format csv -d /tmp/pf_pmc2 -o r -- $CMD > /dev/null 2> /tmp/pf_pmc2.err
python tools/summarize_pmc.py $(find /tmp/pf_pmc1 -name "*counter_collection.csv") $(find /tmp/pf_pmc2 -name "*counter_collection.csv") > $OUT/pmc_summary.txt
tail -6 $OUT/probe.txt
grep -i "scan_\|decode" $OUT/kernel_stats.csv | cut -c1-220
python - <<'PY'
import csv
rows = list(csv.DictReader(open('/tmp/pf_ktrace.csv')))
# the last call of the run: 24 frames with the scan
names = ["scan_walk", "scan_merge", "scan_find", "snappy_decode_fragment_kernel<4096"]
for n in names:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if n in r["Kernel_Name"]]
    if d:
        print("%-40s last 4 launches (us): %s" % (n, ["%.1f" % x for x in d[-4:]]))
PY
