#!/bin/bash
# Dev tool (gpurun box): kernel timeline of one bench step (start offsets, durations, gaps) from rocprofv3's kernel trace.
#   tools/trace_gaps.sh [frames]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg -o r -- python bench.py --frames ${1:-8} --steps 3 --warmup 2 --no-extras > /dev/null 2> /tmp/tg.err
python - <<'PY'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('/tmp/tg/**/*kernel_trace.csv', recursive=True)[0])))
rows = [r for r in rows if 'at::native' not in r['Kernel_Name'] and 'elementwise' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last step = last 14 or so kernels: print the tail
# the last step: the kernels up to the last decode launch
last = max(i for i, r in enumerate(rows) if 'snappy_decode_fields_kernel' in r['Kernel_Name'])
tail = rows[max(0, last - 17):last + 2]
t0 = int(tail[0]['Start_Timestamp']); prev_end = None
for r in tail:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')[:46]
    print("%9.1f us  +%7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
    prev_end = e
PY
