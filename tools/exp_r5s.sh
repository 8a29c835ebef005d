#!/bin/bash
# kernel trace of the decode call of 60 8K frames, plain (no private table) and with the table, in one process: which launch takes what
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/plain_decode.py <<'PY'
import torch, bench, hap_amd
dev = torch.device("cuda:0")
ctx = hap_amd.Context()
for flags in (0, hap_amd.ENCODE_FRAGMENT_INDEX):
    s = bench.Stream(hap_amd, ctx, dev, "C4", list(range(60)), flags)
    s.step()
    for _ in range(3):
        s.decode(s.used)
    torch.cuda.synchronize()
    del s
PY
rm -rf /tmp/pd_trace
PYTHONPATH=$GRAFT_REPO_ROOT rocprofv3 --kernel-trace --output-format csv -d /tmp/pd_trace -o r -- python /tmp/plain_decode.py > /dev/null 2> /tmp/pd.err; tail -2 /tmp/pd.err
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/pd_trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "at::native" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 15:
        print("%-70s %9.1f us  grid %s" % (r["Kernel_Name"][:70], d, r.get("Grid_Size", r.get("Grid_Size_X", "?"))))
PY
