#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cat > /tmp/onef.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np, hap_amd, _libs as L
from hap_amd import synth
w, h, fmt, chunks = 7680, 4320, L.FMT_YCOCG, 24
ctx = hap_amd.Context(0)
nb = (w // 4) * (h // 4) * 16
api = L.ref_api() or L.oracle_api()
rgba = synth.rgba_frame(w, h, 0, device="cuda"); t = torch.zeros(nb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
assert ctx.compress_rgba(rgba, w, h, w * 4, fmt, t) == (0, nb)
r, frame = api.encode_np([t.cpu().numpy()], [fmt], [1], [chunks]); assert r == 0
fr = torch.from_numpy(frame).cuda(); out = torch.zeros(nb, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
for i in range(10):
    ctx.decode_frames([fr], [fr.numel()], 0, [out])
print("ok", bool(torch.equal(out, t)))
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k1 -o r -- python /tmp/onef.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/k1/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('scan_', 'snappy_decode', 'decode_plan', 'decode_expand', 'gather_prefix', 'small_')):
        print('%-60s calls %s avg us %.1f' % (r['Name'].replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs'])/1e3))
PY
