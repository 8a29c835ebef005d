#!/bin/bash
# two-texture frames, first texture placed or not, at several chunk counts (serial steps, kernel classes by HIP events)
cd $GRAFT_REPO_ROOT
for mode in placed gathered; do
if [ $mode = gathered ]; then export HAP_AMD_NO_PLACING=1; else unset HAP_AMD_NO_PLACING; export HAP_AMD_PLACING_MIN_FRAMES=1; fi
timeout 600 python - <<PY 2>&1 | grep -v amdgpu.ids
import time, json, torch
import bench, hap_amd
dev = torch.device("cuda:0")
ctx = hap_amd.Context()
FQ, FA = bench.CONFIGS["C5"][2]
for name, (w, h, chunks, frames) in {"8K a24": (7680, 4320, [24, 24], 16), "8K a64": (7680, 4320, [64, 64], 16), "8K a8": (7680, 4320, [8, 8], 16),
                                      "16K a64": (16384, 16384, [64, 64], 4), "16K a256": (16384, 16384, [256, 64], 4)}.items():
    bench.CONFIGS["X"] = (w, h, [FQ, FA], chunks, frames)
    s = bench.Stream(hap_amd, ctx, dev, "X", list(range(frames)), hap_amd.ENCODE_FRAGMENT_INDEX)
    el, prof = min((s.timed(5, 2, torch.cuda.synchronize), s.timed(5, 0, torch.cuda.synchronize)), key=lambda r: r[0])
    print("$mode", name, "ms/step %.3f" % (el / 5 * 1e3), {k: round(v[1] / 5, 3) for k, v in prof.items() if v[0]}, "bit_exact", s.bit_exact(), "retries", ctx.placement_retries(), flush=True)
    del s
PY
done
