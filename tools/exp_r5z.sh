#!/bin/bash
# the pre-pass's tag loads with cache hints (nt: streaming; sc1: past the L1): does the L2 -> L1 fill traffic go away?
cd $GRAFT_REPO_ROOT
for v in ab nt sc1; do
export HAP_AMD_LIBRARY=$GRAFT_REPO_ROOT/hap_amd/variants/libhap_amd_$v.so
timeout 300 python - <<PY 2>&1 | grep -v amdgpu.ids
import time, torch, bench, hap_amd
dev = torch.device("cuda:0")
ctx = hap_amd.Context()
for frames in (60, 8):
    s = bench.Stream(hap_amd, ctx, dev, "C4", list(range(frames)), 0)
    s.step()
    for rep in range(2):
        ctx.decode_frames(s.frames, s.used, 0, s.dec[0], flags=hap_amd.DECODE_GUESS_FIELDS)
        torch.cuda.synchronize()
        ctx.set_profiling(True); ctx.collect_profile()
        t0 = time.perf_counter()
        for _ in range(6):
            r = ctx.decode_frames(s.frames, s.used, 0, s.dec[0], flags=hap_amd.DECODE_GUESS_FIELDS)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        prof = ctx.collect_profile(); ctx.set_profiling(False)
        print("$v", frames, "frames: decode call %.3f ms" % ms, {k: (v[0], round(v[1] / 6, 3)) for k, v in prof.items() if v[0]}, "bit_exact", s.bit_exact(), flush=True)
    del s
PY
done
