#!/bin/bash
# plain frames through the block-per-lane kernel by the scan's pieces: the new test, then the bench's plain-frames leg
# with the pre-pass (default from 4096 pieces on) and without (--no-field-guess is not a bench flag: HAP_AMD env n/a) --
# the A/B is the decode flag, driven from python below
cd $GRAFT_REPO_ROOT

timeout 600 python - <<'PY' > gpurun_out/r5p_plain.log 2>&1
import time, json, torch, numpy as np
import bench, hap_amd
dev = torch.device("cuda:0")
ctx = hap_amd.Context()
for config, frames in (("C4", 60), ("C4", 30), ("C4", 16), ("C4", 8), ("C4", 4), ("C4", 2), ("C5", 4), ("C5", 1)):
    s = bench.Stream(hap_amd, ctx, dev, config, list(range(frames)), 0)
    s.step()
    for name, flags in (("guess", hap_amd.DECODE_GUESS_FIELDS), ("no_guess", hap_amd.DECODE_NO_FIELD_GUESS), ("guess", hap_amd.DECODE_GUESS_FIELDS), ("no_guess", hap_amd.DECODE_NO_FIELD_GUESS)):
        for idx in range(len(s.fmts)):
            ctx.decode_frames(s.frames, s.used, idx, s.dec[idx], flags=flags)
        torch.cuda.synchronize()
        ctx.set_profiling(True); ctx.collect_profile()
        t0 = time.perf_counter()
        for _ in range(6):
            for idx in range(len(s.fmts)):
                r = ctx.decode_frames(s.frames, s.used, idx, s.dec[idx], flags=flags)
                assert r[0] == 0, r
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        prof = ctx.collect_profile(); ctx.set_profiling(False)
        print(config, frames, name, "decode ms/step %.3f" % ms, {k: (v[0], round(v[1] / 6, 3)) for k, v in prof.items()}, "bit_exact", s.bit_exact(reference=False), flush=True)
print("fallbacks", ctx.table_fallbacks())
for frames in ():
    print("fine chunks", frames, json.dumps(bench.fine_chunks_option(hap_amd, ctx, dev, "C4", frames, torch.cuda.synchronize)), flush=True)
PY
cat gpurun_out/r5p_plain.log | tail -30
