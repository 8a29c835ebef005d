"""tools/prof_bench.sh output -> profiles/<round>_traffic_<cfg>.json (what bench.py's roofline.traffic reads).
   python tools/traffic_json.py gpurun_out/prof_c4 C4 30 [r03]
Per bench.py kernel class: mean FETCH_SIZE / WRITE_SIZE (KiB as the TCC counters report them) per launch, summed over
the kernels of the class (one launch of the class = one launch of each).  FETCH_SIZE is doubled by the reader
(MI355X_MICROARCH.md: wide streaming reads are under-reported 2x on gfx950)."""
import json, re, sys
src, cfg, frames = sys.argv[1], sys.argv[2], int(sys.argv[3])
rnd = sys.argv[4] if len(sys.argv) > 4 else "r03"
# (regular expressions over the kernel names; the block compressor with the block encoder inside -- second template
# argument 0..3 -- is bench.py's class "encode_fused")
FUSED = r"snappy_compress_blocks_kernel<\d+u?, ?[0-3]>"
classes = {"block_encode": ["bc_encode"], "snappy_compress": [r"snappy_compress(?!_blocks_kernel<\d+u?, ?[0-3]>)"], "encode_fused": [FUSED],
           "frame_pack": ["frame_chunk_sums", "frame_pack", "frame_moves"],
           "frame_gather": ["frame_gather"], "decode_plan": ["decode_plan", "decode_expand"], "snappy_decode": ["snappy_decode"]}
cur, vals = None, {}
for line in open(src + "/traffic_summary.txt"):
    if not line.startswith(" "):
        cur = line.strip()
        continue
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+mean/dispatch\s+([\d.]+)", line)
    if m:
        vals.setdefault(cur, {})[m.group(1)] = float(m.group(2))
# vector instructions per dispatch, from the SQ pass of the same command (bench.py's roofline.issue_bound)
valu = {}
try:
    for line in open(src + "/pmc_summary.txt"):
        if not line.startswith(" "):
            cur = line.strip()
            continue
        m = re.match(r"\s+SQ_INSTS_VALU\s+mean/dispatch\s+([\d.]+)", line)
        if m:
            valu[cur] = float(m.group(1))
except OSError:
    pass
out = {"config": cfg, "frames_per_launch": frames, "command": open(src + "/command.txt").read().strip(),
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/prof_bench.sh; raw per-kernel means in %s_traffic_summary_%s.txt" % (rnd, cfg.lower()),
       "kernels": {}, "per_kernel": vals}
# launches of each class per step, from the bench line of the same command (C5 decodes two textures: two launches)
line = json.loads([x for x in open(src + "/bench_traffic.json") if x.startswith("{")][-1])
for cls, pats in classes.items():
    f = sum(v.get("FETCH_SIZE", 0) for k, v in vals.items() if any(re.search(p, k) for p in pats))
    w = sum(v.get("WRITE_SIZE", 0) for k, v in vals.items() if any(re.search(p, k) for p in pats))
    per_step = max(1, round(line["kernels"].get(cls, {}).get("launches", line["steps"]) / line["steps"]))
    if f or w:
        out["kernels"][cls] = {"fetch_kib": round(f / per_step, 1), "write_kib": round(w / per_step, 1), "launches_per_step": per_step}
        vi = sum(v for k, v in valu.items() if any(re.search(p, k) for p in pats))
        if vi:
            out["kernels"][cls]["valu_insts"] = round(vi / per_step, 1)
json.dump(out, open("profiles/%s_traffic_%s.json" % (rnd, cfg.lower()), "w"), indent=1)
print(json.dumps(out["kernels"]))
