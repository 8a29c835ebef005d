"""gpurun_out/evidence (tools/evidence_round.sh) -> profiles/<round>_*: the files the documentation and bench.py quote.
   python tools/evidence_collect.py r03"""
import json, os, shutil, subprocess, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ev, prof = os.path.join(root, "gpurun_out", "evidence"), os.path.join(root, "profiles")
for cfg, frames in (("c4", 60), ("c4sep", 60), ("c5", 4), ("c2", 60), ("c3", 60)):       # (frames per launch: the bench line's)
    if not os.path.isdir(os.path.join(ev, cfg)):
        continue
    src = os.path.join(ev, cfg)
    for a, b in (("kernel_stats.csv", "kernel_stats"), ("pmc_summary.txt", "pmc_summary"), ("traffic_summary.txt", "traffic_summary"),
                 ("command.txt", "command"), ("bench_trace.json", "bench_under_rocprof")):
        ext = os.path.splitext(a)[1]
        shutil.copy(os.path.join(src, a), os.path.join(prof, "%s_%s_%s%s" % (rnd, b, cfg, ext)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "traffic_json.py"), src, cfg.upper(), str(frames), rnd], check=True, cwd=root)
src = os.path.join(ev, "foreign")
for a, b in (("kernel_stats.csv", "kernel_stats"), ("pmc_summary.txt", "pmc_summary"), ("probe.txt", "probe")):
    shutil.copy(os.path.join(src, a), os.path.join(prof, "%s_%s_foreign_frames%s" % (rnd, b, os.path.splitext(a)[1])))
line = [x for x in open(os.path.join(ev, "bench_default.json")) if x.startswith("{")][-1]
json.dump(json.loads(line), open(os.path.join(prof, "%s_bench_default.json" % rnd), "w"), indent=1)
print("collected into profiles/%s_*" % rnd)
