#!/usr/bin/env python3
"""Dev tool (CPU only): static instruction counts of a HIP kernel by source line.

    python tools/isa_by_line.py hap_amd/csrc/snappy_decode_fields.hip [--func decode_fields_unit<4] [-D...]
        [--ranges 234:256=measure,288:355=walk,...]

Compiles the file for gfx950 with -gline-tables-only, attributes every instruction of the kernel to the `.loc`
in front of it and prints VALU / SALU / LDS / VMEM counts per source line (or per named line range).  The counts
are static (loops once, every inlined layout separately when --func is not given); trip counts are the
reader's business.  hipcc cross-compiles: no GPU needed.
"""
import collections
import re
import subprocess
import sys
import tempfile


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    args = sys.argv[1:]
    src = args[0]
    defs = [a for a in args[1:] if a.startswith("-D")]
    ranges = []
    func = None
    for i, a in enumerate(args):
        if a == "--ranges":
            for item in args[i + 1].split(","):
                span, name = item.split("=")
                lo, hi = span.split(":")
                ranges.append((int(lo), int(hi), name))
        if a == "--func":
            func = args[i + 1]
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-gline-tables-only",
               "-DHAP_MEASUREMENT_BUILD", "-Ihap_amd/csrc", "-S", "--cuda-device-only", "-o", tmp.name, src] + defs
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(tmp.name).read().splitlines()
    files = {}
    per = collections.defaultdict(collections.Counter)
    cur = None
    inlined_at = None
    for line in text:
        s = line.strip()
        m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", s)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
            continue
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        if cur is None:
            continue
        per[cur][classify(op)] += 1
    main_file = None
    for k, v in files.items():
        if v.endswith(src.split("/")[-1]):
            main_file = k
    tot = collections.Counter()
    rows = []
    if ranges:
        acc = collections.defaultdict(collections.Counter)
        for (fid, ln), c in per.items():
            name = "other:" + files.get(fid, "?").split("/")[-1]
            if fid == main_file:
                name = "unranged"
                for lo, hi, nm in ranges:
                    if lo <= ln <= hi:
                        name = nm
                        break
            acc[name] += c
        rows = sorted(acc.items(), key=lambda kv: -kv[1]["valu"])
    else:
        rows = sorted((("%s:%d" % (files.get(fid, "?").split("/")[-1], ln), c) for (fid, ln), c in per.items()),
                      key=lambda kv: -kv[1]["valu"])[:60]
    print("%-28s %6s %6s %6s %6s %6s" % ("where", "valu", "salu", "lds", "vmem", "wait"))
    for name, c in rows:
        tot += c
        print("%-28s %6d %6d %6d %6d %6d" % (name, c["valu"], c["salu"], c["lds"], c["vmem"], c["wait"]))
    print("%-28s %6d %6d %6d %6d %6d" % ("total (listed)", tot["valu"], tot["salu"], tot["lds"], tot["vmem"], tot["wait"]))


if __name__ == "__main__":
    main()
