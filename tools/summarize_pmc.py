"""Aggregates rocprofv3 counter_collection CSVs: mean counter value per dispatch, per kernel."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for path in sys.argv[1:]:
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("void ", "")
            k = k.split("(")[0][:70]
            c = row.get("Counter_Name"); v = float(row.get("Counter_Value", 0) or 0)
            a = acc[k][c]; a[0] += v; a[1] += 1
for k, cs in acc.items():
    print(k)
    for c, (s, n) in sorted(cs.items()):
        print("   %-26s mean/dispatch %16.1f   dispatches %d" % (c, s / n, n))
