"""Summarise a rocprofv3 --pmc counter_collection CSV per kernel (mean over dispatches)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", "")[:60]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in sorted(acc.items()):
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("    %-28s mean %14.1f  n=%d" % (c, sum(v) / len(v), len(v)))
