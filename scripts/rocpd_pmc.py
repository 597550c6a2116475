#!/usr/bin/env python
"""Per-kernel mean of every PMC counter in a rocprofv3 rocpd database:  python scripts/rocpd_pmc.py <db> [name-filter]"""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
kn = ix.get("kernel_name", ix.get("name"))
cn = ix.get("counter_name")
cv = ix.get("value", ix.get("counter_value"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[r[kn]][r[cn]].append(float(r[cv]))
print("kernel,counter,mean,calls")
for k in sorted(acc):
    if flt and flt not in k:
        continue
    for cname, v in sorted(acc[k].items()):
        print('"%s",%s,%.1f,%d' % (k[:70], cname, sum(v) / len(v), len(v)))
