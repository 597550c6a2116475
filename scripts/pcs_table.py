#!/usr/bin/env python
"""PC samples of the pipelined region by kernel (and, for stochastic sampling, by what the sampled wavefront was doing):
   python scripts/pcs_table.py <pc_sampling csv> <kernel_trace csv>
The column names of the beta csv are not documented: every column whose name suggests a dispatch / correlation id, an instruction type or a
stall reason is tabulated; the header is printed first."""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
cols = rows.fieldnames
print("columns:", cols)
kn = {}
if len(sys.argv) > 2 and sys.argv[2]:
    kt = csv.DictReader(open(sys.argv[2]))
    print("kernel trace columns:", kt.fieldnames)
    for r in kt:
        name = r.get("Kernel_Name") or r.get("kernel_name") or ""
        for c in ("Dispatch_Id", "dispatch_id", "Correlation_Id", "correlation_id"):
            if c in r:
                kn[(c.lower(), r[c])] = name.replace("void fslic::", "").split("(")[0][:48]
idc = [c for c in cols if c.lower() in ("dispatch_id", "correlation_id")]
cat = [c for c in cols if any(s in c.lower() for s in ("stall", "reason", "inst_type", "instruction_type", "wave_issued", "issued", "arb", "snapshot"))]
print("id columns", idc, "category columns", cat)
by_k = collections.Counter()
by_kc = collections.defaultdict(collections.Counter)
n = 0
for r in rows:
    n += 1
    name = "?"
    for c in idc:
        name = kn.get((c.lower(), r[c]), name)
        if name != "?":
            break
    by_k[name] += 1
    for c in cat:
        by_kc[name][(c, r[c])] += 1
print("samples", n)
for name, cnt in by_k.most_common(30):
    print("%-50s %8d %6.1f %%" % (name, cnt, 100.0 * cnt / max(n, 1)))
    for (c, val), m in by_kc[name].most_common(14):
        print("      %-34s %-30s %7d %5.1f %%" % (c, val, m, 100.0 * m / cnt))
