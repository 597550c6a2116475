"""Summarise the [fslic host] lines of a bench run made with FSLIC_HOST_TIMING=1 (stderr on stdin)."""
import re, sys, collections
acc = collections.defaultdict(list)
for line in sys.stdin:
    if "[fslic host]" not in line:
        continue
    if "group_begin" in line:
        acc["group_begin"].append(float(re.search(r": ([\d.]+) us", line).group(1)))
    else:
        for name, val in re.findall(r"(sync wait|events|host top-K \(\d+ frames\)|write-back|device) ([\d.]+) us", line):
            acc[re.sub(r" \(\d+ frames\)", "", name)].append(float(val))
        m = re.search(r"host top-K \((\d+) frames\)", line)
        if m:
            acc["tie frames"].append(float(m.group(1)))
for k, v in acc.items():
    v2 = sorted(v)
    print("%-12s n=%4d mean %8.1f  p50 %8.1f  p90 %8.1f" % (k, len(v), sum(v) / len(v), v2[len(v) // 2], v2[len(v) * 9 // 10]))
