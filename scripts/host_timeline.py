"""Per-slot accounting of a bench run made with FSLIC_HOST_TIMING=1 (stderr on stdin): where does a slot's cycle go?"""
import re, sys, collections
ev = collections.defaultdict(list)
for line in sys.stdin:
    m = re.search(r"slot (\d+) begin ([\d.]+) \.\. ([\d.]+)", line)
    if m:
        ev[int(m.group(1))].append(("begin", float(m.group(2)), float(m.group(3))))
    m = re.search(r"slot (\d+) at ([\d.]+) \.\. ([\d.]+): sync wait ([\d.]+) us, events [\d.]+ us, host top-K \((\d+) frames\) ([\d.]+) us, write-back ([\d.]+) us \(device ([\d.]+) us\)", line)
    if m:
        ev[int(m.group(1))].append(("finish", float(m.group(2)), float(m.group(3)), float(m.group(4)), float(m.group(6)), float(m.group(8))))
tot = collections.Counter(); n = 0
for s, L in ev.items():
    L.sort(key=lambda t: t[1])
    for a, b in zip(L, L[1:]):
        if a[0] == "begin" and b[0] == "finish":
            tot["begin (host enqueue)"] += a[2] - a[1]
            tot["begin end -> finish start (worker idle, caller's turn)"] += b[1] - a[2]
            tot["finish: waiting for the GPU"] += b[3]
            tot["finish: tie path (host sort + fix-up round trip)"] += b[4]
            tot["finish: rest (events, write-back)"] += (b[2] - b[1]) - b[3] - b[4]
            tot["device time of the group (events)"] += b[5]
            n += 1
        elif a[0] == "finish" and b[0] == "begin":
            tot["finish end -> next begin (collect, resubmit, wake-up)"] += b[1] - a[2]
cyc = sum(v for k, v in tot.items() if not k.startswith("device"))
print("groups: %d, slots: %d, mean slot cycle %.1f us" % (n, len(ev), cyc / max(n, 1)))
for k, v in tot.items():
    print("  %-62s %8.1f us" % (k, v / max(n, 1)))
