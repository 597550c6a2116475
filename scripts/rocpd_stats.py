#!/usr/bin/env python
"""Summaries of a rocprofv3 rocpd database (ROCm 7.2 writes <name>_results.db by default).

  python scripts/rocpd_stats.py stats    <db> [out.csv]   per-kernel calls / total / avg / min / max (us), like --stats
  python scripts/rocpd_stats.py burst    <db> [gap_us]    the last burst of kernels (bench.py's timed region): groups per stream, idle times, concurrency
  python scripts/rocpd_stats.py timeline <db> [marker]    dispatches of the LAST frame (from the last kernel whose name
                                                          contains `marker`, default rgb_to_lab), start/end/duration in us
"""
import sqlite3
import sys


def stats(db, out=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationUs,AverageUs,MinUs,MaxUs,Percentage"]
    for n, cnt, s, a, mn, mx in rows:
        n = n.split("(")[0]          # the kernel with its template arguments, without the parameter list (a long list pushed the numbers out of the clipped lines of round 4)
        lines.append('"%s",%d,%.3f,%.3f,%.3f,%.3f,%.2f' % (n, cnt, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    sys.stdout.write(text)


def timeline(db, marker="rgb_to_lab"):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, grid_y, workgroup_x, lds_size, vgpr_count, sgpr_count, stream_id "
                     "from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    s = idx[-1]
    t0 = rows[s][1]
    for n, st, en, gx, gy, wx, lds, vg, sg, sid in rows[s:]:
        print("%9.1f %9.1f %8.1f  %-64s blocks=%dx%d wg=%d lds=%d vgpr=%d sgpr=%d stream=%d"
              % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, n[:64], gx // max(wx, 1), gy, wx, lds, vg, sg, sid))


def overlap(db, frac=0.5):
    """Concurrency over the last `frac` of the trace (frac > 1: over the last `frac` microseconds): busy time (union of kernel
    intervals), summed kernel time, and how both split between the streams."""
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id from kernels order by start").fetchall()
    t_lo = rows[0][1] + (rows[-1][2] - rows[0][1]) * (1.0 - frac) if frac <= 1.0 else rows[-1][2] - frac * 1e3
    rows = [r for r in rows if r[1] >= t_lo]
    span = rows[-1][2] - rows[0][1]
    ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    busy = 0; depth = 0; last = ev[0][0]; hist = {}
    for t, d in ev:
        if depth > 0: busy += t - last
        hist[depth] = hist.get(depth, 0) + (t - last)
        depth += d; last = t
    tot = sum(r[2] - r[1] for r in rows)
    print("window %.1f us: busy (>=1 kernel running) %.1f us = %.1f %%, summed kernel time %.1f us = %.2f x window, %d kernels"
          % (span / 1e3, busy / 1e3, 100.0 * busy / span, tot / 1e3, tot / span, len(rows)))
    print("time at concurrency depth: " + "  ".join("%d: %.1f%%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
    per = {}
    for n, st, en, sid in rows:
        key = n.split("(")[0][-40:]
        per[key] = per.get(key, 0) + (en - st)
    for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:14]:
        print("  %-42s %8.1f us  %5.1f %% of summed" % (k, v / 1e3, 100.0 * v / tot))


def window(db, back_us=8000.0, us=1200.0):
    """All kernels of every stream in a window of `us` microseconds that starts `back_us` before the last kernel ends."""
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
    t0 = rows[-1][2] - back_us * 1e3
    prev_end = None
    for n, st, en, sid, qid in rows:
        if st < t0 or st > t0 + us * 1e3:
            continue
        gap = "" if prev_end is None else ("   idle %.1f" % ((st - prev_end) / 1e3) if st > prev_end else "")
        print("%9.1f %7.1f  s%-3d q%-3d %s%s" % ((st - t0) / 1e3, (en - st) / 1e3, sid, qid, n.split("(")[0][-36:], gap))
        prev_end = en if prev_end is None else max(prev_end, en)


def burst(db, gap_us=400.0):
    """The LAST burst of the trace (bench.py's timed region when it is the last thing the process does: everything after the last
    pause of more than `gap_us` between two kernel starts): per stream the groups (k_rgb_to_lab .. k_cca_relabel) with their
    start / end / busy time and the idle time before them, per hardware queue the streams it carries, and the concurrency in
    slices of a tenth of the burst."""
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
    i0 = len(rows) - 1
    while i0 > 0 and rows[i0][1] - rows[i0 - 1][1] < gap_us * 1e3:
        i0 -= 1
    rows = rows[i0:]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    print("burst: %d kernels, %.1f us from the first kernel's start to the last kernel's end" % (len(rows), (t1 - t0) / 1e3))
    streams = {}
    for r in rows:
        streams.setdefault(r[3], []).append(r)
    queues = {}
    for sid, rs in streams.items():
        queues.setdefault(rs[0][4], set()).add(sid)
    print("hardware queues -> streams: " + "; ".join("q%d: %s" % (q, " ".join("s%d" % s for s in sorted(ss))) for q, ss in sorted(queues.items())))
    for sid, rs in sorted(streams.items()):
        groups = []
        for r in rs:
            if "rgb_to_lab" in r[0] or not groups:
                groups.append([])
            groups[-1].append(r)
        prev_end = t0
        for g in groups:
            st, en = g[0][1], g[-1][2]
            busy = sum(r[2] - r[1] for r in g)
            print("  s%-3d q%-3d group of %2d kernels: start %8.1f end %8.1f (%7.1f us, kernels running %7.1f us = %4.1f %%), idle before it %7.1f us"
                  % (sid, g[0][4], len(g), (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, busy / 1e3, 100.0 * busy / max(en - st, 1), (st - prev_end) / 1e3))
            prev_end = en
    n = 10
    ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
    print("mean kernels running per tenth of the burst: ", end="")
    for k in range(n):
        lo, hi = t0 + (t1 - t0) * k / n, t0 + (t1 - t0) * (k + 1) / n
        depth = 0; last = t0; acc = 0.0
        for t, d in ev:
            a, b = max(last, lo), min(t, hi)
            if b > a:
                acc += depth * (b - a)
            depth += d; last = t
        print("%.2f " % (acc / (hi - lo)), end="")
    print()


if __name__ == "__main__":
    if sys.argv[1] == "burst":
        burst(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 400.0)
    elif sys.argv[1] == "window":
        window(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 8000.0, float(sys.argv[4]) if len(sys.argv) > 4 else 1200.0)
    elif sys.argv[1] == "overlap":
        overlap(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.5)
    elif sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        timeline(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rgb_to_lab")
