#!/usr/bin/env python
"""Summaries of a rocprofv3 rocpd database (ROCm 7.2 writes <name>_results.db by default).

  python scripts/rocpd_stats.py stats    <db> [out.csv]   per-kernel calls / total / avg / min / max (us), like --stats
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


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        timeline(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "rgb_to_lab")
