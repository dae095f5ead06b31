#!/usr/bin/env python
"""Kernel statistics from a rocprofv3 rocpd database (ROCm 7: `rocprofv3 --kernel-trace --stats` writes *_results.db).
Prints the same columns as rocprofv3's kernel_stats.csv: name, calls, total ns, avg ns, %, min, max."""
import sqlite3
import sys


def main(path, out=None):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                     "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
    for n, k, t, a, mn, mx in rows:
        lines.append('"%s",%d,%d,%.1f,%.2f,%d,%d' % (n, k, t, a, 100.0 * t / tot, mn, mx))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
