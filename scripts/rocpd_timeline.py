#!/usr/bin/env python
"""Timeline of ONE step from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace`): every kernel between two consecutive launches of
`delimiter` (default pack_weights_kernel: once per training step), in start order, with its queue, start offset and duration; then the
busy time per queue, the overlap between queues and the tail (time after the last kernel of the busiest queue).
    python scripts/rocpd_timeline.py results.db [delimiter] [which step from the end, default 2] > profiles/...txt"""
import sqlite3
import sys


def main(path, delim="pack_weights_kernel", back=2):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    extra = [k for k in ("queue_id", "stream_id") if k in cols]
    rows = c.execute("select name, start, end%s from kernels order by start" % "".join(", " + k for k in extra)).fetchall()
    marks = [i for i, r in enumerate(rows) if delim in r[0]]
    if len(marks) < back + 1:
        print("columns:", cols)
        print("not enough '%s' launches (%d)" % (delim, len(marks)))
        return
    a, b = marks[-back - 1], marks[-back]
    win = rows[a:b]
    t0 = win[0][1]
    print("# one step: %d kernels, %.1f us from the first start to the last end; columns of the view: %s" % (len(win), (max(r[2] for r in win) - t0) / 1e3, extra))
    for r in win:
        print("%9.1f us  +%7.1f us  q=%s  %s" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, "/".join(str(v) for v in r[3:]), r[0][:110]))
    if extra:
        qs = sorted(set(r[3:] for r in win))
        for q in qs:
            ks = [r for r in win if r[3:] == q]
            print("# queue %s: %d kernels, busy %.1f us, first start %.1f us, last end %.1f us" % (q, len(ks), sum(k[2] - k[1] for k in ks) / 1e3, (ks[0][1] - t0) / 1e3, (max(k[2] for k in ks) - t0) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3]), *([int(sys.argv[3])] if len(sys.argv) > 3 else []))
