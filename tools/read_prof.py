#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs of tools/profile.sh into plain text (for profiles/)."""
import glob
import os
import sqlite3
import sys


def kernels(db):
    cur = sqlite3.connect(db).cursor()
    return cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc").fetchall()


def counters(db):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    return cols, cur.execute("select * from counters_collection").fetchall()


def main(d):
    out = []
    t = os.path.join(d, "trace", "t_results.db")
    if os.path.exists(t):
        out.append("# kernel trace (rocprofv3 --kernel-trace --stats): name, calls, avg_ns, min_ns, max_ns, total_ns")
        for r in kernels(t):
            out.append("%-110s calls=%d avg=%.0f min=%d max=%d total=%d" % (r[0][:110], r[1], r[2], r[3], r[4], r[5]))
    for p in sorted(glob.glob(os.path.join(d, "pmc*", "p_results.db"))):
        cols, rows = counters(p)
        if not rows:
            out.append(f"# {os.path.relpath(p, d)}: no counter rows; columns {cols}")
            continue
        ik = cols.index("kernel_name") if "kernel_name" in cols else None
        ic, iv = cols.index("counter_name"), cols.index("value")
        agg = {}
        for r in rows:
            k = (r[ik][:60] if ik is not None else "?", r[ic])
            a = agg.setdefault(k, [0, 0.0])
            a[0] += 1
            a[1] += float(r[iv])
        out.append(f"# {os.path.relpath(p, d)}: per-dispatch average of each counter")
        for (k, c), (n, s) in sorted(agg.items()):
            out.append("%-62s %-28s n=%d avg=%.1f" % (k, c, n, s / n))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
