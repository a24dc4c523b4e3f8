#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs of tools/profile_all.sh into plain text (for profiles/).

Round 5: per (case, kernel).  tools/qbench.py launches the empty kernel `uhdr_profile_mark_kernel` in front of every case and
prints "MARK <n> <case>"; the trace (ordered by start time) and the counter rows (ordered by dispatch) are cut at those
launches, so that one process that runs 4K and 8K shapes of the same kernel no longer averages them under one name."""
import glob
import os
import re
import sqlite3
import sys

MARK = "uhdr_profile_mark_kernel"


def case_names(log):
    names = {}
    try:
        for line in open(log, errors="replace"):
            m = re.match(r"MARK (\d+) (\S+)", line)
            if m:
                names[int(m.group(1))] = m.group(2)
    except OSError:
        pass
    return names


def cols_of(cur, table):
    return [r[1] for r in cur.execute(f"pragma table_info({table})")]


def kernels_by_section(db):
    cur = sqlite3.connect(db).cursor()
    cols = cols_of(cur, "kernels")
    extra = [c for c in ("grid_x", "grid_size_x", "grid_size", "workgroup_x", "workgroup_size_x", "workgroup_size") if c in cols]
    rows = cur.execute("select name, start, end" + "".join(", " + c for c in extra) + " from kernels order by start").fetchall()
    sec, out = -1, {}
    for r in rows:
        if MARK in r[0]:
            sec += 1
            continue
        key = (sec, r[0], tuple(r[3:]))
        a = out.setdefault(key, [0, 0, 1 << 62, 0])
        d = r[2] - r[1]
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    return extra, out


def counters_by_section(db):
    cur = sqlite3.connect(db).cursor()
    cols = cols_of(cur, "counters_collection")
    order = next((c for c in ("dispatch_id", "id", "start", "timestamp") if c in cols), None)
    rows = cur.execute("select * from counters_collection" + (f" order by {order}" if order else "")).fetchall()
    if not rows:
        return cols, None
    ik = cols.index("kernel_name") if "kernel_name" in cols else None
    ic, iv = cols.index("counter_name"), cols.index("value")
    idd = cols.index("dispatch_id") if "dispatch_id" in cols else None
    sec, seen_mark, agg = -1, set(), {}
    for r in rows:
        name = r[ik] if ik is not None else "?"
        if MARK in name:
            d = r[idd] if idd is not None else len(seen_mark)
            if d not in seen_mark:  # (one row per counter and dispatch: count a marker dispatch once)
                seen_mark.add(d)
                sec += 1
            continue
        k = (sec, name, r[ic])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r[iv])
    return cols, agg


def main(d):
    out = []
    names = case_names(os.path.join(d, "trace.log"))
    label = lambda s: f"{s}:{names.get(s, '?')}" if s >= 0 else "-"
    t = os.path.join(d, "trace", "t_results.db")
    if os.path.exists(t):
        extra, ks = kernels_by_section(t)
        out.append("# kernel trace (rocprofv3 --kernel-trace), per (case, kernel" + "".join(", " + c for c in extra) + "): calls, avg_ns, min_ns, max_ns")
        for (sec, name, ex), a in sorted(ks.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
            out.append("%-14s %-96s %-14s calls=%d avg=%.0f min=%d max=%d" % (label(sec), name[:96], ",".join(str(x) for x in ex), a[0], a[1] / a[0], a[2], a[3]))
    for p in sorted(glob.glob(os.path.join(d, "pmc*", "p_results.db"))):
        cols, agg = counters_by_section(p)
        if agg is None:
            out.append(f"# {os.path.relpath(p, d)}: no counter rows; columns {cols}")
            continue
        out.append(f"# {os.path.relpath(p, d)}: per-dispatch average of each counter, per (case, kernel)   [columns: {', '.join(cols)}]")
        for (sec, k, c), (n, s) in sorted(agg.items()):
            out.append("%-14s %-62s %-28s n=%d avg=%.1f" % (label(sec), k[:62], c, n, s / n))
    print("\n".join(out))


if __name__ == "__main__":
    main(sys.argv[1])
