#!/usr/bin/env python
"""Summarise tools/profile_tcc.sh: per counter group, each dispatch's duration next to its counter values.

The rocpd database of a `--kernel-trace --pmc` pass holds one row per dispatch in `kernels` (start / end) and one row
per (dispatch, counter) in `counters_collection`; the two are joined on the dispatch id.  Output: for every counter the
average over the dispatches, and -- to root-cause a duration spread -- the values of the fastest and the slowest third.
"""
import glob
import os
import sqlite3
import sys


def table_cols(cur, name):
    return [r[1] for r in cur.execute(f"pragma table_info({name})")]


def one(db):
    cur = sqlite3.connect(db).cursor()
    cc = table_cols(cur, "counters_collection")
    out = []
    if not cc:
        return ["  (no counters_collection table)"]
    dur, vals = {}, {}
    for did, name, cname, value, d in cur.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection"):
        if "apply_" not in name:
            continue
        dur[did] = d / 1e3
        vals.setdefault(cname, {}).setdefault(did, 0.0)
        vals[cname][did] += float(value)
    ids = sorted(dur, key=lambda i: dur[i])
    if not ids:
        return ["  (no apply_* dispatches)"]
    third = max(1, len(ids) // 3)
    fast, slow = ids[:third], ids[-third:]
    avg = lambda xs: sum(xs) / max(1, len(xs))
    out.append("  dispatches %d: duration us min %.1f median %.1f max %.1f; fastest third avg %.1f, slowest third avg %.1f"
               % (len(ids), dur[ids[0]], dur[ids[len(ids) // 2]], dur[ids[-1]], avg([dur[i] for i in fast]), avg([dur[i] for i in slow])))
    for c in sorted(vals):
        v = vals[c]
        out.append("  %-44s avg %16.1f   fastest third %16.1f   slowest third %16.1f" % (
            c, avg([v.get(i, 0.0) for i in ids]), avg([v.get(i, 0.0) for i in fast]), avg([v.get(i, 0.0) for i in slow])))
    return out


def main(d):
    for m in sorted(os.listdir(d)):
        md = os.path.join(d, m)
        if not os.path.isdir(md):
            continue
        print(f"## map {m}")
        t = os.path.join(md, "trace", "t_results.db")
        if os.path.exists(t):
            cur = sqlite3.connect(t).cursor()
            for r in cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name"):
                print("  trace only: %-90s calls=%d avg=%.0f min=%d max=%d ns" % (r[0][:90], r[1], r[2], r[3], r[4]))
        for p in sorted(glob.glob(os.path.join(md, "pmc*", "p_results.db")), key=lambda s: int(os.path.basename(os.path.dirname(s))[3:])):
            print(f"# {os.path.relpath(p, d)}")
            try:
                print("\n".join(one(p)))
            except Exception as e:  # keep going: one odd pass must not lose the others
                print(f"  (failed to read: {e})")


if __name__ == "__main__":
    main(sys.argv[1])
