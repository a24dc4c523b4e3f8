#!/usr/bin/env python
"""Kernel timeline of ONE iteration out of a tools/profile_roundtrip.sh trace: start / end (us, relative), queue, kernel -- to see which
launches overlap and where the device idles.   python tools/timeline.py gpurun_out/prof_rt_4ktwo [iteration]"""
import glob
import sqlite3
import sys

d = sys.argv[1]
it = int(sys.argv[2]) if len(sys.argv) > 2 else 5
db = glob.glob(d + "/t/**/*.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else None
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x{', d.' + qcol if qcol else ''} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if "profile_mark" in r[0]]
sel = rows[marks[-2] + 1: marks[-1]]
# iterations start at generate_quad_kernel (the first kernel of enc())
starts = [i for i, r in enumerate(sel) if "generate_quad_kernel" in r[0]]
a, b = starts[it], (starts[it + 1] if it + 1 < len(starts) else len(sel))
t0 = sel[a][1]
busy_end = t0
idle = 0
for r in sel[a:b]:
    name = r[0].split("(")[0].replace("uhdr::(anonymous namespace)::", "").replace("void ", "")
    if name.startswith("_ZN4uhdr12_GLOBAL__N_1"):
        name = name[len("_ZN4uhdr12_GLOBAL__N_1"):]
    gap = (r[1] - busy_end) / 1e3
    if gap > 0:
        idle += gap
    busy_end = max(busy_end, r[2])
    print(f"{(r[1] - t0) / 1e3:9.1f} {(r[2] - t0) / 1e3:9.1f}  {(r[2] - r[1]) / 1e3:7.1f} us  q{r[5] if qcol else 0:<3} {'IDLE %5.1f' % gap if gap > 1.0 else '          '}  {name[:70]} [{r[3]}x{r[4]}]")
print(f"iteration span {(busy_end - t0) / 1e3:.1f} us, device idle {idle:.1f} us")
