#!/bin/bash
# Per-kernel durations of the API-1 round trip (bench.py's headline step): rocprofv3 kernel trace of tools/roundtrip_once.py,
# kernels between the two uhdr_profile_mark_kernel launches, averaged per round trip.   tools/profile_roundtrip.sh [n] [4k|8k] [seq]
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-12}; SZ=${2:-4k}; MODE=${3:-two}
OUT=$PWD/gpurun_out/prof_rt_$SZ$MODE
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/t -o t -- python $R/tools/roundtrip_once.py $N $SZ $MODE > $OUT/run.log 2>&1
python - <<PY
import sqlite3, glob, collections
db = glob.glob("$OUT/t/**/*.db", recursive=True)
con = sqlite3.connect(db[0]); cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.workgroup_size_x from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if "profile_mark" in r[0]]
sel = rows[marks[-2] + 1: marks[-1]] if len(marks) >= 2 else rows
acc = collections.OrderedDict()
for name, st, en, g, wg in sel:
    key = (name.split("(")[0].replace("uhdr::(anonymous namespace)::", "").replace("void ", "")[:64], g, wg)
    a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += en - st
n = $N
span = (sel[-1][2] - sel[0][1]) / n / 1e3
print(f"== API-1 $SZ round trip ($MODE), {n} iterations: {span:.1f} us of device timeline per round trip, kernels summed {sum(a[1] for a in acc.values()) / n / 1e3:.1f} us")
for (name, g, wg), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:64s} grid {g:8d} x {wg:4d}  calls/rt {c / n:5.1f}  avg {t / c / 1e3:8.1f} us  per round trip {t / n / 1e3:8.1f} us")
PY
