#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$PWD/gpurun_out/prof_dec
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ri in 0 10; do
  timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/ri$ri -o t -- python $R/tools/decode_once.py $ri > $OUT/ri$ri.log 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob("$OUT/ri$ri/**/*.db", recursive=True)
con = sqlite3.connect(db[0])
cur = con.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = cur.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
print("== ri $ri")
tot = 0
for n, c, a, m in rows:
    if c >= 10:
        nm = n.split("(")[0][-60:]
        per = a * c / 12
        tot += per
        print(f"{nm:60s} calls {c:4d} avg {a/1e3:8.1f} us  per decode {per/1e3:8.1f} us")
print("sum per decode", round(tot / 1e3, 1))
PY
done
