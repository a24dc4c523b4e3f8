#!/bin/bash
# SQ counters of the kernels of the API-1 round trip (tools/roundtrip_once.py, one scan after the other): two --pmc passes, never with a trace,
# per kernel name and grid: the value per dispatch.   tools/profile_roundtrip_pmc.sh [4k|8k]  ->  stdout
R=${GRAFT_REPO_ROOT:-/root/repo}
SZ=${1:-4k}
OUT=/tmp/pmc_rt
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for G in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" ; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $G -d $OUT/pmc$i -o p -- python $R/tools/roundtrip_once.py 3 $SZ seq > $OUT/pmc$i.log 2>&1 || echo "pmc group $i failed"
done
python - <<PY
import sqlite3, glob, collections
for i in (1, 2):
    dbs = glob.glob("$OUT/pmc%d/**/*.db" % i, recursive=True)
    if not dbs:
        print("no database for pass", i); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select * from counters_collection").fetchall()
    ik, ic, iv = cols.index("kernel_name"), cols.index("counter_name"), cols.index("value")
    ig = cols.index("grid_size") if "grid_size" in cols else (cols.index("grid_size_x") if "grid_size_x" in cols else None)
    acc = collections.OrderedDict()
    for r in rows:
        name = r[ik].replace("uhdr::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        if not any(t in name for t in ("hyp_", "sync_write2", "huff_stream_kernel", "coef_place")):
            continue
        key = (name[:40], r[ig] if ig is not None else 0, r[ic])
        a = acc.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += r[iv]
    print("== pass", i)
    last = None
    for (name, g, c), (n, v) in sorted(acc.items()):
        if (name, g) != last:
            print(f"{name} grid {g}  ({n} dispatches)")
            last = (name, g)
        print(f"    {c:24s} {v / n:16.0f} per dispatch")
PY
