#!/usr/bin/env python
"""profiles/traffic.json from a tools/profile_bench.sh run (gpurun_out/prof_bench/summary.txt): HBM bytes per launch of the
headline kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 counts 64 B per 128 B read request, MI355X_MICROARCH.md), keyed to
the SHA-256 of the libuhdr_hip.so that was profiled -- run it in the tree whose .so went to the GPU box.
    python tools/update_traffic.py gpurun_out/prof_bench/summary.txt profiles/r04_bench_mapC_batch16_rocprofv3.txt"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

summary, committed_as = sys.argv[1], sys.argv[2]
fetch = write = None
nf = nw = 0
for line in open(summary):
    if "apply_quad_kernel<0, 2, 0," in line:
        m = re.search(r"(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+)\s+avg=([0-9.]+)", line)
        if m and m.group(1) == "FETCH_SIZE":
            fetch, nf = float(m.group(3)), int(m.group(2))
        elif m:
            write, nw = float(m.group(3)), int(m.group(2))
if fetch is None or write is None:
    sys.exit("no FETCH_SIZE / WRITE_SIZE rows for the headline kernel in " + summary)
rd, wr = int(round(2 * fetch * 1024)), int(round(write * 1024))
entry = {
    "traffic_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "library_sha256": bench.library_sha256(),
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_bench.sh) of `python bench.py --steps 10 --warmup 2 --no-extra --no-cpu "
              f"--no-config4` (headline launches only), {nf} / {nw} dispatches; FETCH_SIZE / WRITE_SIZE are in KiB, read bytes = 2 x FETCH_SIZE on gfx950 "
              f"(MI355X_MICROARCH.md); {committed_as}",
}
path = os.path.join(ROOT, "profiles", "traffic.json")
json.dump({"apply_quad_kernel<F16,RGBA8888,scale1>|16x3840x2160": entry}, open(path, "w"), indent=1)
print(json.dumps(entry, indent=1))
