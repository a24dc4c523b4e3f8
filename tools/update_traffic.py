#!/usr/bin/env python
"""profiles/traffic.json from the rocprofv3 counter passes of one GPU call, keyed to the SHA-256 of the libuhdr_hip.so that was
profiled (run it in the tree whose .so went to the GPU box): HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 tallies
a 128-byte read request as 64 bytes, MI355X_MICROARCH.md).
  * the roofline kernel of bench.py's line (applyGainMap, ONE 8K frame per launch, map C) from a tools/profile_bench.sh run
    (gpurun_out/prof_bench/summary.txt);
  * round 5: the encode chains bench.py reports (encode.api1_4k / api1_8k / config3_api0_8k) from the sectioned summary of a
    tools/profile_all.sh run (cases api1f, api1f8k, api0f), per kernel family as bench.py names them.
    python tools/update_traffic.py gpurun_out/prof_bench/summary.txt profiles/r05_bench_mapC_batch16_rocprofv3.txt [gpurun_out/r05_prof/summary.txt profiles/r05_prof_all_summary.txt]"""
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
sha = bench.library_sha256()
rd, wr = int(round(2 * fetch * 1024)), int(round(write * 1024))
out = {"apply_quad_kernel<F16,RGBA8888,scale1>|1x7680x4320": {
    "traffic_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "library_sha256": sha,
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_bench.sh) of `python bench.py --steps 10 --warmup 2 --no-extra --no-cpu "
              f"--no-config4` (UHDR_BENCH_HEADLINE_ONLY: the round-trip steps + the roofline kernel's 8K launches), {nf} / {nw} dispatches; FETCH_SIZE / WRITE_SIZE are in KiB, read bytes = 2 x FETCH_SIZE on gfx950 "
              f"(MI355X_MICROARCH.md); {committed_as}"}}
if len(sys.argv) > 4:
    sect, sect_as = sys.argv[3], sys.argv[4]
    FAM = {"generate_quad_kernel": "generate_gainmap", "minmax_table_kernel": "generate_gainmap", "map_blocks_kernel": "fdct_quant", "base_blocks_kernel": "fdct_quant",
           "encode_api0_fused4_kernel": "encode_api0_fused"}
    CASE = {"api1f": "encode.api1_4k", "api1f8k": "encode.api1_8k", "api0f": "encode.config3_api0_8k"}
    acc = {}
    for line in open(sect, errors="replace"):
        m = re.match(r"\d+:(\S+)\s+(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+) avg=([0-9.]+)", line)
        if not m or m.group(1) not in CASE:
            continue
        fam = next((f for k, f in FAM.items() if k in m.group(2)), None)
        if fam is None:
            continue
        a = acc.setdefault(CASE[m.group(1)], {}).setdefault(fam, [0.0, 0.0])
        a[0 if m.group(3) == "FETCH_SIZE" else 1] += float(m.group(5))
    for key, fams in acc.items():
        per = {f: int(round((2 * v[0] + v[1]) * 1024)) for f, v in fams.items()}
        out[key] = {"traffic_bytes_per_launch": sum(per.values()), "per_family": per, "library_sha256": sha,
                    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of tools/qbench.py (tools/profile_all.sh, per-case sections), summed over the chain's kernels; {sect_as}"}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
