#!/usr/bin/env python
"""profiles/r05_roofline_table.txt from a sectioned tools/read_prof.py summary (tools/profile_all.sh): one row per (case, kernel) --
rocprofv3 average duration, algorithmic bytes (every input read once, every output written once), HBM bytes from the
counters (2 x FETCH_SIZE + WRITE_SIZE, KiB; gfx950 tallies a 128-byte read request as 64 bytes: MI355X_MICROARCH.md) and the
fraction of the 8 TB/s peak the algorithmic bytes make of it.
    python tools/roofline_table.py gpurun_out/r05_prof/summary.txt [gpurun_out/prof_bench/summary.txt] > profiles/r05_roofline_table.txt"""
import re
import sys

PX4, PX8 = 3840 * 2160, 7680 * 4320
ROWS = [  # case, kernel name fragment, algorithmic bytes per launch, what the bytes are
    ("8kC", "apply_quad_kernel<0, 2, 0, 0, 0>", 13.5 * PX8, "8K 4:2:0 + RGBA8888 map -> F16 (1.5 + 4 + 8 B/px)"),
    ("8kB", "apply_quad_kernel<0, 1, 0, 0, 0>", 12.5 * PX8, "8K 4:2:0 + RGB888 map -> F16 (1.5 + 3 + 8)"),
    ("8kA", "apply_quad_kernel_s96<0, 0, 1, 0, 0>", 9.5625 * PX8, "8K 4:2:0 + Y400 map at scale 4 -> F16 (1.5 + 1/16 + 8)"),
    ("4kAhlg", "apply_quad_kernel_s96<1, 0, 1, 0, 0>", 5.5625 * PX4, "4K 4:2:0 + Y400 s4 -> HLG RGBA1010102 (1.5 + 1/16 + 4)"),
    ("b32hlg", "apply_quad_kernel_s96<1, 0, 1, 0, 0>", 5.5625 * PX4 * 16, "config 5: 32 frames per call = two launches of 16 frames"),
    ("tm4k", "tonemap_p010_kernel", 4.5 * PX4, "4K P010 -> 4:2:0 (3 + 1.5)"),
    ("gen4k", "generate_quad_kernel", 16.5 * PX4, "pass 1: 4.5 in + 12 out (ratios)"),
    ("gen4k", "affine_wide_kernel<3>", 15.0 * PX4, "pass 2: 12 in + 3 out"),
    ("api1f", "generate_quad_kernel", 16.5 * PX4, "fused API-1 chain, pass 1"),
    ("api1f", "map_blocks_kernel<3>", 18.0 * PX4, "pass 2 + rgb->ycc + 3 FDCT: 12 in + 6 out"),
    ("api1f", "base_blocks_kernel", 4.5 * PX4, "convertYuv + 3 FDCT: 1.5 in + 3 out"),
    ("api1f8k", "generate_quad_kernel", 16.5 * PX8, "the same chain at 8K"),
    ("api1f8k", "map_blocks_kernel<3>", 18.0 * PX8, ""),
    ("api1f8k", "base_blocks_kernel", 4.5 * PX8, ""),
    ("api0f", "encode_api0_fused4_kernel", 10.0 * PX8, "8K RGBA1010102 -> base 4:4:4 + 3-channel map: 4 in + 3 + 3 out"),
    ("fdct4k", "fdct_quant_kernel", 3.0 * PX4, "4K luma plane: 1 in + 2 out"),
    ("idct4k", "idct_dequant_kernel", 3.0 * PX4, "4K luma plane: 2 in + 1 out"),
    ("huff4k", "hyp_pass0_kernel", None, "entropy decode of the 4K base image (marker-less file): issue / latency bound, not HBM"),
    ("huff4k", "hyp_pass1q_kernel", None, ""),
    ("huff4k", "hyp_straggler_kernel", None, ""),
    ("huff4k", "sync_write2_kernel", None, ""),
    ("huff4k", "coef_place_kernel", 2 * 3.0 * PX4, "scan-order scratch -> JBLOCK arrays: 3 in + 3 out B/px"),
    ("huff4k", "huff_stream_kernel<1, 0>", None, "entropy encode (marker-less), lengths pass"),
    ("huff4k", "huff_stream_kernel<16, 1>", None, "... emit pass"),
]


def parse(path):
    dur, cnt = {}, {}
    for line in open(path, errors="replace"):
        m = re.match(r"(\d+):(\S+)\s+(.*?)\s+(\S*)\s+calls=(\d+) avg=([0-9.]+) min=(\d+) max=(\d+)", line)
        if m:
            dur.setdefault((m.group(2), m.group(3).strip()), []).append((int(m.group(5)), float(m.group(6)), int(m.group(7)), m.group(4)))
            continue
        m = re.match(r"(\d+):(\S+)\s+(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=(\d+) avg=([0-9.]+)", line)
        if m:
            cnt[(m.group(2), m.group(3).strip(), m.group(4))] = float(m.group(6))
    return dur, cnt


def main():
    dur, cnt = parse(sys.argv[1])
    print("# per (case, kernel): rocprofv3 kernel-trace average duration | algorithmic bytes | counter bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB) | algorithmic / duration / 8 TB/s")
    print("# source: " + sys.argv[1] + " (tools/profile_all.sh: one process, sections cut at uhdr_profile_mark_kernel; the dominant grid of each (case, kernel))")
    print("%-9s %-42s %6s %9s %14s %14s %7s %7s  %s" % ("case", "kernel", "calls", "avg us", "algorithmic B", "counter B", "ctr/alg", "frac", "bytes are"))
    for case, frag, algo, what in ROWS:
        best = None
        for (c, name), rows in dur.items():
            if c == case and frag in name:
                r = max(rows, key=lambda x: x[0])  # the grid launched most often (warm-up shapes aside)
                best = (name, r)
        if not best:
            print("%-9s %-42s   (not in this trace)" % (case, frag))
            continue
        name, (calls, avg, mn, grid) = best
        f = w = None
        for (c, n2, which), v in cnt.items():
            if c == case and frag[:40] in n2 or (c == case and n2[:30] in name and frag.split("<")[0] in n2):
                if which == "FETCH_SIZE":
                    f = v
                else:
                    w = v
        ctr = (2 * f + w) * 1024 if f is not None and w is not None else None
        print("%-9s %-42s %6d %9.1f %14s %14s %7s %7s  %s" % (
            case, frag[:42], calls, avg / 1e3, "%d" % algo if algo else "-", "%d" % ctr if ctr else "-",
            "%.3f" % (ctr / algo) if ctr and algo else "-", "%.3f" % (algo / (avg * 1e-9) / 8e12) if algo else "-", what))
    if len(sys.argv) > 2:
        d2, c2 = parse(sys.argv[2])
        for (c, name), rows in d2.items():
            if "apply_quad_kernel<0, 2, 0, 0, 0>" in name:
                calls, avg, mn, grid = max(rows, key=lambda x: x[0])
                algo = 13.5 * PX4 * 16
                print("%-9s %-42s %6d %9.1f %14d %14s %7s %7.3f  %s" % ("headline", "apply_quad_kernel<0, 2, 0, 0, 0> x16", calls, avg / 1e3, algo, "-", "-", algo / (avg * 1e-9) / 8e12,
                                                                 "bench.py's launch: 16 x 4K map C (" + sys.argv[2] + ")"))


main()
