"""Decode time of the 4K three-channel gain-map scan (and the base scan) of bench.py's round trip for several settings of the
hypothesis scheme: python tools/r06_sweep_map.py   (env knobs are read per call by the library)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import bench
from libultrahdr_amd.ultrahdr import Context, UltraHdr

ctx = Context(0)
u = UltraHdr(ctx=ctx)
w, h = 3840, 2160
enc, dec, box = bench.make_roundtrip(ctx, u, "cuda:0", w, h)
S420, S444 = [(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3


def t_dec(which, n=8):
    data, shp, S = (box["sb"], box["shp_b"], S420) if which == "base" else (box["sm"], box["shp_m"], S444)
    fn = lambda: u.huffman_decode(data, shp, w, h, S, 0)
    for _ in range(3):
        fn()
    ctx.synchronize()
    bench.clock_ramp(ctx, fn, seconds=0.4)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    ctx.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


SETS = [{}, {"UHDR_HIP_HUFF_STRAG_MUL": "2"}, {"UHDR_HIP_HUFF_STRAG_MUL": "4"}, {"UHDR_HIP_HUFF_MAIN_LEVELS": "1"}, {"UHDR_HIP_HUFF_MAIN_LEVELS": "2"},
        {"UHDR_HIP_HUFF_MAIN_LEVELS": "3"}, {"UHDR_HIP_HUFF_WRITE": "1"}]
if len(sys.argv) > 1:
    SETS = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]]
for which in ("map", "base"):
    for env in SETS:
        for k, v in env.items():
            os.environ[k] = v
        try:
            us = t_dec(which)
        finally:
            for k in env:
                del os.environ[k]
        print(f"{which:5s} {str(env):70s} {us:8.1f} us per decode (wall, synchronous call)", flush=True)
