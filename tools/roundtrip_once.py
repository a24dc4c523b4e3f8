"""The API-1 4K (or 8K) round trip of bench.py's headline, N times, for a profiler: python tools/roundtrip_once.py [n] [4k|8k] [seq]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import bench
from libultrahdr_amd.ultrahdr import Context, UltraHdr

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
w, h = (7680, 4320) if len(sys.argv) > 2 and sys.argv[2] == "8k" else (3840, 2160)
two = not (len(sys.argv) > 3 and sys.argv[3] == "seq")
ctx = Context(0)
u = UltraHdr(ctx=ctx)
enc, dec, box = bench.make_roundtrip(ctx, u, "cuda:0", w, h)
for _ in range(3):
    enc(two)
    dec(two)
ctx.synchronize()
ctx.lib.uhdr_hip_profile_mark(ctx.handle)
for _ in range(n):
    enc(two)
    dec(two)
ctx.synchronize()
ctx.lib.uhdr_hip_profile_mark(ctx.handle)
print("scan bytes", box["nb"], box["nm"])
