"""One decode of each scan of bench.py's 4K round trip with UHDR_HIP_HUFF_DEBUG=1: merges per level, paths handed to the stragglers."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import bench
from libultrahdr_amd.ultrahdr import Context, UltraHdr

ctx = Context(0)
u = UltraHdr(ctx=ctx)
w, h = (7680, 4320) if len(sys.argv) > 1 and sys.argv[1] == "8k" else (3840, 2160)
enc, dec, box = bench.make_roundtrip(ctx, u, "cuda:0", w, h)
S420, S444 = [(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3
os.environ["UHDR_HIP_HUFF_DEBUG"] = "1"
for name, data, shp, S in (("base", box["sb"], box["shp_b"], S420), ("map", box["sm"], box["shp_m"], S444)):
    print("==", name, int(data.numel()), "bytes", file=sys.stderr, flush=True)
    u.huffman_decode(data, shp, w, h, S, 0)
    ctx.synchronize()
