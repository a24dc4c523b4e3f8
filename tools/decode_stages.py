#!/usr/bin/env python
"""Where the time of uhdr_hip_jpeg_decode_scan goes for the two JPEGs of a 4K UltraHDR file made by the facade (base image
4:2:0, three-channel full-resolution gain map 4:4:4): wall time of the call with and without the download of the samples
(uhdr_hip_resident_lazy), and the HIP-event time of its kernel families."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from libultrahdr_amd import capi as A
from libultrahdr_amd import facade as FA
from libultrahdr_amd import synth
from libultrahdr_amd.ultrahdr import Context, UltraHdr

w, h = 3840, 2160
hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
sdr = synth.make_sdr_yuv420(w, h)
jpg = FA.encode(hdr, sdr, gpu=True)
cut = jpg.rfind(b"\xff\xd8\xff")
files = {"base": jpg[:cut], "map": jpg[cut:]}
ctx = Context(0)
u = UltraHdr(ctx=ctx)
for name, data in files.items():
    hd = u.jpeg_parse(data)
    rgb = 3 if name == "map" else 0
    outs = None
    for lazy in (0, 1):
        u.lib.uhdr_hip_resident_begin(ctx.handle)
        u.lib.uhdr_hip_resident_lazy(ctx.handle, lazy)
        outs = u.jpeg_decode(data, rgb, outs=outs if outs is None or isinstance(outs, list) else [outs])
        walls = []
        ctx.profile(True)
        ctx.profile_read(None, reset=True)
        for _ in range(5):
            t0 = time.perf_counter()
            u.jpeg_decode(data, rgb, outs=outs if isinstance(outs, list) else [outs])
            walls.append((time.perf_counter() - t0) * 1e3)
        fam = {f: ctx.profile_read(f, reset=False) for f in ("huffman_decode", "idct_dequant", "jpeg_color")}
        ctx.profile_read(None, reset=True)
        ctx.profile(False)
        u.lib.uhdr_hip_resident_end(ctx.handle)
        print(f"{name:4s} {len(data):9d} B  lazy={lazy}  wall ms {' '.join(f'{x:.2f}' for x in walls)}  " +
              "  ".join(f"{f} {ms / 5:.3f} ms ({n // 5} launches)" for f, (n, ms) in fam.items()), flush=True)
st = A.Stats()
u.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st))
print({n: getattr(st, n) for n, _ in st._fields_})
