#!/usr/bin/env python
"""Entropy-decoder experiments in one process (round 5): the base image (4:2:0) and the three-channel gain map (4:4:4) of a 4K
UltraHDR file written through the facade, decoded with
  * the round-4 form (all overflow levels in lockstep, write pass form 1),
  * write pass form 2, and 1 / 2 / 3 lockstep levels before the stragglers get a wave each,
  * pass 1 cut off after 1 .. 7 levels (its time per level; the true path is lost below ~6, which is not the point here).
UHDR_HIP_HUFF_DEBUG=1 prints the per-kernel times of every decode on stderr."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench as B
from libultrahdr_amd import capi as A
from libultrahdr_amd import facade as FA
from libultrahdr_amd import synth
from libultrahdr_amd.ultrahdr import Context, UltraHdr

ctx = Context(0)
u = UltraHdr(ctx=ctx)
w, h = 3840, 2160
jpg = FA.encode(synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG), synth.make_sdr_yuv420(w, h), gpu=True)
cut = jpg.rfind(b"\xff\xd8\xff")
streams = {}
for name, f in (("base", jpg[:cut]), ("map3ch", jpg[cut:])):
    hd = u.jpeg_parse(f)
    sc = hd.scan
    nc = sc.num_components
    data = torch.from_numpy(np.frombuffer(f, dtype=np.uint8)[hd.scan_offset: hd.scan_offset + hd.scan_bytes].copy()).to("cuda:0")
    bits = np.frombuffer(hd.tables.bits, dtype=np.uint8).reshape(4, 17)
    vals = np.frombuffer(hd.tables.vals, dtype=np.uint8).reshape(4, 256)
    shp = [(sc.blocks_h[c], sc.blocks_w[c]) for c in range(nc)]
    smp = [(sc.h_samp[c], sc.v_samp[c]) for c in range(nc)]
    streams[name] = (data, shp, sc.w, sc.h, smp, (bits, vals), int(hd.scan_bytes))


def run(name, env, iters=5, ref=None):
    data, shp, sw, sh, smp, tabs, nbytes = streams[name]
    for k in ("UHDR_HIP_HUFF_WRITE", "UHDR_HIP_HUFF_MAIN_LEVELS", "UHDR_HIP_HUFF_LEVELS", "UHDR_HIP_HUFF_SUB_BITS", "UHDR_HIP_HUFF_DEBUG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    fn = lambda: u.huffman_decode(data, shp, sw, sh, smp, 0, tables=tabs)
    out = fn()
    same = None if ref is None else all(bool(torch.equal(a_, b_)) for a_, b_ in zip(out, ref))
    ms = sorted(B.time_kernel(ctx, fn, iters=iters, warm=2) for _ in range(3))[1]
    os.environ["UHDR_HIP_HUFF_DEBUG"] = "1"
    sys.stderr.write(f"--- {name} {env}\n")
    sys.stderr.flush()
    fn()
    os.environ.pop("UHDR_HIP_HUFF_DEBUG", None)
    print(f"{name:7s} {nbytes:8d} B  {str(env):70s} {ms * 1e3:8.1f} us" + ("" if same is None else f"   coefficients == round-4 form: {same}"), flush=True)
    return out


for name in streams:
    ref = run(name, {"UHDR_HIP_HUFF_WRITE": "1", "UHDR_HIP_HUFF_MAIN_LEVELS": "0"})
    run(name, {"UHDR_HIP_HUFF_WRITE": "2", "UHDR_HIP_HUFF_MAIN_LEVELS": "0"}, ref=ref)
    for ml in (1, 2, 3, 4):
        run(name, {"UHDR_HIP_HUFF_WRITE": "2", "UHDR_HIP_HUFF_MAIN_LEVELS": str(ml)}, ref=ref)
    run(name, {"UHDR_HIP_HUFF_WRITE": "1", "UHDR_HIP_HUFF_MAIN_LEVELS": "2"}, ref=ref)
# pass 1's time per lockstep level (base image, 512-bit subsequences): the debug line's "pass1" figure at 1 .. 7 levels
for lv in (1, 2, 3, 4, 5, 6, 7):
    run("base", {"UHDR_HIP_HUFF_WRITE": "1", "UHDR_HIP_HUFF_MAIN_LEVELS": "0", "UHDR_HIP_HUFF_LEVELS": str(lv), "UHDR_HIP_HUFF_SUB_BITS": "512"}, iters=2)
st = A.Stats()
u.lib.uhdr_hip_get_stats(ctx.handle, __import__("ctypes").byref(st))
print({n: getattr(st, n) for n, _ in st._fields_})
