#!/usr/bin/env python
"""The sixteen HLG / PQ applyGainMap variants that interpolate a three-channel map (scale 2 / 4) -- the ones that spilled to
scratch at six waves per SIMD (round-4 review): time them.  Run once with UHDR_HIP_SPILL_WPE3=0 (80 VGPRs, 76 B of scratch per
lane, two workgroups per CU) and once with =1 (the default since round 5: three waves per SIMD, 113 VGPRs, no scratch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401

import bench as B
from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import Context, UltraHdr

ctx = Context(0)
u = UltraHdr(ctx=ctx)
md = synth.default_metadata(use_base_cg=0)
u32 = A.UHDR_IMG_FMT_32bppRGBA1010102
w, h = 3840, 2160
sdrs = [synth.make_sdr_yuv420(w, h, seed=5 + i).to("cuda:0") for i in range(3)]
dsts = [Image(u32, w, h, align=64, device="cuda:0") for _ in range(3)]
for s in sdrs:
    s.raw.cg = A.UHDR_CG_BT_709
print("UHDR_HIP_SPILL_WPE3 =", os.environ.get("UHDR_HIP_SPILL_WPE3", "(default 1)"))
for ct, cn in ((A.UHDR_CT_HLG, "hlg"), (A.UHDR_CT_PQ, "pq")):
    for alpha in (False, True):
        for scale in (2, 4):
            gms = [synth.make_gainmap(w // scale, h // scale, 3, alpha=alpha, seed=70 + i).to("cuda:0") for i in range(3)]
            for g in gms:
                g.raw.cg = A.UHDR_CG_BT_2100
            k = [0]

            def fn():
                i = k[0] % 3
                k[0] += 1
                u.applyGainMap(sdrs[i], gms[i], md, ct, u32, A.FLT_MAX, dsts[i])

            ms = sorted(B.time_kernel(ctx, fn, iters=30, warm=5) for _ in range(3))[1]
            by = (1.5 + (4.0 if alpha else 3.0) / scale / scale + 4.0) * w * h
            print(f"apply_4k_{cn}_map3ch_{'rgba' if alpha else 'rgb'}_s{scale}: {ms * 1e3:7.1f} us  {by / ms / 1e6:7.1f} GB/s ({by / ms / 1e6 / 80:4.1f} % of 8 TB/s)", flush=True)
