#!/usr/bin/env python
"""Runs one kernel configuration a few times -- the target for rocprofv3 (tools/profile.sh)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (first: see capi.load)

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import Context, UltraHdr

ap = argparse.ArgumentParser()
ap.add_argument("--case", default="8kA")  # <size><map>[hlg|pq]: 8kA 8kB 4kA 4kB 8kAhlg ...
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--sets", type=int, default=2)
args = ap.parse_args()

if args.case in ("gen4k", "tm4k"):  # encode-side kernels: two-pass 3-channel generate / P010 tone map, 4K
    w, h = 3840, 2160
    ctx = Context(0)
    sdr = synth.make_sdr_yuv420(w, h).to("cuda:0")
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG).to("cuda:0")
    enc = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_BEST_QUALITY)
    out = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, align=64, device="cuda:0")
    for i in range(args.iters):
        if args.case == "gen4k":
            enc.generateGainMap(sdr, hdr)
        else:
            enc.toneMap(hdr, out)
    ctx.synchronize()
    print("done", args.case, args.iters)
    sys.exit(0)

if args.case.startswith("coef"):  # coef4kA / coef8kC: applyGainMap with the base image in coefficient form (IDCT in the kernel)
    w, h = (7680, 4320) if "8k" in args.case else (3840, 2160)
    mk = args.case[-1]
    ctx = Context(0)
    u = UltraHdr(ctx=ctx)
    md = synth.default_metadata(use_base_cg=0)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    qts = [u.quant_table(95, False), u.quant_table(95, True), u.quant_table(95, True)]
    coefs = [torch.zeros(((h // d + 7) // 8, (w // d + 7) // 8, 64), dtype=torch.int16, device="cuda:0") for d in (1, 2, 2)]
    for c in coefs:
        c[..., 0] = 37
        c[..., 1] = -3
    gm = (synth.make_gainmap(w // 4, h // 4, 1, seed=50) if mk == "A" else synth.make_gainmap(w, h, 3, alpha=True, seed=50)).to("cuda:0")
    gm.raw.cg = A.UHDR_CG_BT_2100
    dst = Image(f16, w, h, align=64, device="cuda:0")
    torch.cuda.synchronize()
    for i in range(args.iters):
        u.applyGainMapFromCoefficients(coefs, qts, w, h, A.UHDR_CG_BT_709, gm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dst)
    ctx.synchronize()
    print("done", args.case, args.iters)
    sys.exit(0)

w, h = (7680, 4320) if args.case.startswith("8k") else (3840, 2160)
mk = args.case[2]
ct = A.UHDR_CT_HLG if "hlg" in args.case else A.UHDR_CT_PQ if "pq" in args.case else A.UHDR_CT_LINEAR
fmt = A.UHDR_IMG_FMT_64bppRGBAHalfFloat if ct == A.UHDR_CT_LINEAR else A.UHDR_IMG_FMT_32bppRGBA1010102
ctx = Context(0)
u = UltraHdr(ctx=ctx)
md = synth.default_metadata(use_base_cg=0)
sets = []
for i in range(args.sets):
    sdr = synth.make_sdr_yuv420(w, h, seed=10 + i)
    gm = synth.make_gainmap(w // 4, h // 4, 1, seed=50 + i) if mk == "A" else synth.make_gainmap(w, h, 3, alpha=(mk == "C"), seed=50 + i)
    sdr.raw.cg, gm.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    sets.append((sdr.to("cuda:0"), gm.to("cuda:0"), Image(fmt, w, h, align=64, device="cuda:0")))
for i in range(args.iters):
    s, g, d = sets[i % args.sets]
    u.applyGainMap(s, g, md, ct, fmt, A.FLT_MAX, d)
ctx.synchronize()
print("done", args.case, args.iters)
