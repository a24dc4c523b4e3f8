#!/usr/bin/env python
"""Quick kernel timings (HIP events through uhdr_hip_profile_*) for optimisation loops:
    python tools/qbench.py 8kC 8kA 4kAhlg 4kApq b16hlg tm4k gen4k gen4k1 api0 ...
"""
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
dev = "cuda:0"
md = synth.default_metadata(use_base_cg=0)
f16, u32 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat, A.UHDR_IMG_FMT_32bppRGBA1010102
REPS = int(os.environ.get("QB_REPS", "3"))


def report(name, ms_list, bytes_, px):
    ms = sorted(ms_list)
    med = ms[len(ms) // 2]
    print(f"{name:28s} min {ms[0]*1e3:8.1f} med {med*1e3:8.1f} max {ms[-1]*1e3:8.1f} us   {bytes_/med/1e6:8.1f} GB/s ({bytes_/med/1e6/80:5.1f}% of 8 TB/s)  {px/med/1e3:9.0f} Mpx/s", flush=True)


def apply_case(name, w, h, mk, ct, nsets=3):
    fmt = f16 if ct == A.UHDR_CT_LINEAR else u32
    sets = B.make_frames(nsets, w, h, mk, dev, fmt, seed0=77)
    for s, g, _ in sets:
        s.raw.cg, g.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    k = [0]

    def fn():
        s, g, d = sets[k[0] % nsets]
        k[0] += 1
        u.applyGainMap(s, g, md, ct, fmt, A.FLT_MAX, d)

    r = [B.time_kernel(ctx, fn, iters=30, warm=5) / 1e3 for _ in range(REPS)]
    report(name, r, B.algo_bytes_per_px(mk, 8 if ct == A.UHDR_CT_LINEAR else 4) * w * h, w * h)


def batch_case(name, n, mk, ct):
    fmt = f16 if ct == A.UHDR_CT_LINEAR else u32
    sets = B.make_frames(n, 3840, 2160, mk, dev, fmt, seed0=555)
    for s, g, _ in sets:
        s.raw.cg, g.raw.cg = A.UHDR_CG_BT_709, A.UHDR_CG_BT_2100
    fn = lambda: u.applyGainMapBatch([f[0] for f in sets], [f[1] for f in sets], md, ct, fmt, A.FLT_MAX, [f[2] for f in sets])
    r = [B.time_kernel(ctx, fn, iters=10, warm=2) / 1e3 for _ in range(REPS)]
    report(name, r, B.algo_bytes_per_px(mk, 8 if ct == A.UHDR_CT_LINEAR else 4) * 3840 * 2160 * n, 3840 * 2160 * n)


def main():
    cases = sys.argv[1:] or ["8kC", "8kA"]
    w4, h4 = 3840, 2160
    enc_inputs = {}

    def enc():
        if not enc_inputs:
            enc_inputs["sdr"] = synth.make_sdr_yuv420(w4, h4).to(dev)
            enc_inputs["hdr"] = synth.make_hdr_p010(w4, h4, ct=A.UHDR_CT_HLG).to(dev)
        return enc_inputs["sdr"], enc_inputs["hdr"]

    for c in cases:
        if c[:2] in ("8k", "4k") and len(c) >= 3 and c[2] in "ABC":
            w, h = (7680, 4320) if c[0] == "8" else (w4, h4)
            ct = A.UHDR_CT_HLG if "hlg" in c else A.UHDR_CT_PQ if "pq" in c else A.UHDR_CT_LINEAR
            apply_case(c, w, h, c[2], ct)
        elif c.startswith("b"):  # b16hlg b32hlg b16C ...
            n = int("".join(ch for ch in c[1:] if ch.isdigit()))
            ct = A.UHDR_CT_HLG if "hlg" in c else A.UHDR_CT_PQ if "pq" in c else A.UHDR_CT_LINEAR
            mk = "C" if c.endswith("C") else "A"
            batch_case(c, n, mk, ct)
        elif c == "tm4k":
            sdr, hdr = enc()
            out = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w4, h4, align=64, device=dev)
            r = [B.time_kernel(ctx, lambda: u.toneMap(hdr, out), iters=10, warm=2) / 1e3 for _ in range(REPS)]
            report(c, r, 4.5 * w4 * h4, w4 * h4)
        elif c in ("gen4k", "gen4k1"):
            sdr, hdr = enc()
            e = (UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_BEST_QUALITY) if c == "gen4k"
                 else UltraHdr(ctx=ctx, mapDimensionScaleFactor=4, useMultiChannelGainMap=False, preset=A.UHDR_USAGE_REALTIME))
            r = [B.time_kernel(ctx, lambda: e.generateGainMap(sdr, hdr), iters=10, warm=2) / 1e3 for _ in range(REPS)]
            report(c, r, (31.5 if c == "gen4k" else 4.5 + 1 / 16) * w4 * h4, w4 * h4)
        elif c in ("api0", "api0f", "tm8k"):
            w8, h8 = 7680, 4320
            hdr8 = synth.make_hdr_rgba1010102(w8, h8, ct=A.UHDR_CT_PQ).to(dev)
            sdr8 = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w8, h8, align=64, device=dev)
            e0 = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_REALTIME)
            if c == "tm8k":
                r = [B.time_kernel(ctx, lambda: u.toneMap(hdr8, sdr8), iters=6, warm=2) / 1e3 for _ in range(REPS)]
                report(c, r, 8.0 * w8 * h8, w8 * h8)
            elif c == "api0f":
                r = [B.time_kernel(ctx, lambda: e0.encodeApi0Fused(hdr8, want_sdr_rgba=False, use_luminance=False), iters=6, warm=2) / 1e3 for _ in range(REPS)]
                report(c, r, 10.0 * w8 * h8, w8 * h8)
            else:
                def f():
                    u.toneMap(hdr8, sdr8)
                    e0.generateGainMap(sdr8, hdr8, False, False)
                r = [B.time_kernel(ctx, f, iters=6, warm=2) / 1e3 for _ in range(REPS)]
                report(c, r, 19.0 * w8 * h8, w8 * h8)
            del hdr8, sdr8
        torch.cuda.empty_cache()


main()
