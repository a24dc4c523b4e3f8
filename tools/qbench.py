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

if os.environ.get("UHDR_EXP_LIB"):  # an experimental build of the library (tools/code4_exp.py)
    A.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["UHDR_EXP_LIB"])
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import Context, UltraHdr

ctx = Context(0)
u = UltraHdr(ctx=ctx)
dev = "cuda:0"
md = synth.default_metadata(use_base_cg=0)
f16, u32 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat, A.UHDR_IMG_FMT_32bppRGBA1010102
REPS = int(os.environ.get("QB_REPS", "3"))
ITERS = int(os.environ.get("QB_ITERS", "0"))  # > 0: override every case's iteration count (profiling runs)
_tk = B.time_kernel
B.time_kernel = lambda ctx_, fn, iters=10, warm=3: _tk(ctx_, fn, iters=ITERS or iters, warm=1 if ITERS else warm)


def report(name, ms_list, bytes_, px):
    ms = sorted(ms_list)
    med = ms[len(ms) // 2]
    print(f"{name:28s} min {ms[0]*1e6:8.1f} med {med*1e6:8.1f} max {ms[-1]*1e6:8.1f} us   {bytes_/med/1e9:8.1f} GB/s ({bytes_/med/1e9/80:5.1f}% of 8 TB/s)  {px/med/1e6:9.0f} Mpx/s", flush=True)


def apply_case(name, w, h, mk, ct, nsets=int(os.environ.get("QB_NSETS", "3"))):
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

    for n_case, c in enumerate(cases):
        # section marker: an empty kernel of its own name, so that a profiler's kernel trace / counter rows can be cut per case
        # (tools/read_prof.py); the mapping section -> case is this line of the log
        ctx.lib.uhdr_hip_profile_mark(ctx.handle)
        print(f"MARK {n_case} {c}", flush=True)
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
        elif c in ("api1f", "api1f8k"):
            if c == "api1f":
                sdr, hdr = enc()
                ww, hh = w4, h4
            else:
                ww, hh = 7680, 4320
                sdr = synth.make_sdr_yuv420(ww, hh).to(dev)
                hdr = synth.make_hdr_p010(ww, hh, ct=A.UHDR_CT_HLG).to(dev)
            e = UltraHdr(ctx=ctx, mapDimensionScaleFactor=1, useMultiChannelGainMap=True, preset=A.UHDR_USAGE_BEST_QUALITY)
            qy, qc = u.quant_table(95, False), u.quant_table(95, True)
            r = [B.time_kernel(ctx, lambda: e.encodeApi1Fused(sdr, hdr, A.UHDR_CG_DISPLAY_P3, (qy, qc), (qy, qc), want_map=False), iters=10, warm=2) / 1e3
                 for _ in range(REPS)]
            report(c, r, 39.0 * ww * hh, ww * hh)
        elif c in ("fdct4k", "idct4k", "huff4k", "cvt4k"):
            sdr, hdr = enc()
            qt = u.quant_table(95, False)
            qts = [qt, u.quant_table(95, True), u.quant_table(95, True)]
            if c == "cvt4k":
                cv = sdr.clone()
                r = [B.time_kernel(ctx, lambda: u.convertYuv(cv, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3), iters=10, warm=2) / 1e3 for _ in range(REPS)]
                report(c, r, 3.0 * w4 * h4, w4 * h4)
            elif c in ("fdct4k", "idct4k"):
                plane = sdr.plane_tensor(0)
                coef = torch.empty((h4 // 8, w4 // 8, 64), dtype=torch.int16, device=dev)
                dec_plane = torch.empty((h4, w4), dtype=torch.uint8, device=dev)
                u.fdct_quant(plane, sdr.layout[0][1], w4 // 8, h4 // 8, qt, coef)
                fn = ((lambda: u.fdct_quant(plane, sdr.layout[0][1], w4 // 8, h4 // 8, qt, coef)) if c == "fdct4k"
                      else (lambda: u.idct_dequant(coef, qt, plane=dec_plane, stride=w4)))
                r = [B.time_kernel(ctx, fn, iters=10, warm=2) / 1e3 for _ in range(REPS)]
                report(c, r, 3.0 * w4 * h4, w4 * h4)
            else:
                hco = []
                for k in range(3):
                    rows, stride, wv = sdr.layout[k]
                    hco.append(u.fdct_quant(sdr.plane_tensor(k), stride, wv // 8, rows // 8, qts[k]))
                hout = torch.empty(w4 * h4 * 2, dtype=torch.uint8, device=dev)
                samp = [(2, 2), (1, 1), (1, 1)]
                r = [B.time_kernel(ctx, lambda: u.huffman_encode(hco, w4, h4, samp, 10, out=hout), iters=5, warm=2) / 1e3 for _ in range(REPS)]
                report("huff4k_encode_ri10", r, 4.5 * w4 * h4, w4 * h4)
                # a scan without restart markers, as the reference writes it: the primary image of an UltraHDR file encoded
                # through the drop-in facade (the reference's own libjpeg Huffman coder)
                from libultrahdr_amd import facade as FA
                if FA.available():
                    jpg = FA.encode(hdr.to_host(), sdr.to_host(), gpu=True)
                    hd = u.jpeg_parse(jpg)
                    import numpy as np
                    sc = hd.scan
                    data = torch.from_numpy(np.frombuffer(jpg, dtype=np.uint8)[hd.scan_offset: hd.scan_offset + hd.scan_bytes].copy()).to(dev)
                    bits = np.frombuffer(hd.tables.bits, dtype=np.uint8).reshape(4, 17)
                    vals = np.frombuffer(hd.tables.vals, dtype=np.uint8).reshape(4, 256)
                    shp0 = [(sc.blocks_h[c], sc.blocks_w[c]) for c in range(3)]
                    for env in (("",) if os.environ.get("QB_NO_SERIAL") else ("", "SERIAL")):
                        if env:
                            os.environ["UHDR_HIP_HUFF_SERIAL"] = "1"
                        r = [B.time_kernel(ctx, lambda: u.huffman_decode(data, shp0, sc.w, sc.h, samp, 0, tables=(bits, vals)), iters=3, warm=1) / 1e3 for _ in range(REPS)]
                        os.environ.pop("UHDR_HIP_HUFF_SERIAL", None)
                        report(f"huff4k_decode_ri0{'_serial_lane' if env else '_sync'} ({hd.scan_bytes} B)", r, 4.5 * w4 * h4, w4 * h4)
                for ri in (2, 10):
                    stream = u.huffman_encode(hco, w4, h4, samp, ri, out=hout).clone()
                    shp = [tuple(t.shape[:2]) for t in hco]
                    r = [B.time_kernel(ctx, lambda: u.huffman_decode(stream, shp, w4, h4, samp, ri), iters=3, warm=1) / 1e3 for _ in range(REPS)]
                    report(f"huff4k_decode_ri{ri}", r, 4.5 * w4 * h4, w4 * h4)
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
