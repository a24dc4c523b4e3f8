#!/usr/bin/env python
"""Randomised parity sweep of the HIP path against the oracle (run on the GPU box):
    python tests/fuzz_parity.py [--seconds 60] [--seed 1] [--log profiles/rNN_fuzz.log]
Random sizes (odd ones included), formats, map layouts / scales, metadata, strides.  Prints one line per
mismatch and a summary; exit code 1 if anything that must be bit-exact differs.
Also collected by pytest (tests/test_gpu_fuzz.py runs `run()` with a bounded budget under -m gpu)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # repo root (this script lives in tests/)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import Context, UltraHdr
from oracle import loader as L

F16, U32 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat, A.UHDR_IMG_FMT_32bppRGBA1010102
rng = np.random.default_rng(1)
ctx = None
u = None
stats = {}
bad = 0
KIND = "port"  # "ref" when the real reference is loadable: the generate / tonemap arms then compare against it


def init(seed, context=None):
    global rng, ctx, u, stats, bad, KIND
    rng = np.random.default_rng(seed)
    ctx = context or Context(0)
    u = UltraHdr(ctx=ctx)
    stats = {}
    bad = 0
    KIND = "ref" if L.ref() is not None else "port"


def rand_image(fmt, w, h, cg=None, align=None):
    img = Image(fmt, w, h, int(rng.integers(0, 3)) if cg is None else cg, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE,
                align=int(rng.choice([1, 2, 8, 64])) if align is None else align)
    img.buf[:] = rng.integers(0, 256, img.buf.size, dtype=np.uint8)
    return img


def note(op, ok, detail=""):
    global bad
    s = stats.setdefault(op, [0, 0])
    s[0] += 1
    if not ok:
        s[1] += 1
        bad += 1
        print("MISMATCH", op, detail, flush=True)


def fuzz_apply():
    w = int(rng.choice([64, 130, 131, 200, 256, 384, 515]))
    h = int(rng.choice([32, 66, 67, 128, 130]))
    base_fmt = int(rng.choice([A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_24bppYCbCr444, A.UHDR_IMG_FMT_16bppYCbCr422,
                               A.UHDR_IMG_FMT_32bppRGBA8888]))
    if base_fmt == A.UHDR_IMG_FMT_16bppYCbCr422:
        w += w % 2
    sdr = rand_image(base_fmt, w, h, align=int(rng.choice([1, 2, 64])) if base_fmt != A.UHDR_IMG_FMT_12bppYCbCr420 else int(rng.choice([2, 64])))
    scale = int(rng.choice([1, 1, 2, 3, 4, 8]))
    mw, mh = max(w // scale, 1), max(h // scale, 1)
    if rng.random() < 0.15:  # non-integer ratio with the same aspect is rare; keep the map covering the image
        mw, mh = w, h
    ch = int(rng.choice([1, 3]))
    alpha = bool(ch == 3 and rng.random() < 0.5)
    gm = synth.make_gainmap(mw, mh, ch, alpha, seed=int(rng.integers(1 << 30)), cg=int(rng.integers(0, 3)), align=int(rng.choice([1, 2, 64])))
    md = synth.default_metadata(max_boost=float(rng.uniform(1.5, 16)), min_boost=float(rng.uniform(0.5, 1.0)),
                                gamma=1.0 if rng.random() < 0.7 else float(rng.uniform(0.5, 2.5)),
                                offset=float(rng.choice([0.0, 1e-7, 1 / 64])), use_base_cg=int(rng.integers(0, 2)), per_channel=bool(ch == 3 and rng.random() < 0.5))
    ct = int(rng.choice([A.UHDR_CT_LINEAR, A.UHDR_CT_HLG, A.UHDR_CT_PQ]))
    boost = A.FLT_MAX if rng.random() < 0.6 else float(rng.uniform(1.0, md.hdr_capacity_max))
    fmt = F16 if ct == A.UHDR_CT_LINEAR else U32
    try:
        want = L.apply_gainmap("port", sdr, gm, md, ct, boost)
    except A.UhdrError as e:
        dest = Image(fmt, w, h, align=2)
        try:
            u.applyGainMap(sdr, gm, md, ct, fmt, boost, dest)
            note("apply-error", False, f"oracle raised {e.code}, hip succeeded")
        except A.UhdrError as e2:
            note("apply-error", e2.code == e.code, f"codes {e.code} vs {e2.code}")
        return
    dest = Image(fmt, w, h, align=2)
    try:
        u.applyGainMap(sdr, gm, md, ct, fmt, boost, dest)
    except A.UhdrError as e:
        note("apply", False, f"hip raised {e.code} {e}: fmt{base_fmt} {w}x{h} s{scale}")
        return
    exact_expected = md.gamma[0] == 1.0 or (mw == w and mh == h)
    a, b = dest.valid(0), want.valid(0)
    if exact_expected:
        note("apply-exact", np.array_equal(a, b), f"fmt{base_fmt} {w}x{h} map {mw}x{mh} ch{ch} a{alpha} ct{ct} diff {(a != b).sum()}")
    else:
        if ct == A.UHDR_CT_LINEAR:
            d = np.abs(a.view(np.uint16).astype(np.int32) - b.view(np.uint16).astype(np.int32))
        else:
            d = np.abs(np.stack([(a >> s) & 0x3FF for s in (0, 10, 20)], -1).astype(np.int32) - np.stack([(b >> s) & 0x3FF for s in (0, 10, 20)], -1).astype(np.int32))
        note("apply-pow", d.max() <= 1 and (d != 0).mean() < 2e-3, f"max {d.max()} frac {(d != 0).mean():.2e}")


def fuzz_generate():
    w, h = int(rng.choice([64, 128, 130, 256])), int(rng.choice([32, 64, 66]))
    kind = rng.choice(["p010", "1010102"])
    ct = int(rng.choice([A.UHDR_CT_HLG, A.UHDR_CT_PQ]))
    if kind == "p010":
        hdr = synth.make_hdr_p010(w, h, seed=int(rng.integers(1 << 30)), ct=ct, cg=int(rng.integers(0, 3)), noise=0.06,
                                  rng_range=int(rng.choice([A.UHDR_CR_LIMITED_RANGE, A.UHDR_CR_FULL_RANGE])))
        sdr = synth.make_sdr_yuv420(w, h, seed=int(rng.integers(1 << 30)), cg=int(rng.integers(0, 3)), noise=0.06)
    else:
        hdr = synth.make_hdr_rgba1010102(w, h, seed=int(rng.integers(1 << 30)), ct=ct, cg=int(rng.integers(0, 3)), noise=0.06)
        sdr = synth.make_sdr_rgba8888(w, h, seed=int(rng.integers(1 << 30)), cg=int(rng.integers(0, 3)), noise=0.06)
    cfg = A.default_encode_cfg(map_dimension_scale_factor=int(rng.choice([1, 1, 2, 4])), use_multi_channel_gainmap=int(rng.integers(0, 2)),
                               preset=int(rng.choice([A.UHDR_USAGE_REALTIME, A.UHDR_USAGE_BEST_QUALITY])), use_luminance=int(rng.integers(0, 2)),
                               sdr_is_601=int(rng.integers(0, 2)), gamma=1.0 if rng.random() < 0.7 else float(rng.uniform(0.6, 2.0)))
    if rng.random() < 0.3:  # user hints on the content boost: from ordinary to so narrow that pass 2 gets no step table (round 4)
        lo = float(rng.uniform(0.5, 3.0))
        cfg.min_content_boost, cfg.max_content_boost = lo, lo * float(rng.choice([1.00005, 1.001, 1.5, 8.0]))
    md_w, gm_w = L.generate_gainmap(KIND, sdr, hdr, cfg)
    from libultrahdr_amd.ultrahdr import UltraHdr as UH

    g = UH(ctx=ctx, mapDimensionScaleFactor=cfg.map_dimension_scale_factor, useMultiChannelGainMap=bool(cfg.use_multi_channel_gainmap),
           gamma=cfg.gamma, preset=cfg.preset, minContentBoost=cfg.min_content_boost, maxContentBoost=cfg.max_content_boost,
           targetDispPeakBrightness=cfg.target_disp_peak_nits)
    md_g, gm_g = g.generateGainMap(sdr, hdr, bool(cfg.sdr_is_601), bool(cfg.use_luminance))
    d = np.abs(gm_g.valid(0).astype(np.int32) - gm_w.valid(0).astype(np.int32))
    tol = 1e-4 if cfg.gamma == 1.0 else 5e-3
    note("generate", d.max() <= 1 and (d != 0).mean() <= tol and md_g.as_dict() == md_w.as_dict(),
         f"{kind} ct{ct} s{cfg.map_dimension_scale_factor} mc{cfg.use_multi_channel_gainmap} preset{cfg.preset} gamma{cfg.gamma:.2f} max {d.max()} frac {(d != 0).mean():.2e} md_eq {md_g.as_dict() == md_w.as_dict()}")


def fuzz_tonemap():
    w, h = int(rng.choice([64, 130, 256])), int(rng.choice([32, 66, 64]))
    kind = rng.choice(["p010", "1010102"])
    ct = int(rng.choice([A.UHDR_CT_HLG, A.UHDR_CT_PQ, A.UHDR_CT_LINEAR]))
    cg = int(rng.integers(0, 3))
    hdr = (synth.make_hdr_p010(w, h, seed=int(rng.integers(1 << 30)), ct=ct, cg=cg, noise=0.06) if kind == "p010"
           else synth.make_hdr_rgba1010102(w, h, seed=int(rng.integers(1 << 30)), ct=ct, cg=cg, noise=0.06))
    want = L.tone_map(KIND, hdr)
    got = Image(want.fmt, w, h, align=64)
    u.toneMap(hdr, got)
    n = tot = mx = 0
    for pg, pw in zip(got.planes_valid(), want.planes_valid()):
        if pg.dtype == np.uint32:
            pg, pw = pg.view(np.uint8), pw.view(np.uint8)
        d = np.abs(pg.astype(np.int32) - pw.astype(np.int32))
        n += int((d != 0).sum()); tot += d.size; mx = max(mx, int(d.max()))
    # +-1 on <= 1e-4 of the samples; a single differing sample passes whatever the image's size (these frames have 3-25 K samples): the reference's
    # srgbOetf goes through glibc's powf (faithfully rounded), the device's is correctly rounded -- about one code in 1e7-1e8 samples (DESIGN.md 1, row
    # a4-a8; seen once in 2.4e8 samples of a 480 s sweep, profiles/r06_fuzz_parity_long.log)
    note("tonemap", mx <= 1 and (n <= 1 or n / tot <= 1e-4), f"{kind} ct{ct} cg{cg} max {mx} differ {n}/{tot}")


def _uh_for(cfg):
    return UltraHdr(ctx=ctx, mapDimensionScaleFactor=cfg.map_dimension_scale_factor, useMultiChannelGainMap=bool(cfg.use_multi_channel_gainmap),
                    gamma=cfg.gamma, preset=cfg.preset, minContentBoost=cfg.min_content_boost, maxContentBoost=cfg.max_content_boost,
                    targetDispPeakBrightness=cfg.target_disp_peak_nits)


def fuzz_generate_formats():
    """generateGainMap over the input formats the API-1 default does not use: SDR 4:2:2 / 4:4:4 / 4:2:0 / RGBA8888 x
    HDR 30bppYCbCr444 (both ranges) / RGBA-F16 (with inf, NaN, negatives) / P010 / RGBA1010102, host and device buffers."""
    w, h = int(rng.choice([64, 128, 130, 258])), int(rng.choice([32, 64, 66]))
    sk = int(rng.choice([A.UHDR_IMG_FMT_16bppYCbCr422, A.UHDR_IMG_FMT_24bppYCbCr444, A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_32bppRGBA8888]))
    seed = int(rng.integers(1 << 30))
    sdr = (synth.make_sdr_rgba8888(w, h, seed=seed, cg=int(rng.integers(0, 3)), noise=0.06) if sk == A.UHDR_IMG_FMT_32bppRGBA8888
           else synth.make_sdr_planar(sk, w, h, seed=seed, cg=int(rng.integers(0, 3)), noise=0.06))
    hk = int(rng.choice([0, 0, 1, 1, 2, 3]))
    seed = int(rng.integers(1 << 30))
    ct = int(rng.choice([A.UHDR_CT_HLG, A.UHDR_CT_PQ]))
    if hk == 0:
        hdr = synth.make_hdr_yuv444_10bit(w, h, seed=seed, ct=ct, cg=int(rng.integers(0, 3)), noise=0.06,
                                          rng_range=int(rng.choice([A.UHDR_CR_LIMITED_RANGE, A.UHDR_CR_FULL_RANGE])))
    elif hk == 1:
        hdr = synth.make_hdr_rgba_f16(w, h, seed=seed, cg=int(rng.integers(0, 3)), noise=0.06, peak=float(rng.choice([4.0, 20.0, 60.0])))
    elif hk == 2:
        hdr = synth.make_hdr_p010(w, h, seed=seed, ct=ct, cg=int(rng.integers(0, 3)), noise=0.06)
    else:
        hdr = synth.make_hdr_rgba1010102(w, h, seed=seed, ct=ct, cg=int(rng.integers(0, 3)), noise=0.06)
    cfg = A.default_encode_cfg(map_dimension_scale_factor=int(rng.choice([1, 1, 2, 4])), use_multi_channel_gainmap=int(rng.integers(0, 2)),
                               preset=int(rng.choice([A.UHDR_USAGE_REALTIME, A.UHDR_USAGE_BEST_QUALITY])), use_luminance=int(rng.integers(0, 2)),
                               sdr_is_601=int(rng.integers(0, 2)))
    md_w, gm_w = L.generate_gainmap(KIND, sdr, hdr, cfg)
    g = _uh_for(cfg)
    if rng.random() < 0.5:
        md_g, gm_g = g.generateGainMap(sdr.to("cuda:0"), hdr.to("cuda:0"), bool(cfg.sdr_is_601), bool(cfg.use_luminance))
        ctx.synchronize()
        gm_g = gm_g.to_host()
    else:
        md_g, gm_g = g.generateGainMap(sdr, hdr, bool(cfg.sdr_is_601), bool(cfg.use_luminance))
    d = np.abs(gm_g.valid(0).astype(np.int32) - gm_w.valid(0).astype(np.int32))
    md_ok = all(np.allclose(md_g.as_dict()[k], md_w.as_dict()[k], rtol=1e-6, atol=0) for k in md_w.as_dict())
    note("generate-formats", d.max() <= 1 and (d != 0).mean() <= 1e-4 and md_ok,
         f"sdr fmt{sk} hdr kind{hk} ct{ct} {w}x{h} s{cfg.map_dimension_scale_factor} mc{cfg.use_multi_channel_gainmap} preset{cfg.preset} max {d.max()} frac {(d != 0).mean():.2e} md_ok {md_ok}")


def fuzz_tonemap_formats():
    """toneMap of 30bppYCbCr444 -> YCbCr444 and RGBA-F16 -> RGBA8888 (jpegr.cpp:1986-2103)."""
    w, h = int(rng.choice([64, 130, 256])), int(rng.choice([32, 66, 64]))
    seed = int(rng.integers(1 << 30))
    if rng.random() < 0.5:
        hdr = synth.make_hdr_yuv444_10bit(w, h, seed=seed, ct=int(rng.choice([A.UHDR_CT_HLG, A.UHDR_CT_PQ, A.UHDR_CT_LINEAR])), cg=int(rng.integers(0, 3)),
                                          noise=0.06, rng_range=int(rng.choice([A.UHDR_CR_LIMITED_RANGE, A.UHDR_CR_FULL_RANGE])))
    else:
        hdr = synth.make_hdr_rgba_f16(w, h, seed=seed, cg=int(rng.integers(0, 3)), noise=0.06, peak=float(rng.choice([4.0, 20.0, 60.0])))
    want = L.tone_map(KIND, hdr)
    got = Image(want.fmt, w, h, align=64, device="cuda:0")
    u.toneMap(hdr.to("cuda:0"), got)
    ctx.synchronize()
    got = got.to_host()
    n = tot = mx = 0
    for pg, pw in zip(got.planes_valid(), want.planes_valid()):
        if pg.dtype == np.uint32:
            pg, pw = pg.view(np.uint8), pw.view(np.uint8)
        d = np.abs(pg.astype(np.int32) - pw.astype(np.int32))
        n += int((d != 0).sum()); tot += d.size; mx = max(mx, int(d.max()))
    note("tonemap-formats", mx <= 1 and (n <= 1 or n / tot <= 1e-4), f"fmt{hdr.fmt} ct{hdr.raw.ct} cg{hdr.raw.cg} max {mx} differ {n}/{tot}")


def fuzz_converts():
    w, h = int(rng.choice([64, 130, 256])), int(rng.choice([32, 66, 64]))
    fmt = int(rng.choice([A.UHDR_IMG_FMT_32bppRGBA1010102, A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_24bppRGB888]))
    img = rand_image(fmt, w, h, align=64)
    chroma = bool(rng.integers(0, 2))
    want = L.convert_raw_input_to_ycbcr("port", img, chroma)
    got = u.convert_raw_input_to_ycbcr(img, chroma)
    note("raw2ycc", all(np.array_equal(a, b) for a, b in zip(got.planes_valid(), want.planes_valid())), f"fmt{fmt} chroma{chroma}")
    yfmt = int(rng.choice([A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_24bppYCbCr444]))
    src, dst = [int(v) for v in rng.choice(3, 2, replace=False)]
    img = rand_image(yfmt, w, h, cg=src, align=64)
    want = L.convert_yuv("port", img, src, dst)
    got = img.clone()
    u.convertYuv(got, src, dst)
    note("convertYuv", all(np.array_equal(a, b) for a, b in zip(got.planes_valid(), want.planes_valid())), f"fmt{yfmt} {src}->{dst}")
    # JPEG stage
    bw, bh = int(rng.integers(1, 40)), int(rng.integers(1, 12))
    plane = np.ascontiguousarray(rng.integers(0, 256, (bh * 8, bw * 8), dtype=np.uint8))
    q = int(rng.integers(1, 101))
    qt = u.quant_table(q, bool(rng.integers(0, 2)))
    coef = u.fdct_quant(plane, bw * 8, bw, bh, qt)
    note("fdct", np.array_equal(coef, L.fdct_quant_port(plane, bw * 8, bw, bh, qt)), f"{bw}x{bh} q{q}")
    note("idct", np.array_equal(u.idct_dequant(coef, qt), L.idct_dequant_port(coef, qt)), f"{bw}x{bh} q{q}")


def fuzz_decode_fused():
    """apply_gainmap_coef (IDCT inside the apply kernel) and idct_dequant_rgb (3-channel map in one pass) against the
    oracle's IDCT / colour conversion / applyGainMap chain."""
    w = int(rng.choice([128, 130, 200, 256, 384, 514]))
    h = int(rng.choice([16, 34, 66, 128, 130]))
    cw, chh = (w + 1) // 2, (h + 1) // 2
    dims = [((w + 7) // 8, (h + 7) // 8), ((cw + 7) // 8, (chh + 7) // 8), ((cw + 7) // 8, (chh + 7) // 8)]
    wild = rng.random() < 0.2
    qts = [np.full(64, 255, dtype=np.uint16)] * 3 if wild else [u.quant_table(int(rng.integers(1, 101)), c > 0) for c in range(3)]
    coefs = []
    for c, (bw, bh) in enumerate(dims):
        if wild:
            coefs.append(rng.integers(-32768, 32768, (bh, bw, 64), dtype=np.int16))
        else:
            pl = np.ascontiguousarray(rng.integers(0, 256, (bh * 8, bw * 8), dtype=np.uint8))
            coefs.append(L.fdct_quant_port(pl, bw * 8, bw, bh, qts[c]))
    dec = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, int(rng.integers(0, 3)), A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=2)
    for c in range(3):
        full = L.idct_dequant_port(coefs[c], qts[c])
        dec.valid(c)[:] = full[: dec.valid(c).shape[0], : dec.valid(c).shape[1]]
    scale = int(rng.choice([1, 1, 2, 4, 8]))
    if w % scale or h % scale:
        scale = 1
    ch = int(rng.choice([1, 3]))
    alpha = bool(ch == 3 and rng.random() < 0.5)
    gm = synth.make_gainmap(w // scale, h // scale, ch, alpha, seed=int(rng.integers(1 << 30)), cg=int(rng.integers(0, 3)), align=int(rng.choice([2, 64])))
    md = synth.default_metadata(max_boost=float(rng.uniform(1.5, 16)), min_boost=float(rng.uniform(0.5, 1.0)),
                                gamma=1.0 if scale > 1 or rng.random() < 0.7 else float(rng.uniform(0.5, 2.5)),
                                use_base_cg=int(rng.integers(0, 2)), per_channel=bool(ch == 3 and rng.random() < 0.5))
    ct = int(rng.choice([A.UHDR_CT_LINEAR, A.UHDR_CT_HLG, A.UHDR_CT_PQ]))
    fmt = F16 if ct == A.UHDR_CT_LINEAR else U32
    want = L.apply_gainmap("port", dec, gm, md, ct, A.FLT_MAX)
    dest = Image(fmt, w, h, align=4, device="cuda:0")
    u.applyGainMapFromCoefficients([torch.from_numpy(c).to("cuda:0") for c in coefs], qts, w, h, dec.raw.cg, gm.to("cuda:0"), md, ct, fmt,
                                   A.FLT_MAX, dest)
    ctx.synchronize()
    a, b = dest.to_host().valid(0), want.valid(0)
    note("apply-coef", np.array_equal(a, b), f"{w}x{h} s{scale} ch{ch} a{alpha} ct{ct} wild{wild} diff {(a != b).sum()}")
    # 3-channel map straight from its coefficients
    mw, mh = int(rng.integers(1, 300)), int(rng.integers(1, 70))
    bw, bh = (mw + 7) // 8, (mh + 7) // 8
    ql, qc = u.quant_table(int(rng.integers(1, 101)), False), u.quant_table(int(rng.integers(1, 101)), True)
    mco = [L.fdct_quant_port(np.ascontiguousarray(rng.integers(0, 256, (bh * 8, bw * 8), dtype=np.uint8)), bw * 8, bw, bh, ql if c == 0 else qc)
           for c in range(3)]
    planes = [L.idct_dequant_port(mco[c], ql if c == 0 else qc)[:mh, :mw] for c in range(3)]
    variant, bpp = int(rng.integers(0, 2)), int(rng.choice([3, 4]))
    want_rgb = L.jpeg_ycc_to_rgb_port(*[np.ascontiguousarray(p_) for p_ in planes], out_bpp=bpp, variant=variant)
    mfmt = A.UHDR_IMG_FMT_32bppRGBA8888 if bpp == 4 else A.UHDR_IMG_FMT_24bppRGB888
    mdst = Image(mfmt, mw, mh, align=int(rng.choice([1, 4, 64])), device="cuda:0")
    u.idct_dequant_rgb([torch.from_numpy(c).to("cuda:0") for c in mco], ql, qc, mw, mh, mfmt, variant, dst=mdst)
    ctx.synchronize()
    got = mdst.to_host().valid(0).view(np.uint8).reshape(mh, -1)[:, : mw * bpp]
    note("idct-rgb", np.array_equal(got, want_rgb), f"{mw}x{mh} bpp{bpp} v{variant}")


def fuzz_huffman():
    """huffman_encode against the oracle's sequential restatement: random geometry / sampling / restart interval / density."""
    ncomp = int(rng.choice([1, 3]))
    sampling = [(1, 1)] if ncomp == 1 else ([(2, 2), (1, 1), (1, 1)] if rng.random() < 0.6 else ([(2, 1), (1, 1), (1, 1)] if rng.random() < 0.5 else [(1, 1)] * 3))
    bpm = sum(a * b for a, b in sampling)
    ri = int(rng.integers(1, 64 // bpm + 1))
    w, h = int(rng.integers(1, 400)), int(rng.integers(1, 120))
    hmax, vmax = max(s_[0] for s_ in sampling), max(s_[1] for s_ in sampling)
    density = float(rng.choice([0.02, 0.1, 0.4, 1.0]))
    amp = int(rng.choice([3, 40, 1023]))
    coefs = []
    for hs, vs in sampling:
        cw, chh = -(-w * hs // hmax), -(-h * vs // vmax)
        bw, bh = -(-cw // 8), -(-chh // 8)
        if rng.random() < 0.3 and ncomp > 1:  # MCU-padded grid (what fdct_quant of padded planes yields): no dummy blocks
            bw, bh = -(-w // (8 * hmax)) * hs, -(-h // (8 * vmax)) * vs
        a = (rng.integers(-amp, amp + 1, (bh, bw, 64)) * (rng.random((bh, bw, 64)) < density)).astype(np.int16)
        a[..., 0] = rng.integers(-1020, 1021, (bh, bw))
        coefs.append(np.ascontiguousarray(a))
    want = L.huffman_encode_port(coefs, w, h, sampling, ri)
    got = u.huffman_encode([torch.from_numpy(c).to("cuda:0") for c in coefs], w, h, sampling, ri).cpu().numpy().tobytes()
    note("huffman", got == want, f"{w}x{h} {sampling} ri{ri} density{density} amp{amp} len {len(got)} vs {len(want)}")
    # and back: the oracle's stream through the device decoder (dummy blocks dropped: compare the real grid)
    data = torch.from_numpy(np.frombuffer(want, dtype=np.uint8).copy()).to("cuda:0")
    back = u.huffman_decode(data, [c.shape[:2] for c in coefs], w, h, sampling, ri)
    note("huffman-decode", all(np.array_equal(b.cpu().numpy(), c) for b, c in zip(back, coefs)), f"{w}x{h} {sampling} ri{ri}")


def fuzz_huffman_streams():
    """Decode only, frames large enough for the self-synchronising decoder (>= 4 KiB of entropy-coded data): no restart markers
    -- what the reference writes -- or restart intervals of any length (long ones are spliced out and take the same decoder,
    short ones one lane each); the oracle's stream must come back as the coefficients that went in."""
    ncomp = int(rng.choice([1, 3]))
    sampling = [(1, 1)] if ncomp == 1 else ([(2, 2), (1, 1), (1, 1)] if rng.random() < 0.6 else ([(2, 1), (1, 1), (1, 1)] if rng.random() < 0.5 else [(1, 1)] * 3))
    w, h = int(rng.integers(200, 1100)), int(rng.integers(64, 420))
    hmax, vmax = max(s_[0] for s_ in sampling), max(s_[1] for s_ in sampling)
    mcus = -(-w // (8 * hmax)) * -(-h // (8 * vmax))
    ri = 0 if rng.random() < 0.3 else int(rng.integers(1, max(2, mcus // 2)))
    density = float(rng.choice([0.03, 0.08, 0.15]))
    amp = int(rng.choice([3, 40, 300]))
    # noise: independent blocks; flat: constant blocks under a noisy band (a periodic bit pattern); smooth: a slowly varying DC
    # with the two lowest AC terms now and then -- what a gain map looks like (few, short symbols per block: the content the
    # self-synchronising decoder needs its longest windows for)
    style = str(rng.choice(["noise", "noise", "flat", "smooth"]))
    if style != "noise":
        w, h = w * 2, h * 2
    coefs = []
    for hs, vs in sampling:
        cw, chh = -(-w * hs // hmax), -(-h * vs // vmax)
        bw, bh = -(-cw // 8), -(-chh // 8)
        a = (rng.integers(-amp, amp + 1, (bh, bw, 64)) * (rng.random((bh, bw, 64)) < density)).astype(np.int16)
        a[..., 0] = rng.integers(-1020, 1021, (bh, bw))
        if style == "flat":
            top = int(rng.integers(1, 4))
            a[top:] = 0
            a[top:, :, 0] = int(rng.integers(-300, 300))
        elif style == "smooth":
            a[:] = 0
            a[..., 0] = np.clip(np.cumsum(rng.integers(-2, 3, (bh, bw)), axis=1) + int(rng.integers(-200, 200)), -1020, 1020)
            for k in (1, 8):
                a[..., k] = rng.integers(-2, 3, (bh, bw)) * (rng.random((bh, bw)) < 0.3)
        coefs.append(np.ascontiguousarray(a))
    scan = L.huffman_encode_port(coefs, w, h, sampling, ri)
    data = torch.from_numpy(np.frombuffer(scan, dtype=np.uint8).copy()).to("cuda:0")
    back = u.huffman_decode(data, [c.shape[:2] for c in coefs], w, h, sampling, ri)
    note("huffman-decode-streams", all(np.array_equal(b.cpu().numpy(), c) for b, c in zip(back, coefs)),
         f"{w}x{h} {sampling} ri{ri} {style} density{density} amp{amp} {len(scan)} B")


JOBS = None


def fuzz_api1_fused():
    """uhdr_hip_encode_api1_fused_dev against the operators it fuses (round 4): coefficient blocks, map bytes and metadata."""
    import torch

    from libultrahdr_amd.ultrahdr import UltraHdr as UH

    w, h = 16 * int(rng.integers(1, 24)), 16 * int(rng.integers(1, 12))
    scale = int(rng.choice([1, 1, 2]))
    if (w // scale) % 8 or (h // scale) % 8:
        scale = 1
    multi = bool(rng.integers(0, 2))
    cfg = A.default_encode_cfg(map_dimension_scale_factor=scale, use_multi_channel_gainmap=int(multi), use_luminance=int(rng.integers(0, 2)))
    if rng.random() < 0.25:
        lo = float(rng.uniform(0.5, 3.0))
        cfg.min_content_boost, cfg.max_content_boost = lo, lo * float(rng.choice([1.00005, 1.01, 4.0]))
    ct = int(rng.choice([A.UHDR_CT_HLG, A.UHDR_CT_PQ]))
    sdr = synth.make_sdr_yuv420(w, h, seed=int(rng.integers(1 << 30)), cg=int(rng.integers(0, 3)), noise=0.06).to("cuda:0")
    hdr = synth.make_hdr_p010(w, h, seed=int(rng.integers(1 << 30)), ct=ct, cg=int(rng.integers(0, 3)), noise=0.06,
                              rng_range=int(rng.choice([A.UHDR_CR_LIMITED_RANGE, A.UHDR_CR_FULL_RANGE]))).to("cuda:0")
    g = UH(ctx=ctx, mapDimensionScaleFactor=scale, useMultiChannelGainMap=multi, preset=cfg.preset, minContentBoost=cfg.min_content_boost,
           maxContentBoost=cfg.max_content_boost)
    ql, qc = L.quant_table_port(int(rng.integers(50, 100)), False), L.quant_table_port(int(rng.integers(50, 100)), True)
    enc = int(rng.choice([A.UHDR_CG_UNSPECIFIED, A.UHDR_CG_DISPLAY_P3, A.UHDR_CG_BT_709]))
    base_f, map_f, md_f, gm_f = g.encodeApi1Fused(sdr, hdr, enc, (ql, qc), (ql, qc), want_map=True, use_luminance=bool(cfg.use_luminance))
    md_s, gm_s = g.generateGainMap(sdr, hdr, False, bool(cfg.use_luminance))
    base = sdr.clone()
    if enc != A.UHDR_CG_UNSPECIFIED:
        g.convertYuv(base, sdr.raw.cg, enc)
    ok = md_f.as_dict() == md_s.as_dict() and torch.equal(gm_f.buf, gm_s.buf)
    for i in range(3):
        ref_c = g.fdct_quant(base.plane_tensor(i), base.raw.stride[i], (w if i == 0 else w // 2) // 8, (h if i == 0 else h // 2) // 8, ql if i == 0 else qc)
        ok = ok and torch.equal(base_f[i], ref_c.reshape(base_f[i].shape))
    map_s = g.fdct_quant_rgb(gm_s, ql, qc) if multi else [g.fdct_quant(gm_s.plane_tensor(0), gm_s.raw.stride[0], gm_s.w // 8, gm_s.h // 8, ql)]
    for i in range(len(map_s)):
        ok = ok and torch.equal(map_f[i], map_s[i].reshape(map_f[i].shape))
    ctx.synchronize()
    note("api1_fused", bool(ok), f"{w}x{h} s{scale} mc{int(multi)} enc{enc} ct{ct} hints {cfg.min_content_boost:.5g}..{cfg.max_content_boost:.5g}")


def fuzz_encode_image():
    """uhdr_hip_jpeg_encode_image (partial edge blocks padded on the device, round 4) against the REAL reference's entropy-coded bytes."""
    if KIND != "ref":
        return
    from libultrahdr_amd.images import Image
    from libultrahdr_amd.ultrahdr import UltraHdr as UH

    u = UH(ctx=ctx)
    kind = int(rng.integers(0, 3))
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 120))
    align = int(rng.choice([1, 1, 16, 64]))
    q = int(rng.integers(40, 100))
    if kind == 0:
        w, h = max(2, w & ~1), max(2, h & ~1)
        img, samp, rgb = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, align=align), [(2, 2), (1, 1), (1, 1)], 0
    elif kind == 1:
        img, samp, rgb = Image(A.UHDR_IMG_FMT_8bppYCbCr400, w, h, align=align), [(1, 1)], 0
    else:
        img, samp, rgb = Image(A.UHDR_IMG_FMT_24bppRGB888, w, h, align=max(align, 1)), [(1, 1)] * 3, 3
    for i, pl in enumerate(img.layout):
        if pl is not None:
            img.plane(i)[...] = rng.integers(0, 256, img.plane(i).shape, dtype=np.uint8)
    jpeg = L.ref_jpeg_compress(img, q)
    hd = u.jpeg_parse(jpeg)
    want = jpeg[hd.scan_offset: hd.scan_offset + hd.scan_bytes]
    ql = np.array(hd.qtable[0][:], dtype=np.uint16)
    qc = np.array(hd.qtable[1][:] if hd.scan.num_components == 3 else hd.qtable[0][:], dtype=np.uint16)
    planes = img.plane(0).reshape(h, img.raw.stride[0], 3) if rgb else [img.plane(i) for i, pl in enumerate(img.layout) if pl is not None]
    got = u.jpeg_encode_image(planes, w, h, samp, ql, qc, rgb_channels=rgb)
    note("encode_image", got == want, f"kind{kind} {w}x{h} align{align} q{q} {len(got)} vs {len(want)} bytes")


def fuzz_encode_errors():
    """Randomly poisoned descriptors (formats with their real layouts, gamut / transfer / range codes in and out of range, several at
    once so that the ORDER of the checks matters) through generateGainMap and toneMap, host and device entry points: the error code
    the real reference returns for the same descriptors (jpegr.cpp:536-562, 1986-2037); the matrices of tests/test_gpu_validation.py
    cover the single rejections one by one, this arm their combinations."""
    if os.path.dirname(os.path.abspath(__file__)) not in sys.path:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import test_gpu_validation as V

    def codes(n):
        return int(rng.choice([-1, 0, 1, 2, n, n + 1, n + 4]))

    sdr_fmts = [A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_24bppYCbCr444,
                A.UHDR_IMG_FMT_16bppYCbCr422, A.UHDR_IMG_FMT_8bppYCbCr400, A.UHDR_IMG_FMT_24bppYCbCrP010, A.UHDR_IMG_FMT_32bppRGBA1010102]
    hdr_fmts = [A.UHDR_IMG_FMT_24bppYCbCrP010, A.UHDR_IMG_FMT_24bppYCbCrP010, A.UHDR_IMG_FMT_32bppRGBA1010102, A.UHDR_IMG_FMT_64bppRGBAHalfFloat,
                A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_32bppRGBA8888]

    def mk_pair():
        sf, hf = int(rng.choice(sdr_fmts)), int(rng.choice(hdr_fmts))
        kw_s, kw_h = {}, {}
        if rng.random() < 0.4:
            kw_s["cg"] = codes(3)
        if rng.random() < 0.2:
            kw_s["fmt"] = int(rng.choice([A.UHDR_IMG_FMT_UNSPECIFIED, 40]))
        if rng.random() < 0.4:
            kw_h["cg"] = codes(3)
        if rng.random() < 0.4:
            kw_h["ct"] = codes(4)
        if rng.random() < 0.2:
            kw_h["range"] = codes(2)
        return (lambda: V.poison(V.sdr_of(sf), **kw_s)), (lambda: V.poison(V.hdr_of(hf, ct=A.UHDR_CT_HLG), **kw_h)), (sf, hf, kw_s, kw_h)

    device = bool(rng.integers(0, 2))
    mk_sdr, mk_hdr, what = mk_pair()
    cfg = V.default_cfg()
    if rng.random() < 0.3:
        cfg.map_dimension_scale_factor = int(rng.choice([1, 2, 4, 128]))
    want, _ = V.ref_code_generate(mk_sdr(), mk_hdr(), cfg)
    got, detail = V.hip_code_generate(ctx, mk_sdr(), mk_hdr(), cfg, device)
    note("generate-error" if want else "generate-accepted", got == want, f"dev{int(device)} {what}: hip {got} ({detail!r}), reference {want}")
    # toneMap: hdr descriptor poisoned the same way, destination of a random format
    mk_sdr, mk_hdr, what = mk_pair()
    df = int(rng.choice([A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_24bppYCbCr444, A.UHDR_IMG_FMT_8bppYCbCr400]))
    want, _ = V.ref_code_tonemap(mk_hdr(), V.tm_sdr(df))
    got, detail = V.hip_code_tonemap(ctx, mk_hdr(), V.tm_sdr(df), device)
    note("tonemap-error" if want else "tonemap-accepted", got == want, f"dev{int(device)} {what[1]} {what[3]} -> fmt{df}: hip {got} ({detail!r}), reference {want}")


def run(seconds, seed=1, context=None, log=None):
    """Runs the sweep for `seconds`; returns (stats, mismatches).  `log`: a path that receives the summary line."""
    init(seed, context)
    jobs = [fuzz_huffman, fuzz_apply, fuzz_apply, fuzz_apply, fuzz_generate, fuzz_generate_formats, fuzz_tonemap, fuzz_tonemap_formats,
            fuzz_converts, fuzz_decode_fused, fuzz_huffman_streams, fuzz_api1_fused, fuzz_encode_image, fuzz_encode_errors]
    t_end = time.time() + seconds
    i = 0
    while time.time() < t_end:
        jobs[i % len(jobs)]()
        i += 1
    summary = {k: tuple(v) for k, v in stats.items()}
    line = f"fuzz_parity seed={seed} seconds={seconds} oracle={KIND} cases per op (run, mismatched): {summary}"
    print(line)
    if log:
        os.makedirs(os.path.dirname(os.path.abspath(log)), exist_ok=True)
        with open(log, "a") as f:
            f.write(line + "\n")
    return summary, bad


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--log", default=None)
    args = ap.parse_args()
    _, nbad = run(args.seconds, args.seed, log=args.log)
    sys.exit(1 if nbad else 0)
