"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Bit-exact for integer / LUT-only paths and for applyGainMap's HLG tail (host-built
threshold tables); the tolerance for the remaining per-pixel transcendental sites (SURVEY.md 8a:
hlgOotfApprox, srgbOetf, encodeGain, computeGain -- glibc's own results for these vary with the
CPU's ifunc variant) is +-1 output code with a stated bound on how many samples may differ.
Run on the GPU box: pytest -m gpu."""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from oracle import loader as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uhdr(hip_ctx):
    from libultrahdr_amd.ultrahdr import UltraHdr

    return UltraHdr(ctx=hip_ctx)


def oracle_kind():
    return "ref" if L.ref() is not None else "port"


def planes_equal(a: Image, b: Image):
    return all(np.array_equal(x, y) for x, y in zip(a.to_host().planes_valid(), b.to_host().planes_valid()))


def unpack1010102(v):
    return np.stack([(v >> s) & 0x3FF for s in (0, 10, 20)], -1).astype(np.int32), (v >> 30)


def assert_close_codes(got, want, max_code_diff=1, max_frac=0.01, what=""):
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= max_code_diff, f"{what}: max code diff {d.max()}"
    frac = (d != 0).mean()
    assert frac <= max_frac, f"{what}: {frac:.4%} of samples differ (allowed {max_frac:.2%})"


def hip_apply(uhdr, sdr, gm, md, ct, boost=A.FLT_MAX, device=False):
    fmt = A.UHDR_IMG_FMT_64bppRGBAHalfFloat if ct == A.UHDR_CT_LINEAR else A.UHDR_IMG_FMT_32bppRGBA1010102
    if device:
        dsdr, dgm = sdr.to("cuda:0"), gm.to("cuda:0")
        dest = Image(fmt, sdr.w, sdr.h, align=2, device="cuda:0")
        uhdr.applyGainMap(dsdr, dgm, md, ct, fmt, boost, dest)
        uhdr.ctx.synchronize()
        return dest.to_host()
    dest = Image(fmt, sdr.w, sdr.h, align=1)
    uhdr.applyGainMap(sdr, gm, md, ct, fmt, boost, dest)
    return dest


def check_apply(uhdr, sdr, gm, md, ct, boost=A.FLT_MAX, device=False, what=""):
    want = L.apply_gainmap(oracle_kind(), sdr, gm, md, ct, boost)
    got = hip_apply(uhdr, sdr, gm, md, ct, boost, device)
    # LINEAR (F16) and PQ are table-only; the HLG tail (3 x powf per pixel in the reference) runs on
    # threshold tables built with this host's libm: all three are bit exact against the oracle
    # evaluated on the same host
    assert np.array_equal(got.valid(0), want.valid(0)), f"{what}: {(got.valid(0) != want.valid(0)).sum()} pixels differ"
    assert got.raw.cg == want.raw.cg


@pytest.mark.parametrize("ch,alpha,scale", [(1, False, 4), (1, False, 2), (1, False, 1), (3, False, 1), (3, True, 1),
                                            (3, False, 2), (3, True, 4), (1, False, 8), (1, False, 16)])
@pytest.mark.parametrize("out_ct", [A.UHDR_CT_LINEAR, A.UHDR_CT_HLG, A.UHDR_CT_PQ])
def test_apply_gainmap_quad_path(uhdr, ch, alpha, scale, out_ct):
    """4:2:0 base, even geometry: the quad kernel (scale 16 falls back to the generic kernel)."""
    w, h = 384, 192
    sdr = synth.make_sdr_yuv420(w, h, noise=0.05)
    gm = synth.make_gainmap(w // scale, h // scale, ch, alpha, cg=A.UHDR_CG_BT_2100)
    for use_base_cg in (0, 1):
        md = synth.default_metadata(use_base_cg=use_base_cg, per_channel=(ch == 3))
        check_apply(uhdr, sdr, gm, md, out_ct, what=f"host ubc={use_base_cg}")
    check_apply(uhdr, sdr, gm, synth.default_metadata(), out_ct, device=True, what="device")


def test_apply_gainmap_generic_path(uhdr):
    w, h = 130, 66
    rng = np.random.default_rng(7)
    gm1, gm3 = synth.make_gainmap(w // 2, h // 2, 1), synth.make_gainmap(w, h, 3)
    for fmt in (A.UHDR_IMG_FMT_24bppYCbCr444, A.UHDR_IMG_FMT_16bppYCbCr422, A.UHDR_IMG_FMT_32bppRGBA8888,
                A.UHDR_IMG_FMT_24bppRGB888):
        sdr = Image(fmt, w, h, A.UHDR_CG_DISPLAY_P3, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
        sdr.buf[:] = rng.integers(0, 256, sdr.buf.size, dtype=np.uint8)
        for gm in (gm1, gm3):
            for ct in (A.UHDR_CT_LINEAR, A.UHDR_CT_PQ):
                check_apply(uhdr, sdr, gm, synth.default_metadata(), ct, what=f"fmt{fmt}")
                check_apply(uhdr, sdr, gm, synth.default_metadata(), ct, boost=2.5, what=f"fmt{fmt} weight<1")
    # odd-sized 4:2:0, ragged strides
    sdr = synth.make_sdr_yuv420(131, 67, align=1)
    check_apply(uhdr, sdr, synth.make_gainmap(131, 67, 3), synth.default_metadata(), A.UHDR_CT_LINEAR, what="odd 420")


def test_apply_gainmap_gamma_and_fractional_scale(uhdr):
    """gamma != 1 needs pow() per sample, the non-integer scale path needs sqrt(): device double
    math vs glibc -> tolerance: identical half-float codes for >= 99.9% of channels, never more
    than 1 ulp (of the half) apart."""
    sdr = synth.make_sdr_yuv420(240, 120)
    cases = [(synth.make_gainmap(60, 30, 1), synth.default_metadata(gamma=1.7)),
             (synth.make_gainmap(160, 80, 1), synth.default_metadata()),
             (synth.make_gainmap(96, 48, 3), synth.default_metadata(per_channel=True))]
    for gm, md in cases:
        want = L.apply_gainmap(oracle_kind(), sdr, gm, md, A.UHDR_CT_LINEAR)
        got = hip_apply(uhdr, sdr, gm, md, A.UHDR_CT_LINEAR)
        a = got.valid(0).view(np.uint16).astype(np.int32)
        b = want.valid(0).view(np.uint16).astype(np.int32)
        assert_close_codes(a, b, 1, 0.001, "gamma/fractional")
    # scale 1 with gamma != 1 goes through the host-built byte->factor table: exact
    check_apply(uhdr, sdr, synth.make_gainmap(240, 120, 3), synth.default_metadata(gamma=2.2, per_channel=True), A.UHDR_CT_LINEAR)


def test_apply_gainmap_error_behaviour(uhdr):
    """Same codes as UltraHdr::applyGainMap (jpegr.cpp:1538-1614, tests/jpegr_test.cpp:1425-1478)."""
    sdr = synth.make_sdr_yuv420(64, 32)
    gm = synth.make_gainmap(16, 8, 1)
    md = synth.default_metadata()
    f16, u32 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat, A.UHDR_IMG_FMT_32bppRGBA1010102

    def code(fn):
        with pytest.raises(A.UhdrError) as e:
            fn()
        return e.value.code

    dest = Image(f16, 64, 32, align=1)
    assert code(lambda: uhdr.applyGainMap(sdr, gm, md, A.UHDR_CT_SRGB, f16, 4.0, dest)) == A.UHDR_CODEC_INVALID_PARAM
    assert code(lambda: uhdr.applyGainMap(sdr, gm, md, A.UHDR_CT_HLG, f16, 4.0, dest)) == A.UHDR_CODEC_INVALID_PARAM
    assert code(lambda: uhdr.applyGainMap(sdr, gm, md, A.UHDR_CT_LINEAR, u32, 4.0, Image(u32, 64, 32, align=1))) == A.UHDR_CODEC_INVALID_PARAM
    bad = synth.default_metadata()
    bad.hdr_capacity_max = 0.5
    assert code(lambda: uhdr.applyGainMap(sdr, gm, bad, A.UHDR_CT_LINEAR, f16, 4.0, dest)) == A.UHDR_CODEC_INVALID_PARAM
    p010 = synth.make_hdr_p010(64, 32)
    assert code(lambda: uhdr.applyGainMap(p010, gm, md, A.UHDR_CT_LINEAR, f16, 4.0, dest)) == A.UHDR_CODEC_UNSUPPORTED_FEATURE
    assert code(lambda: uhdr.applyGainMap(sdr, sdr, md, A.UHDR_CT_LINEAR, f16, 4.0, dest)) == A.UHDR_CODEC_UNSUPPORTED_FEATURE
    narrow = Image(f16, 64, 32, align=1)
    narrow.raw.stride[0] = 32
    assert code(lambda: uhdr.applyGainMap(sdr, gm, md, A.UHDR_CT_LINEAR, f16, 4.0, narrow)) == A.UHDR_CODEC_INVALID_PARAM


def test_apply_gainmap_stripes_equal_whole(uhdr):
    """Row-stripe sharding: 4 stripes with the replicated map == whole image."""
    from libultrahdr_amd.images import stripe_view

    w, h = 512, 256
    sdr = synth.make_sdr_yuv420(w, h).to("cuda:0")
    gm = synth.make_gainmap(w // 4, h // 4, 1).to("cuda:0")
    md = synth.default_metadata()
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    whole = Image(f16, w, h, align=2, device="cuda:0")
    uhdr.applyGainMap(sdr, gm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, whole)
    parts = Image(f16, w, h, align=2, device="cuda:0")
    lib, ctx = uhdr.lib, uhdr.ctx
    for k in range(4):
        r0, rows = k * 64, 64
        s_view, d_view = stripe_view(sdr, r0, rows), stripe_view(parts, r0, rows)
        A.check(lib.uhdr_hip_apply_gainmap_dev(ctx.handle, C.byref(s_view), C.byref(gm.raw), C.byref(md), A.UHDR_CT_LINEAR,
                                               f16, A.FLT_MAX, C.byref(d_view), r0, h))
    ctx.synchronize()
    assert planes_equal(whole, parts)


@pytest.mark.parametrize("ch,alpha,scale", [(1, False, 4), (3, True, 1), (3, False, 1)])
def test_apply_gainmap_batch_equals_per_frame(uhdr, ch, alpha, scale):
    """uhdr_hip_apply_gainmap_batch_dev: one launch over n frames == n single launches == oracle."""
    w, h, n = 256, 128, 5
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    md = synth.default_metadata(use_base_cg=0, per_channel=(ch == 3))
    sdrs = [synth.make_sdr_yuv420(w, h, seed=100 + i, noise=0.05) for i in range(n)]
    gms = [synth.make_gainmap(w // scale, h // scale, ch, alpha, seed=200 + i, cg=A.UHDR_CG_BT_2100) for i in range(n)]
    dsdr, dgm = [s.to("cuda:0") for s in sdrs], [g.to("cuda:0") for g in gms]
    dests = [Image(f16, w, h, align=2, device="cuda:0") for _ in range(n)]
    uhdr.applyGainMapBatch(dsdr, dgm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dests)
    uhdr.ctx.synchronize()
    for i in range(n):
        want = L.apply_gainmap(oracle_kind(), sdrs[i], gms[i], md, A.UHDR_CT_LINEAR)
        assert np.array_equal(dests[i].to_host().valid(0), want.valid(0)), f"frame {i}"
        assert dests[i].raw.cg == want.raw.cg
    # a batch the quad kernel does not cover (4:4:4 base) silently runs frame by frame
    s444 = Image(A.UHDR_IMG_FMT_24bppYCbCr444, 64, 32, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
    s444.buf[:] = np.random.default_rng(1).integers(0, 256, s444.buf.size, dtype=np.uint8)
    g1 = synth.make_gainmap(32, 16, 1)
    d2 = [Image(f16, 64, 32, align=2, device="cuda:0") for _ in range(2)]
    uhdr.applyGainMapBatch([s444.to("cuda:0")] * 2, [g1.to("cuda:0")] * 2, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, d2)
    uhdr.ctx.synchronize()
    want = L.apply_gainmap(oracle_kind(), s444, g1, md, A.UHDR_CT_LINEAR)
    assert np.array_equal(d2[1].to_host().valid(0), want.valid(0))


# ---------------------------------------------------------------------------------------------------
GEN_CASES = [
    (dict(kind="p010", ct=A.UHDR_CT_HLG), "yuv420", dict()),
    (dict(kind="p010", ct=A.UHDR_CT_PQ), "yuv420", dict()),
    (dict(kind="p010", ct=A.UHDR_CT_PQ), "yuv420", dict(map_dimension_scale_factor=4, use_multi_channel_gainmap=0, preset=A.UHDR_USAGE_REALTIME)),
    (dict(kind="p010", ct=A.UHDR_CT_HLG), "yuv420", dict(map_dimension_scale_factor=2, use_multi_channel_gainmap=0)),
    (dict(kind="1010102", ct=A.UHDR_CT_PQ), "rgba8888", dict(preset=A.UHDR_USAGE_REALTIME, use_luminance=0)),
    (dict(kind="1010102", ct=A.UHDR_CT_PQ), "rgba8888", dict(preset=A.UHDR_USAGE_REALTIME, use_luminance=0, use_multi_channel_gainmap=0, map_dimension_scale_factor=2, gamma=1.4)),
    (dict(kind="1010102", ct=A.UHDR_CT_HLG, cg=A.UHDR_CG_DISPLAY_P3), "rgba8888", dict(min_content_boost=0.8, max_content_boost=6.0, target_disp_peak_nits=1600.0)),
    (dict(kind="p010", ct=A.UHDR_CT_HLG), "yuv420", dict(sdr_is_601=1, gamma=0.8, use_multi_channel_gainmap=0, map_dimension_scale_factor=3)),
    # round 4 (two-pass maps are ratio planes + per-channel step tables built on the device): ranges the table builder must cope with
    # -- user hints that cut the range down to 1.4e-4 log2 units (too dense for a table: pass 2 evaluates per sample), a range of
    # one binade, and gamma != 1 at full resolution with three channels (no tables at all: the per-sample kernel)
    (dict(kind="p010", ct=A.UHDR_CT_HLG), "yuv420", dict(min_content_boost=2.0, max_content_boost=2.0002)),
    (dict(kind="p010", ct=A.UHDR_CT_PQ), "yuv420", dict(min_content_boost=1.0, max_content_boost=2.0)),
    (dict(kind="p010", ct=A.UHDR_CT_HLG), "yuv420", dict(gamma=1.25)),
]


def _pair(w, h, hdr_kw, sdr_kind):
    kw = dict(hdr_kw)
    kind = kw.pop("kind")
    if kind == "p010":
        hdr = synth.make_hdr_p010(w, h, ct=kw.get("ct", A.UHDR_CT_HLG), cg=kw.get("cg", A.UHDR_CG_BT_2100), noise=0.04)
    else:
        hdr = synth.make_hdr_rgba1010102(w, h, ct=kw.get("ct", A.UHDR_CT_PQ), cg=kw.get("cg", A.UHDR_CG_BT_2100), noise=0.04)
    sdr = synth.make_sdr_yuv420(w, h, noise=0.04) if sdr_kind == "yuv420" else synth.make_sdr_rgba8888(w, h, noise=0.04)
    return sdr, hdr


def _uhdr_for(hip_ctx, cfg):
    from libultrahdr_amd.ultrahdr import UltraHdr

    return UltraHdr(ctx=hip_ctx, mapDimensionScaleFactor=cfg.map_dimension_scale_factor,
                    useMultiChannelGainMap=bool(cfg.use_multi_channel_gainmap), gamma=cfg.gamma, preset=cfg.preset,
                    minContentBoost=cfg.min_content_boost, maxContentBoost=cfg.max_content_boost,
                    targetDispPeakBrightness=cfg.target_disp_peak_nits)


@pytest.mark.parametrize("hdr_kw,sdr_kind,cfg_kw", GEN_CASES)
@pytest.mark.parametrize("device", [False, True])
def test_generate_gainmap(hip_ctx, hdr_kw, sdr_kind, cfg_kw, device):
    """The per-sample double log2 runs on float64 tables (csrc/exact_math.h) whose result is the
    correctly rounded float in all but ~1e-8 of the cases, HLG's powf is folded into a host-built
    table: measured bit-identical to the real reference on every case at 1280x720
    (tests/parity_stats.py).  Allowed here: +-1 map code on <= 1e-4 of the samples (gamma != 1 keeps a
    device powf per sample); metadata within 1e-6 relative."""
    sdr, hdr = _pair(256, 128, hdr_kw, sdr_kind)
    cfg = A.default_encode_cfg(**cfg_kw)
    md_w, gm_w = L.generate_gainmap(oracle_kind(), sdr, hdr, cfg)
    u = _uhdr_for(hip_ctx, cfg)
    if device:
        md_g, gm_g = u.generateGainMap(sdr.to("cuda:0"), hdr.to("cuda:0"), bool(cfg.sdr_is_601), bool(cfg.use_luminance))
        hip_ctx.synchronize()
        gm_g = gm_g.to_host()
    else:
        md_g, gm_g = u.generateGainMap(sdr, hdr, bool(cfg.sdr_is_601), bool(cfg.use_luminance))
    assert (gm_g.raw.fmt, gm_g.raw.w, gm_g.raw.h) == (gm_w.raw.fmt, gm_w.raw.w, gm_w.raw.h)
    assert (gm_g.raw.cg, gm_g.raw.ct, gm_g.raw.range) == (gm_w.raw.cg, gm_w.raw.ct, gm_w.raw.range)
    assert_close_codes(gm_g.valid(0), gm_w.valid(0), 1, 1e-4, "gain map")
    dg, dw = md_g.as_dict(), md_w.as_dict()
    for k in dw:
        assert np.allclose(dg[k], dw[k], rtol=1e-6, atol=0), (k, dg[k], dw[k])


def test_generate_then_apply_roundtrip_full_size(uhdr, hip_ctx):
    """Size-independent property at BASELINE's 4K size: encode (2-pass, 3ch, s=1) -> decode recovers
    the HDR rendition: PSNR of the recovered linear HDR vs the tone-curve-free ground truth."""
    w, h = 3840, 2160
    sdr = synth.make_sdr_yuv420(w, h).to("cuda:0")
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_PQ).to("cuda:0")
    u = _uhdr_for(hip_ctx, A.default_encode_cfg())
    md, gm = u.generateGainMap(sdr, hdr)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    dest = Image(f16, w, h, align=2, device="cuda:0")
    u.applyGainMap(sdr, gm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest)
    hip_ctx.synchronize()
    # idempotence: the same call again gives the same bytes; checksum-of-checksums over 8 row bands
    dest2 = Image(f16, w, h, align=2, device="cuda:0")
    u.applyGainMap(sdr, gm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest2)
    hip_ctx.synchronize()
    a, b = dest.to_host().valid(0), dest2.to_host().valid(0)
    assert np.array_equal(a, b)
    # compare a 256-row band against the oracle (seconds on the CPU)
    from libultrahdr_amd.images import stripe_view

    band = 256
    sdr_h, gm_h = sdr.to_host(), gm.to_host()
    sdr_band = Image(sdr_h.fmt, w, band, sdr_h.raw.cg, sdr_h.raw.ct, sdr_h.raw.range)
    sdr_band.valid(0)[:] = sdr_h.valid(0)[:band]
    sdr_band.valid(1)[:] = sdr_h.valid(1)[: band // 2]
    sdr_band.valid(2)[:] = sdr_h.valid(2)[: band // 2]
    gm_band = Image(gm_h.fmt, w, band + 1, gm_h.raw.cg)
    gm_band.valid(0)[:] = gm_h.valid(0)[: band + 1]
    # aspect ratio of the band differs from the whole image but the band pair is self-consistent
    want = L.apply_gainmap("port", sdr_band, gm_band, md, A.UHDR_CT_LINEAR)
    assert np.array_equal(want.valid(0)[: band - 1], a[: band - 1])
    # recovered HDR has the right scale: mean luminance ratio vs SDR within the metadata's boost range
    rec = a[:band].view(np.float16).astype(np.float32).reshape(band, w, 4)[..., :3]
    assert np.isfinite(rec).all() and rec.min() >= 0.0 and rec.max() <= 10000.0 / 203.0 + 1e-3


@pytest.mark.parametrize("kind,ct,cg", [("p010", A.UHDR_CT_HLG, A.UHDR_CG_BT_2100), ("p010", A.UHDR_CT_PQ, A.UHDR_CG_DISPLAY_P3),
                                        ("1010102", A.UHDR_CT_PQ, A.UHDR_CG_BT_2100), ("1010102", A.UHDR_CT_HLG, A.UHDR_CG_BT_709),
                                        ("p010", A.UHDR_CT_LINEAR, A.UHDR_CG_BT_2100)])
def test_tone_map(uhdr, kind, ct, cg):
    """srgbOetf's powf runs on float64 tables (correctly rounded; glibc's powf is faithfully rounded and
    differs from that by one ulp on ~5e-4 of its inputs, which moves an 8-bit code about once per 1e7
    samples); HLG's OOTF powf is folded into a host-built table.  Measured bit-identical to the real
    reference on every case at 1280x720 (tests/parity_stats.py).  Allowed: +-1 code on <= 1e-4 of samples."""
    w, h = 256, 128
    hdr = synth.make_hdr_p010(w, h, ct=ct, cg=cg) if kind == "p010" else synth.make_hdr_rgba1010102(w, h, ct=ct, cg=cg)
    want = L.tone_map(oracle_kind(), hdr)
    got = Image(want.fmt, w, h, align=64)
    uhdr.toneMap(hdr, got)
    assert (got.raw.cg, got.raw.ct, got.raw.range) == (A.UHDR_CG_DISPLAY_P3, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
    for pg, pw in zip(got.planes_valid(), want.planes_valid()):
        if pg.dtype == np.uint32:
            pg, pw = pg.view(np.uint8), pw.view(np.uint8)
        assert_close_codes(pg, pw, 1, 1e-4, "tone map")
    dgot = Image(want.fmt, w, h, align=64, device="cuda:0")
    uhdr.toneMap(hdr.to("cuda:0"), dgot)
    uhdr.ctx.synchronize()
    assert planes_equal(dgot, got)


@pytest.mark.parametrize("src,dst", [(0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1), (1, 1)])
def test_convert_yuv_bit_exact(uhdr, src, dst):
    rng = np.random.default_rng(11)
    for fmt, (w, h) in ((A.UHDR_IMG_FMT_12bppYCbCr420, (256, 128)), (A.UHDR_IMG_FMT_24bppYCbCr444, (130, 66))):
        img = Image(fmt, w, h, src, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
        img.buf[:] = rng.integers(0, 256, img.buf.size, dtype=np.uint8)
        want = L.convert_yuv(oracle_kind(), img, src, dst)
        got = img.clone()
        uhdr.convertYuv(got, src, dst)
        assert planes_equal(got, want)
        dgot = img.to("cuda:0")
        uhdr.convertYuv(dgot, src, dst)
        uhdr.ctx.synchronize()
        assert planes_equal(dgot, want)


@pytest.mark.parametrize("fmt", [A.UHDR_IMG_FMT_32bppRGBA1010102, A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_24bppRGB888])
@pytest.mark.parametrize("chroma", [False, True])
def test_convert_raw_input_to_ycbcr_bit_exact(uhdr, fmt, chroma):
    rng = np.random.default_rng(13)
    for cg in (0, 1, 2):
        img = Image(fmt, 256, 128, cg, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
        img.buf[:] = rng.integers(0, 256, img.buf.size, dtype=np.uint8)
        want = L.convert_raw_input_to_ycbcr(oracle_kind(), img, chroma)
        got = uhdr.convert_raw_input_to_ycbcr(img, chroma)
        assert got.raw.fmt == want.raw.fmt and got.raw.range == A.UHDR_CR_FULL_RANGE
        assert planes_equal(got, want)
        dgot = uhdr.convert_raw_input_to_ycbcr(img.to("cuda:0"), chroma)
        uhdr.ctx.synchronize()
        assert planes_equal(dgot, want)


@pytest.mark.parametrize("quality", [95, 75, 30])
def test_fdct_quant_bit_exact(uhdr, quality):
    rng = np.random.default_rng(17)
    for (w, h) in ((512, 256), (72, 40)):  # second: block count not a multiple of 8
        plane = np.ascontiguousarray(rng.integers(0, 256, (h, w), dtype=np.uint8))
        smooth = synth.make_sdr_yuv420(w, h, align=8).plane(0)[:h, :w]
        for pl in (plane, np.ascontiguousarray(smooth)):
            for chroma in (False, True):
                qt = uhdr.quant_table(quality, chroma)
                assert np.array_equal(qt, L.quant_table_port(quality, chroma))
                want = L.fdct_quant_port(pl, w, w // 8, h // 8, qt)
                got = uhdr.fdct_quant(pl, w, w // 8, h // 8, qt)
                assert np.array_equal(got, want)


def test_fdct_quant_full_size_properties(uhdr):
    """8K plane: DC of every block == round-half-away((sum - 64*128) * 8 / (q0 << 3)) and the whole
    coefficient field equals the oracle on a 64-row band."""
    import torch

    w, h = 7680, 4320
    img = synth.make_sdr_yuv420(w, h, align=64)
    plane = torch.from_numpy(img.plane(0)).to("cuda:0")
    qt = uhdr.quant_table(95, False)
    coef = uhdr.fdct_quant(plane, img.plane(0).shape[1], w // 8, h // 8, qt)
    uhdr.ctx.synchronize()
    coef = coef.cpu().numpy()
    band = L.fdct_quant_port(img.plane(0), img.plane(0).shape[1], w // 8, 8, qt)
    assert np.array_equal(coef[:8], band)
    blocks = img.plane(0)[:h, :w].reshape(h // 8, 8, w // 8, 8).astype(np.int64).sum(axis=(1, 3)) - 64 * 128
    q = int(qt[0]) << 3
    dc = blocks  # the islow output is scaled by 8, the quantizer divides by q << 3: DC == sum of (p - 128)
    want_dc = np.sign(dc) * ((np.abs(dc) + (q >> 1)) // q)
    assert np.array_equal(coef[..., 0].astype(np.int64), want_dc)


# ---- JPEG decode stage ------------------------------------------------------------------
@pytest.mark.parametrize("quality", [95, 40])
def test_idct_dequant_bit_exact(uhdr, quality):
    """dequant + islow IDCT + range limit == the oracle (itself pinned against libjpeg through the
    reference's JpegDecoderHelper, tests/test_oracle_vs_ref.py), host and device entry points; the last
    case feeds garbage coefficients (the 32-bit multiply / modulo-1024 range-limit path)."""
    import torch

    rng = np.random.default_rng(23)
    for (w, h) in ((512, 256), (72, 40)):
        plane = np.ascontiguousarray(synth.make_sdr_yuv420(w, h, align=8, noise=0.1).plane(0)[:h, :w])
        qt = uhdr.quant_table(quality, False)
        coef = L.fdct_quant_port(plane, w, w // 8, h // 8, qt)
        want = L.idct_dequant_port(coef, qt)
        got = uhdr.idct_dequant(coef, qt)
        assert np.array_equal(got, want)
        dgot = uhdr.idct_dequant(torch.from_numpy(coef).to("cuda:0"), qt)
        uhdr.ctx.synchronize()
        assert np.array_equal(dgot.cpu().numpy(), want)
        # ragged stride (byte stores) and a round trip: decode(encode(x)) stays within the quantisation error
        got2 = uhdr.idct_dequant(coef, qt, stride=w + 3)
        assert np.array_equal(got2[:, :w], want)
        assert np.abs(want.astype(np.int32) - plane.astype(np.int32)).mean() <= (2.0 if quality == 95 else 16.0)
    wild = rng.integers(-32768, 32768, (4, 9, 64), dtype=np.int16)
    qt = np.full(64, 255, dtype=np.uint16)
    assert np.array_equal(uhdr.idct_dequant(wild, qt), L.idct_dequant_port(wild, qt))


@pytest.mark.parametrize("fmt", [A.UHDR_IMG_FMT_24bppRGB888, A.UHDR_IMG_FMT_32bppRGBA8888])
@pytest.mark.parametrize("size", [(256, 64), (131, 17)])
def test_jpeg_colour_conversions_bit_exact(uhdr, fmt, size):
    """libjpeg's rgb_ycc_convert / ycc_rgb_convert (both constant variants), vector and scalar paths."""
    w, h = size
    bpp = 4 if fmt == A.UHDR_IMG_FMT_32bppRGBA8888 else 3
    rng = np.random.default_rng(29)
    rgb = Image(fmt, w, h, align=4 if w % 4 == 0 else 1)
    rgb.buf[:] = rng.integers(0, 256, rgb.buf.size, dtype=np.uint8)
    packed = np.ascontiguousarray(rgb.valid(0).view(np.uint8).reshape(h, -1)[:, : w * bpp])
    rgb888 = packed if bpp == 3 else np.ascontiguousarray(packed.reshape(h, w, 4)[:, :, :3].reshape(h, w * 3))
    want = L.jpeg_rgb_to_ycc_port(rgb888, w, w, h)
    for src in (rgb, rgb.to("cuda:0")):
        ycc = uhdr.jpeg_rgb_to_ycc(src)
        uhdr.ctx.synchronize()
        ycc = ycc.to_host()
        assert ycc.raw.fmt == A.UHDR_IMG_FMT_24bppYCbCr444
        for c in range(3):
            assert np.array_equal(ycc.valid(c), want[c]), f"plane {c}"
    # back: random YCbCr (covers out-of-gamut triples that exercise the clamps)
    ycc = Image(A.UHDR_IMG_FMT_24bppYCbCr444, w, h, align=4 if w % 4 == 0 else 1)
    ycc.buf[:] = rng.integers(0, 256, ycc.buf.size, dtype=np.uint8)
    for variant in (0, 1):
        want_rgb = L.jpeg_ycc_to_rgb_port(ycc.valid(0), ycc.valid(1), ycc.valid(2), out_bpp=bpp, variant=variant)
        for src in (ycc, ycc.to("cuda:0")):
            out = uhdr.jpeg_ycc_to_rgb(src, fmt, variant)
            uhdr.ctx.synchronize()
            got = out.to_host().valid(0).view(np.uint8).reshape(h, -1)[:, : w * bpp]
            assert np.array_equal(got, want_rgb), f"variant {variant}"


def test_decode_stage_feeds_apply_gainmap(uhdr):
    """The 8f-1 chain on the device: coefficient blocks of the base image (Y, Cb, Cr at 4:2:0) and of a
    Y400 gain map -> idct_dequant -> planes -> applyGainMap == the oracle on the oracle's own decode."""
    import torch

    w, h = 256, 128
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    sdr = synth.make_sdr_yuv420(w, h, align=8, noise=0.05)
    gm = synth.make_gainmap(w // 4, h // 4, 1, align=8)
    md = synth.default_metadata()
    qy, qc = uhdr.quant_table(95, False), uhdr.quant_table(95, True)
    # host: what the Huffman decoder would hand over
    coefs = [L.fdct_quant_port(np.ascontiguousarray(sdr.valid(c)), sdr.valid(c).shape[1], sdr.valid(c).shape[1] // 8,
                               sdr.valid(c).shape[0] // 8, qy if c == 0 else qc) for c in range(3)]
    coef_gm = L.fdct_quant_port(np.ascontiguousarray(gm.valid(0)), w // 4, w // 32, h // 32, qy)
    # oracle chain
    sdr_dec = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, sdr.raw.cg, sdr.raw.ct, sdr.raw.range, align=8)
    for c in range(3):
        sdr_dec.valid(c)[:] = L.idct_dequant_port(coefs[c], qy if c == 0 else qc)
    gm_dec = Image(A.UHDR_IMG_FMT_8bppYCbCr400, w // 4, h // 4, gm.raw.cg, align=8)
    gm_dec.valid(0)[:] = L.idct_dequant_port(coef_gm, qy)
    want = L.apply_gainmap(oracle_kind(), sdr_dec, gm_dec, md, A.UHDR_CT_LINEAR)
    # device chain: only the coefficients cross PCIe
    dsdr = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, sdr.raw.cg, sdr.raw.ct, sdr.raw.range, align=8, device="cuda:0")
    for c in range(3):
        pl = dsdr.plane_tensor(c)
        uhdr.idct_dequant(torch.from_numpy(coefs[c]).to("cuda:0"), qy if c == 0 else qc, plane=pl, stride=pl.shape[1])
    dgm = Image(A.UHDR_IMG_FMT_8bppYCbCr400, w // 4, h // 4, gm.raw.cg, align=8, device="cuda:0")
    pl = dgm.plane_tensor(0)
    uhdr.idct_dequant(torch.from_numpy(coef_gm).to("cuda:0"), qy, plane=pl, stride=pl.shape[1])
    dest = Image(f16, w, h, align=2, device="cuda:0")
    uhdr.applyGainMap(dsdr, dgm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest)
    uhdr.ctx.synchronize()
    assert planes_equal(dsdr, sdr_dec) and planes_equal(dgm, gm_dec)
    assert np.array_equal(dest.to_host().valid(0), want.valid(0))


def _coef_case(uhdr, w, h, quality, rng, wild=False):
    """Coefficients of a random 4:2:0 image (as the Huffman decoder would hand them over), the three tables, and the
    oracle's decode of them as an Image."""
    cw, chh = (w + 1) // 2, (h + 1) // 2
    dims = [((w + 7) // 8, (h + 7) // 8), ((cw + 7) // 8, (chh + 7) // 8), ((cw + 7) // 8, (chh + 7) // 8)]
    qts = [uhdr.quant_table(quality, False), uhdr.quant_table(quality, True), uhdr.quant_table(max(quality - 10, 1), True)]
    coefs = []
    for c, (bw, bh) in enumerate(dims):
        if wild:
            coefs.append(rng.integers(-32768, 32768, (bh, bw, 64), dtype=np.int16))
        else:
            # smooth field + noise so that the decoded image is not just clipped garbage
            yy, xx = np.mgrid[0:bh * 8, 0:bw * 8]
            pl = 128 + 90 * np.sin(xx / (13.0 + 5 * c)) * np.cos(yy / (9.0 + 3 * c)) + rng.normal(0, 12, (bh * 8, bw * 8))
            pl = np.clip(pl, 0, 255).astype(np.uint8)
            coefs.append(L.fdct_quant_port(np.ascontiguousarray(pl), bw * 8, bw, bh, qts[c]))
    dec = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=2)
    for c in range(3):
        full = L.idct_dequant_port(coefs[c], qts[c])
        dec.valid(c)[:] = full[: dec.valid(c).shape[0], : dec.valid(c).shape[1]]
    return coefs, qts, dec


@pytest.mark.parametrize("ch,alpha,scale", [(1, False, 4), (1, False, 1), (3, False, 1), (3, True, 1), (3, True, 2), (1, False, 8)])
@pytest.mark.parametrize("out_ct", [A.UHDR_CT_LINEAR, A.UHDR_CT_HLG, A.UHDR_CT_PQ])
def test_apply_gainmap_from_coefficients(uhdr, ch, alpha, scale, out_ct):
    """SURVEY 8f-1 as worded: dequant + IDCT fused into applyGainMap.  == the oracle's IDCT followed by the oracle's
    applyGainMap, bit for bit; sizes that are not whole tiles (128 x 16) / MCUs, three different quantization tables."""
    import torch

    rng = np.random.default_rng(61)
    fmt = A.UHDR_IMG_FMT_64bppRGBAHalfFloat if out_ct == A.UHDR_CT_LINEAR else A.UHDR_IMG_FMT_32bppRGBA1010102
    for (w, h, quality) in ((512, 64, 95), (392, 200, 70)):
        if w % scale or h % scale:
            continue
        coefs, qts, dec = _coef_case(uhdr, w, h, quality, rng)
        gm = synth.make_gainmap(w // scale, h // scale, ch, alpha, cg=A.UHDR_CG_BT_2100)
        md = synth.default_metadata(use_base_cg=0, per_channel=(ch == 3))
        want = L.apply_gainmap(oracle_kind(), dec, gm, md, out_ct)
        dest = Image(fmt, w, h, align=4, device="cuda:0")
        uhdr.applyGainMapFromCoefficients([torch.from_numpy(c).to("cuda:0") for c in coefs], qts, w, h, A.UHDR_CG_BT_709,
                                          gm.to("cuda:0"), md, out_ct, fmt, A.FLT_MAX, dest)
        uhdr.ctx.synchronize()
        assert np.array_equal(dest.to_host().valid(0), want.valid(0)), (w, h)


def test_apply_gainmap_from_coefficients_corrupt_and_errors(uhdr):
    """Garbage coefficients take the 32-bit multiply path and the modulo-1024 range limit, exactly as libjpeg would;
    geometry the quad kernel does not cover is refused (no silent fallback), a wrong block grid is an invalid parameter."""
    import torch

    rng = np.random.default_rng(67)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    w, h = 256, 48
    coefs, qts, dec = _coef_case(uhdr, w, h, 50, rng, wild=True)
    qts = [np.full(64, 255, dtype=np.uint16)] * 3
    for c in range(3):
        full = L.idct_dequant_port(coefs[c], qts[c])
        dec.valid(c)[:] = full[: dec.valid(c).shape[0], : dec.valid(c).shape[1]]
    gm = synth.make_gainmap(w // 4, h // 4, 1)
    md = synth.default_metadata()
    want = L.apply_gainmap(oracle_kind(), dec, gm, md, A.UHDR_CT_LINEAR)
    dcoefs = [torch.from_numpy(c).to("cuda:0") for c in coefs]
    dest = Image(f16, w, h, align=2, device="cuda:0")
    uhdr.applyGainMapFromCoefficients(dcoefs, qts, w, h, A.UHDR_CG_BT_709, gm.to("cuda:0"), md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest)
    uhdr.ctx.synchronize()
    assert np.array_equal(dest.to_host().valid(0), want.valid(0))
    with pytest.raises(A.UhdrError) as e:  # block grid of a different image
        uhdr.applyGainMapFromCoefficients(dcoefs, qts, w + 16, h, A.UHDR_CG_BT_709, gm.to("cuda:0"), md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM
    # a 3 x 3 map scale is the generic kernel's business: refused here
    w3, h3 = 264, 48
    coefs3, qts3, _ = _coef_case(uhdr, w3, h3, 80, rng)
    with pytest.raises(A.UhdrError) as e:
        uhdr.applyGainMapFromCoefficients([torch.from_numpy(c).to("cuda:0") for c in coefs3], qts3, w3, h3, A.UHDR_CG_BT_709,
                                          synth.make_gainmap(w3 // 3, h3 // 3, 1).to("cuda:0"), md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX,
                                          Image(f16, w3, h3, align=2, device="cuda:0"))
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE


# ---- entropy stage ----------------------------------------------------------------------------------
def _random_coefs(rng, w, h, sampling, kind):
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    coefs = []
    for hs, vs in sampling:
        cw, chh = -(-w * hs // hmax), -(-h * vs // vmax)
        bw, bh = -(-cw // 8), -(-chh // 8)
        if kind == "sparse":  # what a q ~ 90 photo looks like: few non-zero terms, long zero runs (ZRL), many EOBs
            a = (rng.normal(0, 25, (bh, bw, 64)) * (rng.random((bh, bw, 64)) < 0.12)).astype(np.int16)
            a[..., 0] = rng.integers(-1000, 1000, (bh, bw))
            a[..., 63] = np.where(rng.random((bh, bw)) < 0.3, 1, a[..., 63])  # runs of > 16 zeros ending in the last term
        elif kind == "dense":  # every term non-zero, all size categories
            a = rng.integers(-1023, 1024, (bh, bw, 64)).astype(np.int16)
            a[a == 0] = 1
        elif kind == "worst":  # 16-bit codes + 10 magnitude bits for every AC term: 1660 bits per block
            a = np.full((bh, bw, 64), 1023, dtype=np.int16)
            a[..., 1::2] = -1023
            a[..., 0] = np.where(rng.random((bh, bw)) < 0.5, 1023, -1024)
        else:  # "zero": only EOBs
            a = np.zeros((bh, bw, 64), dtype=np.int16)
        coefs.append(np.ascontiguousarray(a))
    return coefs


@pytest.mark.parametrize("kind", ["sparse", "dense", "worst", "zero"])
def test_huffman_encode_equals_oracle_byte_for_byte(uhdr, kind):
    """One wavefront per restart interval == the sequential restatement of libjpeg's encoder (itself byte-identical to the
    reference encoder, tests/test_oracle_vs_ref.py), for 4:2:0 / 4:4:4 / single-component scans, sizes with dummy blocks
    at the right and bottom edges, the smallest and the largest restart intervals, byte stuffing included."""
    import torch

    rng = np.random.default_rng(83)
    cases = [(256, 64, [(2, 2), (1, 1), (1, 1)], 10), (72, 40, [(2, 2), (1, 1), (1, 1)], 1), (50, 30, [(2, 2), (1, 1), (1, 1)], 3),
             (41, 23, [(1, 1)] * 3, 21), (200, 24, [(1, 1)] * 3, 4), (37, 19, [(1, 1)], 64), (520, 16, [(1, 1)], 7)]
    for (w, h, sampling, ri) in cases:
        coefs = _random_coefs(rng, w, h, sampling, kind)
        want = L.huffman_encode_port(coefs, w, h, sampling, ri)
        got = uhdr.huffman_encode([torch.from_numpy(c).to("cuda:0") for c in coefs], w, h, sampling, ri)
        got = got.cpu().numpy().tobytes()
        assert len(got) == len(want), (kind, w, h, ri, len(got), len(want))
        assert got == want, (kind, w, h, ri)


@pytest.mark.parametrize("kind", ["sparse", "dense", "worst", "zero"])
def test_huffman_encode_without_restart_markers_equals_oracle_byte_for_byte(uhdr, kind):
    """restart_interval 0 -- the stream the reference itself writes (jpegencoderhelper.cpp:187-201): DC prediction chained
    through the whole scan, no byte alignment between the wavefront segments, flush_bits only at the very end.  The device's
    three passes (lengths, scan, emit + stuff) equal the sequential restatement of jchuff.c byte for byte: every sampling
    layout, dummy blocks on both edges, scans of a single segment and of hundreds, the 1660-bit worst-case block."""
    import torch

    rng = np.random.default_rng(131)
    cases = [(256, 64, [(2, 2), (1, 1), (1, 1)]), (72, 40, [(2, 2), (1, 1), (1, 1)]), (50, 30, [(2, 2), (1, 1), (1, 1)]), (16, 16, [(2, 2), (1, 1), (1, 1)]),
             (41, 23, [(1, 1)] * 3), (200, 24, [(1, 1)] * 3), (45, 21, [(2, 1), (1, 1), (1, 1)]), (37, 19, [(1, 1)]), (520, 16, [(1, 1)]), (8, 8, [(1, 1)]),
             (1000, 520, [(2, 2), (1, 1), (1, 1)]), (1030, 260, [(1, 1)] * 3), (2048, 600, [(1, 1)])]
    for (w, h, sampling) in cases:
        if kind == "worst" and w * h > 300000:
            continue  # the oracle walks these serially
        coefs = _random_coefs(rng, w, h, sampling, kind)
        want = L.huffman_encode_port(coefs, w, h, sampling, 0)
        got = uhdr.huffman_encode([torch.from_numpy(c).to("cuda:0") for c in coefs], w, h, sampling, 0)
        got = got.cpu().numpy().tobytes()
        assert len(got) == len(want), (kind, w, h, len(got), len(want))
        assert got == want, (kind, w, h)


def test_huffman_encode_without_restart_markers_reports_the_size_it_needs(uhdr):
    """An output buffer that is too small: UHDR_CODEC_MEM_ERROR and the required size, nothing written beyond the buffer."""
    import torch

    rng = np.random.default_rng(5)
    w, h, sampling = 256, 128, [(2, 2), (1, 1), (1, 1)]
    coefs = _random_coefs(rng, w, h, sampling, "dense")
    want = L.huffman_encode_port(coefs, w, h, sampling, 0)
    dev = [torch.from_numpy(c).to("cuda:0") for c in coefs]
    out = torch.full((len(want) // 2 + 64,), 0xA5, dtype=torch.uint8, device="cuda:0")
    guard = out[len(want) // 2:]
    with pytest.raises(A.UhdrError) as e:
        uhdr.huffman_encode(dev, w, h, sampling, 0, out=out[: len(want) // 2])
    assert e.value.code == A.UHDR_CODEC_MEM_ERROR
    assert bool((guard == 0xA5).all())
    out = torch.empty(len(want), dtype=torch.uint8, device="cuda:0")
    got = uhdr.huffman_encode(dev, w, h, sampling, 0, out=out)
    assert got.cpu().numpy().tobytes() == want


def test_huffman_encode_without_restart_markers_writes_the_reference_encoders_scan(uhdr):
    """Device FDCT + quantize + marker-less Huffman coding of a 4:2:0 frame == the entropy-coded segment of the file the
    REFERENCE's JpegEncoderHelper writes for the same planes and quality (needs the reference build)."""
    import torch

    if oracle_kind() != "ref":
        pytest.skip("needs oracle/_ref")
    ref = L.ref()
    for (w, h, q) in [(384, 160, 90), (1280, 720, 95)]:
        img = synth.make_sdr_yuv420(w, h, align=8, noise=0.1)
        ql, qc = uhdr.quant_table(q, False), uhdr.quant_table(q, True)
        coefs = []
        for c in range(3):
            pl = np.ascontiguousarray(img.valid(c))
            coefs.append(uhdr.fdct_quant(torch.from_numpy(pl).to("cuda:0"), pl.shape[1], (pl.shape[1] + 7) // 8, (pl.shape[0] + 7) // 8, ql if c == 0 else qc))
        scan = uhdr.huffman_encode(coefs, w, h, [(2, 2), (1, 1), (1, 1)], 0).cpu().numpy().tobytes()
        out = np.zeros(1 << 23, dtype=np.uint8)
        n = ref.ref_jpeg_compress(C.byref(img.raw), q, out.ctypes.data, out.size)
        jpeg = out[:n].tobytes()
        i = 2
        while jpeg[i + 1] != 0xDA:
            i += 2 + ((jpeg[i + 2] << 8) | jpeg[i + 3])
        start = i + 2 + ((jpeg[i + 2] << 8) | jpeg[i + 3])
        assert scan == jpeg[start:-2], (w, h, q, len(scan), len(jpeg) - start - 2)


def test_huffman_encode_of_a_real_frame_decodes_with_libjpeg(uhdr):
    """The full device encode chain of a 4:2:0 frame -- FDCT + quantize, Huffman coding, file wrapper -- read back by the
    real libjpeg (through the reference build, when it travelled with the snapshot): same coefficients, same tables."""
    import torch

    w, h, ri = 384, 160, 8
    img = synth.make_sdr_yuv420(w, h, align=8, noise=0.1)
    ql, qc = uhdr.quant_table(90, False), uhdr.quant_table(90, True)
    coefs = []
    for c in range(3):
        pl = np.ascontiguousarray(img.valid(c))
        coefs.append(uhdr.fdct_quant(torch.from_numpy(pl).to("cuda:0"), pl.shape[1], pl.shape[1] // 8, pl.shape[0] // 8, ql if c == 0 else qc))
    sampling = [(2, 2), (1, 1), (1, 1)]
    scan = uhdr.huffman_encode(coefs, w, h, sampling, ri).cpu().numpy().tobytes()
    host = [c.cpu().numpy() for c in coefs]
    assert scan == L.huffman_encode_port(host, w, h, sampling, ri)
    jpeg = uhdr.jpeg_assemble(coefs, w, h, sampling, ri, ql, qc, scan)
    assert jpeg == L.jpeg_assemble_port(host, w, h, sampling, ri, ql, qc, scan)
    if oracle_kind() == "ref":
        ref = L.ref()
        qt = np.zeros((3, 64), dtype=np.uint16)
        bw, bh, nc = (C.c_int * 3)(), (C.c_int * 3)(), C.c_int(0)
        back = [np.zeros_like(c) for c in host]
        ptrs = (C.c_void_p * 3)(*[b.ctypes.data for b in back])
        buf = np.frombuffer(jpeg, dtype=np.uint8)
        assert ref.ref_jpeg_read_coefficients(buf.ctypes.data, buf.size, ptrs, qt.ctypes.data, bw, bh, C.byref(nc)) == 0
        assert nc.value == 3 and list(bw) == [c.shape[1] for c in host] and list(bh) == [c.shape[0] for c in host]
        for c in range(3):
            assert np.array_equal(back[c], host[c]), c
        assert np.array_equal(qt[0], ql) and np.array_equal(qt[1], qc)


@pytest.mark.parametrize("kind", ["sparse", "dense", "worst", "zero"])
def test_huffman_decode_inverts_the_encoder(uhdr, kind):
    """One lane per restart interval: the oracle's streams (byte-identical to libjpeg's) decode to exactly the coefficients
    that went in -- dummy blocks dropped, every sampling layout, restart intervals from 1 MCU to none at all."""
    import torch

    rng = np.random.default_rng(97)
    cases = [(256, 64, [(2, 2), (1, 1), (1, 1)], 10), (72, 40, [(2, 2), (1, 1), (1, 1)], 1), (50, 30, [(2, 2), (1, 1), (1, 1)], 3),
             (41, 23, [(1, 1)] * 3, 21), (45, 21, [(2, 1), (1, 1), (1, 1)], 2), (37, 19, [(1, 1)], 64), (520, 16, [(1, 1)], 7),
             (64, 48, [(2, 2), (1, 1), (1, 1)], 0), (2000, 24, [(1, 1)], 5)]
    for (w, h, sampling, ri) in cases:
        coefs = _random_coefs(rng, w, h, sampling, kind)
        scan = L.huffman_encode_port(coefs, w, h, sampling, ri)
        data = torch.from_numpy(np.frombuffer(scan, dtype=np.uint8).copy()).to("cuda:0")
        got = uhdr.huffman_decode(data, [c.shape[:2] for c in coefs], w, h, sampling, ri)
        for c in range(len(coefs)):
            assert np.array_equal(got[c].cpu().numpy(), coefs[c]), (kind, w, h, ri, c)


def test_huffman_device_round_trip_and_reference_files(uhdr):
    """FDCT -> Huffman encode -> Huffman decode on the device is the identity; and a file written by the REFERENCE encoder
    (no restart markers: one interval), decoded with the tables from its own DHT segments, gives libjpeg's coefficients."""
    import torch

    w, h, ri = 384, 160, 4
    img = synth.make_sdr_yuv420(w, h, align=8, noise=0.1)
    ql, qc = uhdr.quant_table(90, False), uhdr.quant_table(90, True)
    coefs = []
    for c in range(3):
        pl = np.ascontiguousarray(img.valid(c))
        coefs.append(uhdr.fdct_quant(torch.from_numpy(pl).to("cuda:0"), pl.shape[1], pl.shape[1] // 8, pl.shape[0] // 8, ql if c == 0 else qc))
    sampling = [(2, 2), (1, 1), (1, 1)]
    scan = uhdr.huffman_encode(coefs, w, h, sampling, ri)
    back = uhdr.huffman_decode(scan.clone(), [tuple(c.shape[:2]) for c in coefs], w, h, sampling, ri)
    for c in range(3):
        assert torch.equal(back[c], coefs[c]), c
    if oracle_kind() != "ref":
        return
    ref = L.ref()
    out = np.zeros(1 << 22, dtype=np.uint8)
    n = ref.ref_jpeg_compress(C.byref(img.raw), 90, out.ctypes.data, out.size)
    jpeg = out[:n].tobytes()
    # parse the file: DHT tables and the entropy-coded data
    bits, vals = np.zeros((4, 17), np.uint8), np.zeros((4, 256), np.uint8)
    i = 2
    while jpeg[i + 1] != 0xDA:
        m, ln = jpeg[i + 1], (jpeg[i + 2] << 8) | jpeg[i + 3]
        if m == 0xC4:
            seg, j = jpeg[i + 4: i + 2 + ln], 0
            while j < len(seg):
                nsym = sum(seg[j + 1: j + 17])
                t = {0x00: 0, 0x10: 1, 0x01: 2, 0x11: 3}[seg[j]]
                bits[t, 1:] = np.frombuffer(seg[j + 1: j + 17], np.uint8)
                vals[t, :nsym] = np.frombuffer(seg[j + 17: j + 17 + nsym], np.uint8)
                j += 17 + nsym
        i += 2 + ln
    start = i + 2 + ((jpeg[i + 2] << 8) | jpeg[i + 3])
    data = torch.from_numpy(np.frombuffer(jpeg[start:-2], dtype=np.uint8).copy()).to("cuda:0")
    qt = np.zeros((3, 64), dtype=np.uint16)
    bw, bh, nc = (C.c_int * 3)(), (C.c_int * 3)(), C.c_int(0)
    want = [np.zeros(tuple(c.shape), dtype=np.int16) for c in coefs]
    ptrs = (C.c_void_p * 3)(*[b.ctypes.data for b in want])
    buf = np.frombuffer(jpeg, dtype=np.uint8)
    assert ref.ref_jpeg_read_coefficients(buf.ctypes.data, buf.size, ptrs, qt.ctypes.data, bw, bh, C.byref(nc)) == 0
    got = uhdr.huffman_decode(data, [w_.shape[:2] for w_ in want], w, h, sampling, 0, tables=(bits, vals))
    for c in range(3):
        assert np.array_equal(got[c].cpu().numpy(), want[c]), c


def test_huffman_decode_error_behaviour(uhdr):
    import torch

    rng = np.random.default_rng(101)
    w, h, sampling, ri = 128, 32, [(2, 2), (1, 1), (1, 1)], 2
    coefs = _random_coefs(rng, w, h, sampling, "sparse")
    scan = bytearray(L.huffman_encode_port(coefs, w, h, sampling, ri))
    shapes = [c.shape[:2] for c in coefs]

    def dec(buf, ri_=ri):
        return uhdr.huffman_decode(torch.from_numpy(np.frombuffer(bytes(buf), dtype=np.uint8).copy()).to("cuda:0"), shapes, w, h, sampling, ri_)

    with pytest.raises(A.UhdrError) as e:  # the stream has markers every 2 MCUs, not every 4
        dec(scan, 4)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM
    k = scan.find(b"\xff\xd1")
    swapped = bytearray(scan)
    swapped[k + 1] = 0xD5  # marker out of sequence
    with pytest.raises(A.UhdrError) as e:
        dec(swapped)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM
    with pytest.raises(A.UhdrError) as e:  # block grid of another image
        uhdr.huffman_decode(torch.from_numpy(np.frombuffer(bytes(scan), dtype=np.uint8).copy()).to("cuda:0"), shapes, w + 64, h, sampling, ri)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM


@pytest.mark.parametrize("world", [2, 3])
def test_huffman_encode_striped_on_device(uhdr, world):
    """stripes.huffman_encode_striped with the real kernel as the stripe encoder (the ranks run one after the other on
    this GPU): the stitched stream equals the single-launch stream and the oracle's."""
    import torch

    from libultrahdr_amd.stripes import huffman_encode_striped, stitch_entropy_streams

    rng = np.random.default_rng(103)
    for (w, h, sampling, ri) in ((640, 400, [(2, 2), (1, 1), (1, 1)], 5), (333, 203, [(2, 2), (1, 1), (1, 1)], 3), (256, 136, [(1, 1)], 16)):
        host = _random_coefs(rng, w, h, sampling, "sparse")
        dev = [torch.from_numpy(c).to("cuda:0") for c in host]
        enc = lambda part, w_, h_, s_, ri_: uhdr.huffman_encode(part, w_, h_, s_, ri_).cpu().numpy().tobytes()  # noqa: E731
        parts = [huffman_encode_striped(enc, dev, w, h, sampling, ri, r, world) for r in range(world)]
        whole = uhdr.huffman_encode(dev, w, h, sampling, ri).cpu().numpy().tobytes()
        assert stitch_entropy_streams(parts) == whole == L.huffman_encode_port(host, w, h, sampling, ri), (w, h, world)


def test_huffman_encode_error_behaviour(uhdr):
    import torch

    rng = np.random.default_rng(89)
    w, h, sampling = 64, 32, [(2, 2), (1, 1), (1, 1)]
    coefs = [torch.from_numpy(c).to("cuda:0") for c in _random_coefs(rng, w, h, sampling, "sparse")]
    with pytest.raises(A.UhdrError) as e:  # 11 MCUs x 6 blocks > one wavefront
        uhdr.huffman_encode(coefs, w, h, sampling, 11)
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE
    with pytest.raises(A.UhdrError) as e:
        uhdr.huffman_encode(coefs, w, h, sampling, -1)
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE
    with pytest.raises(A.UhdrError) as e:  # block grid of another image
        uhdr.huffman_encode(coefs, w + 64, h, sampling, 4)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM
    small = torch.empty(16, dtype=torch.uint8, device="cuda:0")
    with pytest.raises(A.UhdrError) as e:
        uhdr.huffman_encode(coefs, w, h, sampling, 4, out=small)
    assert e.value.code == A.UHDR_CODEC_MEM_ERROR
    wild = [torch.full_like(c, 32767) for c in coefs]
    for c in wild:
        c[..., 1::2] = -32768
    with pytest.raises(A.UhdrError) as e:  # far outside the baseline range: more bits than a block can have
        uhdr.huffman_encode(wild, w, h, sampling, 10)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM
    with pytest.raises(A.UhdrError) as e:  # the same without restart markers
        uhdr.huffman_encode(wild, w, h, sampling, 0)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM


def test_apply_gainmap_calls_capture_into_a_hip_graph(uhdr):
    """BASELINE config 5 (batch decode to HLG, hipGraph-captured): once the per-metadata tables are
    cached a device-resident applyGainMap call enqueues nothing but its kernel, so a burst of calls on a
    caller-provided stream records into a HIP graph and replays bit-identically."""
    import torch

    w, h, n = 512, 256, 4
    u32 = A.UHDR_IMG_FMT_32bppRGBA1010102
    md = synth.default_metadata()
    sdr = [synth.make_sdr_yuv420(w, h, seed=10 + i).to("cuda:0") for i in range(n)]
    gm = [synth.make_gainmap(w // 4, h // 4, 1, seed=50 + i).to("cuda:0") for i in range(n)]
    dst = [Image(u32, w, h, align=64, device="cuda:0") for _ in range(n)]
    ref = [Image(u32, w, h, align=64, device="cuda:0") for _ in range(n)]
    for i in range(n):  # warm: builds and caches the tables, answers the one-time occupancy queries
        uhdr.applyGainMap(sdr[i], gm[i], md, A.UHDR_CT_HLG, u32, A.FLT_MAX, ref[i])
    uhdr.ctx.synchronize()
    stream = torch.cuda.Stream()
    uhdr.ctx.set_stream(stream.cuda_stream)
    try:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for i in range(n):
                uhdr.applyGainMap(sdr[i], gm[i], md, A.UHDR_CT_HLG, u32, A.FLT_MAX, dst[i])
        for d in dst:
            d.buf.zero_()
        torch.cuda.synchronize()
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a.buf, b.buf) for a, b in zip(dst, ref))
    finally:
        uhdr.ctx.set_stream(None)


def test_copy_raw_image_bit_exact(uhdr):
    """uhdr_hip_copy_raw_image_dev == copy_raw_image (oracle), same formats / repacks / error codes."""
    fmts = [A.UHDR_IMG_FMT_24bppYCbCrP010, A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_8bppYCbCr400,
            A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_64bppRGBAHalfFloat, A.UHDR_IMG_FMT_32bppRGBA1010102,
            A.UHDR_IMG_FMT_24bppRGB888]
    pairs = [(f, f) for f in fmts] + [(A.UHDR_IMG_FMT_24bppRGB888, A.UHDR_IMG_FMT_32bppRGBA8888),
                                      (A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_8bppYCbCr400)]
    rng = np.random.default_rng(31)

    def rand(fmt, w, h, align):
        img = Image(fmt, w, h, A.UHDR_CG_DISPLAY_P3, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=align)
        img.buf[:] = rng.integers(0, 256, img.buf.size, dtype=np.uint8)
        return img

    for sf, df in pairs:
        for (w, h) in ((256, 64), (37, 19)):
            src, dst = rand(sf, w, h, 16), rand(df, w, h, 64)
            want = dst.clone()
            assert L.copy_raw_image("port", src, want) == 0
            got = uhdr.copy_raw_image(src.to("cuda:0"), dst.to("cuda:0"))
            uhdr.ctx.synchronize()
            assert np.array_equal(got.to_host().buf, want.buf), (sf, df, w, h)
            assert (got.raw.cg, got.raw.ct, got.raw.range) == (want.raw.cg, want.raw.ct, want.raw.range)
    with pytest.raises(A.UhdrError) as e:
        uhdr.copy_raw_image(rand(fmts[2], 32, 16, 16).to("cuda:0"), rand(fmts[2], 32, 18, 16).to("cuda:0"))
    assert e.value.code == A.UHDR_CODEC_MEM_ERROR
    with pytest.raises(A.UhdrError) as e:
        uhdr.copy_raw_image(rand(fmts[2], 32, 16, 16).to("cuda:0"), rand(fmts[3], 32, 16, 16).to("cuda:0"))
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE


@pytest.mark.parametrize("base_fmt", [A.UHDR_IMG_FMT_24bppYCbCr444, A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_16bppYCbCr422])
@pytest.mark.parametrize("ch,alpha,scale", [(1, False, 4), (3, False, 1), (3, True, 1), (3, True, 2)])
@pytest.mark.parametrize("out_ct", [A.UHDR_CT_LINEAR, A.UHDR_CT_HLG, A.UHDR_CT_PQ])
def test_apply_gainmap_quad_path_444_and_rgba_bases(uhdr, base_fmt, ch, alpha, scale, out_ct):
    """The quad kernel's BASE 1 (4:4:4, what an API-0 stream decodes to), BASE 2 (RGBA8888) and BASE 3 (4:2:2) variants."""
    w, h = 384, 192
    rng = np.random.default_rng(37)
    sdr = Image(base_fmt, w, h, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
    sdr.buf[:] = rng.integers(0, 256, sdr.buf.size, dtype=np.uint8)
    gm = synth.make_gainmap(w // scale, h // scale, ch, alpha, cg=A.UHDR_CG_BT_2100)
    for use_base_cg in (0, 1):
        md = synth.default_metadata(use_base_cg=use_base_cg, per_channel=(ch == 3))
        check_apply(uhdr, sdr, gm, md, out_ct, what=f"host ubc={use_base_cg}")
    check_apply(uhdr, sdr, gm, synth.default_metadata(), out_ct, device=True, what="device")


@pytest.mark.parametrize("ct,cg", [(A.UHDR_CT_PQ, A.UHDR_CG_BT_2100), (A.UHDR_CT_HLG, A.UHDR_CG_DISPLAY_P3), (A.UHDR_CT_LINEAR, A.UHDR_CG_BT_709)])
@pytest.mark.parametrize("cfg_kw", [dict(preset=A.UHDR_USAGE_REALTIME), dict(preset=A.UHDR_USAGE_BEST_QUALITY),
                                    dict(preset=A.UHDR_USAGE_REALTIME, use_multi_channel_gainmap=0),
                                    dict(preset=A.UHDR_USAGE_BEST_QUALITY, use_multi_channel_gainmap=0, gamma=1.3)])
def test_fused_api0_front_end_equals_the_three_operators(hip_ctx, ct, cg, cfg_kw):
    """uhdr_hip_encode_api0_fused_dev == toneMap -> generateGainMap -> convert_raw_input_to_ycbcr(4:4:4), bit for bit
    (same arithmetic, the 8-bit quantisation between the stages is kept), and within the tone-map tolerance of the
    oracle chain."""
    w, h = 200, 72  # not a multiple of the 256-pixel tile
    hdr = synth.make_hdr_rgba1010102(w, h, ct=ct, cg=cg, noise=0.05)
    cfg = A.default_encode_cfg(use_luminance=0, **cfg_kw)
    u = _uhdr_for(hip_ctx, cfg)
    dh = hdr.to("cuda:0")
    sdr_f, ycc_f, md_f, gm_f = u.encodeApi0Fused(dh, want_sdr_rgba=True, use_luminance=False)
    hip_ctx.synchronize()
    # the three operators on the device
    sdr_s = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w, h, align=64, device="cuda:0")
    u.toneMap(dh, sdr_s)
    md_s, gm_s = u.generateGainMap(sdr_s, dh, False, False)
    ycc_s = u.convert_raw_input_to_ycbcr(sdr_s, False)
    hip_ctx.synchronize()
    assert planes_equal(sdr_f, sdr_s) and planes_equal(ycc_f, ycc_s) and planes_equal(gm_f, gm_s)
    assert md_f.as_dict() == md_s.as_dict()
    assert (ycc_f.raw.fmt, ycc_f.raw.cg, ycc_f.raw.range) == (ycc_s.raw.fmt, ycc_s.raw.cg, ycc_s.raw.range)
    # without the optional RGBA8888 output
    _, ycc_n, md_n, gm_n = u.encodeApi0Fused(dh, want_sdr_rgba=False, use_luminance=False)
    hip_ctx.synchronize()
    assert planes_equal(ycc_n, ycc_s) and planes_equal(gm_n, gm_s) and md_n.as_dict() == md_s.as_dict()
    # oracle chain
    sdr_o = L.tone_map(oracle_kind(), hdr)
    md_o, gm_o = L.generate_gainmap(oracle_kind(), sdr_o, hdr, cfg)
    tol = 1e-4 if cfg.gamma == 1.0 else 5e-3
    assert_close_codes(sdr_f.to_host().valid(0).view(np.uint8), sdr_o.valid(0).view(np.uint8), 1, 1e-4, "fused sdr")
    if np.array_equal(sdr_f.to_host().valid(0), sdr_o.valid(0)):  # same SDR bytes -> the map must agree like generate does
        assert_close_codes(gm_f.to_host().valid(0), gm_o.valid(0), 1, tol, "fused gain map")


@pytest.mark.parametrize("w,h,scale,multi,convert,hints", [(640, 352, 1, True, True, None), (336, 208, 1, True, False, None), (1280, 704, 4, False, True, None),
                                                          (16, 16, 1, True, True, None), (272, 48, 2, False, True, None), (1296, 720, 1, True, True, None),
                                                          (640, 352, 1, True, True, (2.0, 2.0002)), (336, 208, 2, False, True, (1.5, 1.50004))])
def test_api1_fused_chain_equals_the_operators(hip_ctx, w, h, scale, multi, convert, hints):
    """uhdr_hip_encode_api1_fused_dev (pass 1 -> range + tables -> pass 2 fused with rgb->ycc + FDCT; convertYuv fused with the
    base image's three FDCTs) == generateGainMap -> fdct_quant_rgb / fdct_quant, convertYuv -> 3 x fdct_quant: coefficient
    blocks, the 8-bit map and the metadata, bit for bit."""
    import torch

    cfg = A.default_encode_cfg(map_dimension_scale_factor=scale, use_multi_channel_gainmap=int(multi), preset=A.UHDR_USAGE_BEST_QUALITY)
    if hints:  # a range too narrow for a step table: the fused map kernel's per-sample path
        cfg.min_content_boost, cfg.max_content_boost = hints
    u = _uhdr_for(hip_ctx, cfg)
    sdr = synth.make_sdr_yuv420(w, h, seed=w + h)
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG, seed=w * 3 + h)
    ds, dh = sdr.to("cuda:0"), hdr.to("cuda:0")
    ql, qc = L.quant_table_port(95, False), L.quant_table_port(95, True)
    qml, qmc = L.quant_table_port(90, False), L.quant_table_port(90, True)
    enc = A.UHDR_CG_DISPLAY_P3 if convert else A.UHDR_CG_UNSPECIFIED
    base_f, map_f, md_f, gm_f = u.encodeApi1Fused(ds, dh, enc, (ql, qc), (qml, qmc), want_map=True)
    base_n, map_n, md_n, gm_n = u.encodeApi1Fused(ds, dh, enc, (ql, qc), (qml, qmc), want_map=False)
    hip_ctx.synchronize()
    assert gm_n is None
    # the operators
    md_s, gm_s = u.generateGainMap(ds, dh)
    base = ds.clone()
    if convert:
        u.convertYuv(base, ds.raw.cg, A.UHDR_CG_DISPLAY_P3)
    base_s = [u.fdct_quant(base.plane_tensor(i), base.raw.stride[i], (w if i == 0 else w // 2) // 8, (h if i == 0 else h // 2) // 8,
                           ql if i == 0 else qc) for i in range(3)]
    if multi:
        map_s = u.fdct_quant_rgb(gm_s, qml, qmc)
    else:
        map_s = [u.fdct_quant(gm_s.plane_tensor(0), gm_s.raw.stride[0], gm_s.w // 8, gm_s.h // 8, qml)]
    hip_ctx.synchronize()
    assert md_f.as_dict() == md_s.as_dict() == md_n.as_dict()
    assert planes_equal(gm_f, gm_s)
    if hints:
        st = A.Stats()
        hip_ctx.lib.uhdr_hip_get_stats(hip_ctx.handle, C.byref(st))
        assert st.generate_channels_per_sample > 0
    for i in range(3):
        assert torch.equal(base_f[i], base_s[i].reshape(base_f[i].shape)), f"base component {i}"
        assert torch.equal(base_n[i], base_f[i])
    for i in range(len(map_s)):
        assert torch.equal(map_f[i], map_s[i].reshape(map_f[i].shape)), f"map component {i}"
        assert torch.equal(map_n[i], map_f[i])


def test_api1_fused_chain_rejects_what_it_cannot_fuse(hip_ctx):
    sdr, hdr = synth.make_sdr_yuv420(72, 40).to("cuda:0"), synth.make_hdr_p010(72, 40).to("cuda:0")
    q = (L.quant_table_port(90, False), L.quant_table_port(90, True))
    for cfg in (A.default_encode_cfg(), A.default_encode_cfg(preset=A.UHDR_USAGE_REALTIME), A.default_encode_cfg(gamma=1.5)):
        u = _uhdr_for(hip_ctx, cfg)
        a, b = (sdr, hdr) if cfg.preset == A.UHDR_USAGE_BEST_QUALITY and cfg.gamma == 1.0 else (synth.make_sdr_yuv420(64, 32).to("cuda:0"), synth.make_hdr_p010(64, 32).to("cuda:0"))
        with pytest.raises(A.UhdrError) as e:
            u.encodeApi1Fused(a, b, A.UHDR_CG_DISPLAY_P3, q, q)
        assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE


def test_fused_api0_front_end_rejects_what_it_cannot_fuse(hip_ctx):
    u = _uhdr_for(hip_ctx, A.default_encode_cfg(map_dimension_scale_factor=2))
    with pytest.raises(A.UhdrError) as e:
        u.encodeApi0Fused(synth.make_hdr_rgba1010102(64, 32).to("cuda:0"))
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE
    u = _uhdr_for(hip_ctx, A.default_encode_cfg())
    with pytest.raises(A.UhdrError) as e:
        u.encodeApi0Fused(synth.make_hdr_p010(64, 32).to("cuda:0"))
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE


@pytest.mark.parametrize("base_fmt", [A.UHDR_IMG_FMT_24bppYCbCr444, A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_16bppYCbCr422])
def test_apply_gainmap_batch_444_and_rgba_bases(uhdr, base_fmt):
    """Batch launch with the quad kernel's BASE 1 / BASE 2 variants == per-frame oracle results."""
    w, h, n = 320, 96, 3
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    md = synth.default_metadata(use_base_cg=0, per_channel=True)
    rng = np.random.default_rng(41)
    sdrs = []
    for _ in range(n):
        s = Image(base_fmt, w, h, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
        s.buf[:] = rng.integers(0, 256, s.buf.size, dtype=np.uint8)
        sdrs.append(s)
    gms = [synth.make_gainmap(w, h, 3, True, seed=300 + i, cg=A.UHDR_CG_BT_2100) for i in range(n)]
    dests = [Image(f16, w, h, align=2, device="cuda:0") for _ in range(n)]
    uhdr.applyGainMapBatch([s.to("cuda:0") for s in sdrs], [g.to("cuda:0") for g in gms], md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dests)
    uhdr.ctx.synchronize()
    for i in range(n):
        want = L.apply_gainmap(oracle_kind(), sdrs[i], gms[i], md, A.UHDR_CT_LINEAR)
        assert np.array_equal(dests[i].to_host().valid(0), want.valid(0)), i


def test_apply_gainmap_at_the_reference_maximum_dimension(uhdr):
    """8192 x 8192 (UHDR_MAX_DIMENSION, ultrahdrcommon.h): the quad kernel's 32-bit addressing, the
    resident-grid / row-group arithmetic and the batch-free path at the largest image the reference accepts.
    Size-independent checks: a 64-row band at the top and one at the bottom equal the oracle on the same rows
    (scale-1 RGBA map: no cross-row taps), and the same call twice gives the same bytes."""
    import torch

    w = h = 8192
    rng = np.random.default_rng(43)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    sdr = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
    sdr.buf[:] = rng.integers(0, 256, sdr.buf.size, dtype=np.uint8)
    gm = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w, h, A.UHDR_CG_BT_2100)
    gm.buf[:] = rng.integers(0, 256, gm.buf.size, dtype=np.uint8)
    md = synth.default_metadata(use_base_cg=0, per_channel=True)
    dsdr, dgm = sdr.to("cuda:0"), gm.to("cuda:0")
    out1 = Image(f16, w, h, align=64, device="cuda:0")
    out2 = Image(f16, w, h, align=64, device="cuda:0")
    uhdr.applyGainMap(dsdr, dgm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, out1)
    uhdr.applyGainMap(dsdr, dgm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, out2)
    uhdr.ctx.synchronize()
    assert torch.equal(out1.buf, out2.buf)
    got = out1.to_host().valid(0)
    band = 64
    for r0 in (0, h - band):
        s_b = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, band, sdr.raw.cg, sdr.raw.ct, sdr.raw.range)
        s_b.valid(0)[:] = sdr.valid(0)[r0: r0 + band]
        s_b.valid(1)[:] = sdr.valid(1)[r0 // 2: (r0 + band) // 2]
        s_b.valid(2)[:] = sdr.valid(2)[r0 // 2: (r0 + band) // 2]
        g_b = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w, band, gm.raw.cg)
        g_b.valid(0)[:] = gm.valid(0)[r0: r0 + band]
        want = L.apply_gainmap("port", s_b, g_b, md, A.UHDR_CT_LINEAR)
        assert np.array_equal(got[r0: r0 + band], want.valid(0)), r0


@pytest.mark.parametrize("fmt", [A.UHDR_IMG_FMT_24bppRGB888, A.UHDR_IMG_FMT_32bppRGBA8888])
@pytest.mark.parametrize("quality", [95, 60])
def test_fdct_quant_rgb_fused_equals_two_step_route(uhdr, fmt, quality):
    """uhdr_hip_fdct_quant_rgb_dev == libjpeg's rgb_ycc_convert followed by the FDCT of each component (oracle)."""
    rng = np.random.default_rng(47)
    for (w, h) in ((256, 64), (72, 40)):  # second: block count not a multiple of 8
        rgb = Image(fmt, w, h, align=64)
        rgb.buf[:] = rng.integers(0, 256, rgb.buf.size, dtype=np.uint8)
        bpp = 4 if fmt == A.UHDR_IMG_FMT_32bppRGBA8888 else 3
        packed = np.ascontiguousarray(rgb.valid(0).view(np.uint8).reshape(h, -1)[:, : w * bpp])
        rgb888 = packed if bpp == 3 else np.ascontiguousarray(packed.reshape(h, w, 4)[:, :, :3].reshape(h, w * 3))
        planes = L.jpeg_rgb_to_ycc_port(rgb888, w, w, h)
        ql, qc = uhdr.quant_table(quality, False), uhdr.quant_table(quality, True)
        got = uhdr.fdct_quant_rgb(rgb.to("cuda:0"), ql, qc)
        uhdr.ctx.synchronize()
        for c in range(3):
            want = L.fdct_quant_port(planes[c], w, w // 8, h // 8, ql if c == 0 else qc)
            assert np.array_equal(got[c].cpu().numpy(), want), (w, h, c)
    with pytest.raises(A.UhdrError) as e:
        uhdr.fdct_quant_rgb(Image(fmt, 36, 16, align=64, device="cuda:0"), ql, qc)
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE


@pytest.mark.parametrize("fmt", [A.UHDR_IMG_FMT_24bppRGB888, A.UHDR_IMG_FMT_32bppRGBA8888])
@pytest.mark.parametrize("variant", [0, 1])
def test_idct_dequant_rgb_fused_equals_four_step_route(uhdr, fmt, variant):
    """uhdr_hip_idct_dequant_rgb_dev == three IDCTs + libjpeg's ycc_rgb_convert (oracle), incl. images that are not
    a whole number of blocks (cropped stores), strides that break the vector-store alignment, and garbage
    coefficients (32-bit multiply path, range-limit wrap)."""
    import torch

    rng = np.random.default_rng(53)
    bpp = 4 if fmt == A.UHDR_IMG_FMT_32bppRGBA8888 else 3
    for (w, h, align, quality) in ((256, 64, 64, 95), (100, 52, 64, 60), (75, 21, 1, 80)):
        bw, bh = (w + 7) // 8, (h + 7) // 8
        ql, qc = uhdr.quant_table(quality, False), uhdr.quant_table(quality, True)
        src = rng.integers(0, 256, (3, bh * 8, bw * 8), dtype=np.uint8)
        coefs = [L.fdct_quant_port(np.ascontiguousarray(src[c]), bw * 8, bw, bh, ql if c == 0 else qc) for c in range(3)]
        planes = [L.idct_dequant_port(coefs[c], ql if c == 0 else qc) for c in range(3)]
        want = L.jpeg_ycc_to_rgb_port(*[np.ascontiguousarray(pl[:h, :w]) for pl in planes], out_bpp=bpp, variant=variant)
        dst = Image(fmt, w, h, align=align, device="cuda:0")
        out = uhdr.idct_dequant_rgb([torch.from_numpy(c).to("cuda:0") for c in coefs], ql, qc, w, h, fmt, variant, dst=dst)
        uhdr.ctx.synchronize()
        got = out.to_host().valid(0).view(np.uint8).reshape(h, -1)[:, : w * bpp]
        assert np.array_equal(got, want), (w, h)
        # the same through the library's own four-step route
        ycc = Image(A.UHDR_IMG_FMT_24bppYCbCr444, w, h, align=64, device="cuda:0")
        for c in range(3):
            pl = uhdr.idct_dequant(torch.from_numpy(coefs[c]).to("cuda:0"), ql if c == 0 else qc)
            uhdr.ctx.synchronize()  # the library's stream is not torch's
            ycc.plane_tensor(c)[:h, :w] = pl[:h, :w]
        torch.cuda.synchronize()
        four = uhdr.jpeg_ycc_to_rgb(ycc, fmt, variant)
        uhdr.ctx.synchronize()
        assert np.array_equal(four.to_host().valid(0).view(np.uint8).reshape(h, -1)[:, : w * bpp], want)
    wild = [rng.integers(-32768, 32768, (3, 9, 64), dtype=np.int16) for _ in range(3)]
    q255 = np.full(64, 255, dtype=np.uint16)
    planes = [L.idct_dequant_port(c, q255) for c in wild]
    want = L.jpeg_ycc_to_rgb_port(*planes, out_bpp=bpp, variant=variant)
    out = uhdr.idct_dequant_rgb([torch.from_numpy(c).to("cuda:0") for c in wild], q255, q255, 72, 24, fmt, variant)
    uhdr.ctx.synchronize()
    assert np.array_equal(out.to_host().valid(0).view(np.uint8).reshape(24, -1)[:, : 72 * bpp], want)
    with pytest.raises(A.UhdrError) as e:  # block grid must match the image
        uhdr.idct_dequant_rgb([torch.from_numpy(c).to("cuda:0") for c in wild], q255, q255, 64, 24, fmt, variant)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM


@pytest.mark.parametrize("n", [16, 17, 33])
def test_apply_gainmap_batch_chunking(uhdr, n):
    """Batches above 16 frames are issued as several launches (incl. a last launch of a single frame)."""
    w, h = 256, 64
    u32 = A.UHDR_IMG_FMT_32bppRGBA1010102
    md = synth.default_metadata()
    sdrs = [synth.make_sdr_yuv420(w, h, seed=400 + i, noise=0.05) for i in range(n)]
    gms = [synth.make_gainmap(w // 2, h // 2, 1, seed=500 + i) for i in range(n)]
    dests = [Image(u32, w, h, align=4, device="cuda:0") for _ in range(n)]
    uhdr.applyGainMapBatch([s.to("cuda:0") for s in sdrs], [g.to("cuda:0") for g in gms], md, A.UHDR_CT_PQ, u32, A.FLT_MAX, dests)
    uhdr.ctx.synchronize()
    for i in (0, 15, n - 1):
        want = L.apply_gainmap(oracle_kind(), sdrs[i], gms[i], md, A.UHDR_CT_PQ)
        assert np.array_equal(dests[i].to_host().valid(0), want.valid(0)), i
