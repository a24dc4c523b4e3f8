"""Committed golden vectors (tests/golden/hotpath_golden.npz, produced from the real reference by
tests/golden/make_golden.py).  CPU part: the C oracle reproduces every vector bit for bit (this is
what pins the oracle on machines without /root/reference).  GPU part (-m gpu): the HIP path
against the same vectors."""
import ctypes as C
import os

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd.images import Image
from oracle import loader as L
import golden_cases as G

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz"))
CASES = G.cases()


def gold(name):
    pfx = name + "/"
    return {k[len(pfx):]: GOLD[k] for k in GOLD.files if k.startswith(pfx)}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_reference_vectors(name):
    got, want = G.run(CASES[name], "port"), gold(name)
    assert set(got) == set(want)
    for k in want:
        assert np.array_equal(got[k], want[k]), (name, k)


@pytest.mark.parametrize("q", [95, 50])
def test_oracle_dct_reproduces_libjpeg_coefficients(q):
    img = G.jpeg_image()
    for c in range(3):
        qt = L.quant_table_port(q, c > 0)
        assert np.array_equal(qt, GOLD[f"jpeg_q{q}/qt{c}"])
        want = GOLD[f"jpeg_q{q}/coef{c}"]
        plane = img.plane(c)
        got = L.fdct_quant_port(plane, plane.shape[1], want.shape[1], want.shape[0], qt)
        assert np.array_equal(got, want)


@pytest.mark.parametrize("q", [95, 50])
def test_oracle_decode_stage_reproduces_libjpeg(q):
    """dequant + IDCT (+ colour conversion for the RGB map) == what the reference's JpegDecoderHelper got
    from libjpeg for the same JPEGs (vectors generated in the build container, IJG libjpeg 9d)."""
    for c in range(3):
        got = L.idct_dequant_port(GOLD[f"jpeg_q{q}/coef{c}"], GOLD[f"jpeg_q{q}/qt{c}"])
        want = GOLD[f"jpeg_q{q}/dec{c}"]
        assert np.array_equal(got[: want.shape[0], : want.shape[1]], want)
    gm = G.jpeg_rgb_map()
    planes = L.jpeg_rgb_to_ycc_port(gm.valid(0), gm.w, gm.w, gm.h)
    for c in range(3):
        qt = GOLD[f"jpegrgb_q{q}/qt{c}"]
        assert np.array_equal(L.fdct_quant_port(planes[c], gm.w, gm.w // 8, gm.h // 8, qt), GOLD[f"jpegrgb_q{q}/coef{c}"])
    dec = [L.idct_dequant_port(GOLD[f"jpegrgb_q{q}/coef{c}"], GOLD[f"jpegrgb_q{q}/qt{c}"])[: gm.h, : gm.w] for c in range(3)]
    bpp = int(GOLD[f"jpegrgb_q{q}/dec_bpp"][0])
    assert np.array_equal(L.jpeg_ycc_to_rgb_port(*dec, out_bpp=bpp, variant=1), GOLD[f"jpegrgb_q{q}/dec_rgb"])


# ---- GPU ------------------------------------------------------------------------------------------
EXACT_OPS = {"convert_yuv", "raw2ycc"}


def _close(got, want, what):
    if got.dtype == np.uint32:  # packed 8888 / 1010102
        if "1010102" in what or "hlg" in what or "pq" in what:
            g = np.stack([(got >> s) & 0x3FF for s in (0, 10, 20)], -1).astype(np.int64)
            w = np.stack([(want >> s) & 0x3FF for s in (0, 10, 20)], -1).astype(np.int64)
        else:
            g, w = got.view(np.uint8).astype(np.int64), want.view(np.uint8).astype(np.int64)
    else:
        g, w = got.astype(np.int64), want.astype(np.int64)
    d = np.abs(g - w)
    assert d.max() <= 1 and (d != 0).mean() <= 1e-4, (what, int(d.max()), float((d != 0).mean()))


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_hip_reproduces_reference_vectors(hip_ctx, name):
    from libultrahdr_amd.ultrahdr import UltraHdr

    case, want = CASES[name], gold(name)
    op = case["op"]
    if op == "apply":
        sdr, gm, md = G.inputs(case)
        fmt = A.UHDR_IMG_FMT_64bppRGBAHalfFloat if case["ct"] == A.UHDR_CT_LINEAR else A.UHDR_IMG_FMT_32bppRGBA1010102
        dest = Image(fmt, sdr.w, sdr.h, align=1)
        UltraHdr(ctx=hip_ctx).applyGainMap(sdr, gm, md, case["ct"], fmt, A.FLT_MAX, dest)
        if case["ct"] == A.UHDR_CT_HLG:
            _close(dest.valid(0), want["plane0"], name)
        else:
            assert np.array_equal(dest.valid(0), want["plane0"]), name
        return
    if op == "gen":
        sdr, hdr = G.inputs(case)
        cfg = A.default_encode_cfg(**case["cfg"])
        u = UltraHdr(ctx=hip_ctx, mapDimensionScaleFactor=cfg.map_dimension_scale_factor,
                     useMultiChannelGainMap=bool(cfg.use_multi_channel_gainmap), gamma=cfg.gamma, preset=cfg.preset)
        md, gm = u.generateGainMap(sdr, hdr, bool(cfg.sdr_is_601), bool(cfg.use_luminance))
        _close(gm.valid(0), want["plane0"], name)
        ref_md = A.GainmapMetadata.from_buffer_copy(want["metadata"].tobytes())
        for k, v in ref_md.as_dict().items():
            assert np.allclose(md.as_dict()[k], v, rtol=1e-5), (name, k)
        return
    u = UltraHdr(ctx=hip_ctx)
    if op == "tonemap":
        _, hdr = G.inputs(case)
        fmt = A.UHDR_IMG_FMT_12bppYCbCr420 if case["hdr"] == "p010" else A.UHDR_IMG_FMT_32bppRGBA8888
        out = Image(fmt, hdr.w, hdr.h, align=64)
        u.toneMap(hdr, out)
    elif op == "convert_yuv":
        (out,) = G.inputs(case)
        u.convertYuv(out, case["src"], case["dst"])
    else:
        (img,) = G.inputs(case)
        out = u.convert_raw_input_to_ycbcr(img, case["chroma"])
    for i, p in enumerate(out.planes_valid()):
        if op in EXACT_OPS:
            assert np.array_equal(p, want[f"plane{i}"]), (name, i)
        else:
            _close(p, want[f"plane{i}"], name)


@pytest.mark.parametrize("q", [95, 50])
def test_oracle_entropy_stage_reproduces_the_reference_encoders_bytes(q):
    """Huffman restatement: the golden coefficients -> exactly the entropy-coded bytes the reference encoder wrote
    (no restart markers), and those bytes -> the golden coefficients."""
    for tag, w, h, sampling in (("jpeg", G.W, G.H, [(2, 2), (1, 1), (1, 1)]), ("jpegrgb", 96, 48, [(1, 1)] * 3)):
        coefs = [GOLD[f"{tag}_q{q}/coef{c}"] for c in range(3)]
        scan = GOLD[f"{tag}_q{q}/scan"].tobytes()
        assert L.huffman_encode_port(coefs, w, h, sampling, 0) == scan
        rc, back = L.huffman_decode_port([c.shape[:2] for c in coefs], w, h, sampling, 0, scan)
        assert rc == 0 and all(np.array_equal(b, c) for b, c in zip(back, coefs))


@pytest.mark.gpu
@pytest.mark.parametrize("q", [95, 50])
def test_hip_dct_reproduces_libjpeg_coefficients(hip_ctx, q):
    from libultrahdr_amd.ultrahdr import UltraHdr

    u = UltraHdr(ctx=hip_ctx)
    img = G.jpeg_image()
    for c in range(3):
        qt = u.quant_table(q, c > 0)
        assert np.array_equal(qt, GOLD[f"jpeg_q{q}/qt{c}"])
        want = GOLD[f"jpeg_q{q}/coef{c}"]
        plane = np.ascontiguousarray(img.plane(c))
        got = u.fdct_quant(plane, plane.shape[1], want.shape[1], want.shape[0], qt)
        assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("q", [95, 50])
def test_hip_decode_stage_reproduces_libjpeg(hip_ctx, q):
    from libultrahdr_amd.ultrahdr import UltraHdr

    u = UltraHdr(ctx=hip_ctx)
    for c in range(3):
        got = u.idct_dequant(GOLD[f"jpeg_q{q}/coef{c}"], GOLD[f"jpeg_q{q}/qt{c}"])
        want = GOLD[f"jpeg_q{q}/dec{c}"]
        assert np.array_equal(got[: want.shape[0], : want.shape[1]], want)
    gm = G.jpeg_rgb_map()
    ycc = u.jpeg_rgb_to_ycc(gm)
    for c in range(3):
        qt = GOLD[f"jpegrgb_q{q}/qt{c}"]
        plane = np.ascontiguousarray(ycc.plane(c))
        assert np.array_equal(u.fdct_quant(plane, plane.shape[1], gm.w // 8, gm.h // 8, qt), GOLD[f"jpegrgb_q{q}/coef{c}"])
    dec = Image(A.UHDR_IMG_FMT_24bppYCbCr444, gm.w, gm.h, align=8)
    for c in range(3):
        dec.valid(c)[:] = u.idct_dequant(GOLD[f"jpegrgb_q{q}/coef{c}"], GOLD[f"jpegrgb_q{q}/qt{c}"])[: gm.h, : gm.w]
    bpp = int(GOLD[f"jpegrgb_q{q}/dec_bpp"][0])
    fmt = A.UHDR_IMG_FMT_32bppRGBA8888 if bpp == 4 else A.UHDR_IMG_FMT_24bppRGB888
    rgb = u.jpeg_ycc_to_rgb(dec, fmt, libjpeg_variant=1)
    got = rgb.valid(0).view(np.uint8).reshape(gm.h, -1)[:, : gm.w * bpp]
    assert np.array_equal(got, GOLD[f"jpegrgb_q{q}/dec_rgb"])


# ---- the reference's own 1280x720 raw fixture (BASELINE config 1's inputs) ------------------------------------------
def test_fixture_720p_oracle_reproduces_the_reference():
    """C oracle == real reference on tests/data/raw_p010_image.p010 + raw_yuv420_image.yuv420: generateGainMap (two
    pass, 3 channels), convertYuv, FDCT + quantize of the base image and of the map (libjpeg's own coefficients),
    applyGainMap to all three output transfers."""
    import fixture720 as F

    g = F.gold()
    sdr, hdr = F.inputs()
    md, gm = L.generate_gainmap("port", sdr, hdr, A.default_encode_cfg())
    assert np.array_equal(gm.valid(0), g["gainmap"])
    assert md.as_dict() == F.metadata().as_dict()
    conv = L.convert_yuv("port", sdr, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3)
    for c in range(3):
        assert np.array_equal(conv.valid(c), g[f"sdr601_{c}"])
        want = g[f"base_coef{c}"]
        qt = L.quant_table_port(95, c > 0)
        assert np.array_equal(qt, g[f"base_qt{c}"])
        # the encoder pads planes to the MCU grid by edge replication (jpegencoderhelper.cpp:246-309); 1280x720 is MCU aligned
        plane = np.ascontiguousarray(conv.valid(c))
        assert np.array_equal(L.fdct_quant_port(plane, plane.shape[1], want.shape[1], want.shape[0], qt), want)
    planes = L.jpeg_rgb_to_ycc_port(np.ascontiguousarray(gm.valid(0)), gm.w, gm.w, gm.h)
    for c in range(3):
        want = g[f"map_coef{c}"]
        assert np.array_equal(L.fdct_quant_port(planes[c], gm.w, want.shape[1], want.shape[0], g[f"map_qt{c}"]), want)
    for name, ct in (("linear", A.UHDR_CT_LINEAR), ("hlg", A.UHDR_CT_HLG), ("pq", A.UHDR_CT_PQ)):
        res = L.apply_gainmap("port", sdr, F.gainmap(), F.metadata(), ct)
        assert np.array_equal(res.valid(0)[::45], g[f"apply_{name}_rows"]), name
        assert F.crc(res.valid(0)) == int(g[f"apply_{name}_crc"][0]), name
