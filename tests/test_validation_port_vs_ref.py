"""CPU side of tests/test_gpu_validation.py: the same rejection matrices, C port against the real reference
(oracle/_ref) -- pins the port's validation (it stands in for the reference on a box without oracle/_ref) and proves
that no case of the matrix makes the reference read past a buffer before the GPU run does."""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd.images import Image
from oracle import loader as L

import test_gpu_validation as V

needs_ref = pytest.mark.skipif(L.ref() is None, reason="oracle/_ref not built (no /root/reference on this box)")


@needs_ref
@pytest.mark.parametrize("case", V.GENERATE_CASES, ids=[c[0] for c in V.GENERATE_CASES])
def test_generate_gainmap_codes_port_equals_reference(case):
    _, mk_sdr, mk_hdr, _ = case
    cfg = V.default_cfg()
    want, _ = V.ref_code_generate(mk_sdr(), mk_hdr(), cfg)
    sdr, hdr = mk_sdr(), mk_hdr()
    gm = Image(A.UHDR_IMG_FMT_24bppRGB888, sdr.w, sdr.h, align=64)
    md = A.GainmapMetadata()
    got = L.port().uo_generate_gainmap(C.byref(sdr.raw), C.byref(hdr.raw), C.byref(cfg), C.byref(md), C.byref(gm.raw))
    assert got == want
    assert (want != 0) == (not case[0].startswith("accepted"))


@needs_ref
@pytest.mark.parametrize("case", V.TONEMAP_CASES, ids=[c[0] for c in V.TONEMAP_CASES])
def test_tone_map_codes_port_equals_reference(case):
    _, mk_hdr, mk_sdr, _ = case
    want, _ = V.ref_code_tonemap(mk_hdr(), mk_sdr())
    hdr, sdr = mk_hdr(), mk_sdr()
    got = L.port().uo_tone_map(C.byref(hdr.raw), C.byref(sdr.raw))
    assert got == want
    assert (want != 0) == (not case[0].startswith("accepted"))


@needs_ref
@pytest.mark.parametrize("fmt", V.YUV_FMTS)
def test_convert_yuv_codes_port_equals_reference(fmt):
    rng = np.random.default_rng(fmt)
    for src in V.GAMUTS:
        for dst in V.GAMUTS:
            a = Image(fmt, V.W, V.H, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=64)
            a.buf[:] = rng.integers(0, 256, a.nbytes, dtype=np.uint8)
            b = a.clone()
            want, _ = V.ref_code_convert_yuv(a, src, dst)
            got = L.port().uo_convert_yuv(C.byref(b.raw), src, dst)
            assert got == want, (fmt, src, dst)
            if want == 0:
                assert np.array_equal(a.buf, b.buf)
