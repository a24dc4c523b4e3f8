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


def test_random_combinations_of_bad_descriptor_fields_port_vs_reference(monkeypatch):
    """The fuzz sweep's `generate-error` / `tonemap-error` arm (tests/fuzz_parity.py::fuzz_encode_errors: several poisoned fields at
    once, so the ORDER of the checks matters) with the C port in the HIP path's place: 400 random descriptor pairs, the reference's code."""
    import ctypes as C

    import fuzz_parity as FZ
    import test_gpu_validation as V

    if L.ref() is None:
        pytest.skip("oracle/_ref not built")

    def port_generate(ctx, sdr, hdr, cfg, device):
        gm = Image(A.UHDR_IMG_FMT_24bppRGB888, max(sdr.w, 1), max(sdr.h, 1), align=64)
        md = A.GainmapMetadata()
        return L.port().uo_generate_gainmap(C.byref(sdr.raw), C.byref(hdr.raw), C.byref(cfg), C.byref(md), C.byref(gm.raw)), ""

    monkeypatch.setattr(V, "hip_code_generate", port_generate)
    monkeypatch.setattr(V, "hip_code_tonemap", lambda ctx, hdr, sdr, device: (L.port().uo_tone_map(C.byref(hdr.raw), C.byref(sdr.raw)), ""))
    monkeypatch.setattr(FZ, "rng", np.random.default_rng(77))
    monkeypatch.setattr(FZ, "stats", {})
    monkeypatch.setattr(FZ, "bad", 0)
    for _ in range(400):
        FZ.fuzz_encode_errors()
    assert FZ.bad == 0, FZ.stats
    assert FZ.stats["generate-error"][0] > 100 and FZ.stats["tonemap-error"][0] > 100
