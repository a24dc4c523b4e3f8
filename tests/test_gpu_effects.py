"""GPU: the effects chain of the reference (lib/src/editorhelper.cpp:20-87, 210-520: rotate / mirror / crop / resize) as
element remaps on the device -- against the remap formulas themselves (numpy) for every raw format, and through the
facade's uhdr_add_effect_* with uhdr_enable_gpu_acceleration against the reference's CPU code."""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image

pytestmark = pytest.mark.gpu

FMTS = [A.UHDR_IMG_FMT_24bppYCbCrP010, A.UHDR_IMG_FMT_12bppYCbCr420, A.UHDR_IMG_FMT_8bppYCbCr400, A.UHDR_IMG_FMT_24bppYCbCr444,
        A.UHDR_IMG_FMT_30bppYCbCr444, A.UHDR_IMG_FMT_32bppRGBA8888, A.UHDR_IMG_FMT_32bppRGBA1010102, A.UHDR_IMG_FMT_64bppRGBAHalfFloat]


def _elements(img: Image, c: int) -> np.ndarray:
    """Plane c as the reference walks it: one array element per remapped element (P010 chroma: (U, V) pairs as uint32,
    RGBA-F16 pixels as uint64)."""
    a = img.valid(c)
    if img.fmt == A.UHDR_IMG_FMT_24bppYCbCrP010 and c == 1:
        return np.ascontiguousarray(a).view(np.uint32)
    if img.fmt == A.UHDR_IMG_FMT_64bppRGBAHalfFloat:
        return np.ascontiguousarray(a).view(np.uint64)
    return a


def _remap(a: np.ndarray, effect, p0, p1, dw, dh) -> np.ndarray:
    if effect == 0:
        return {90: np.rot90(a, -1), 180: a[::-1, ::-1], 270: np.rot90(a, 1)}[p0]
    if effect == 1:
        return a[::-1] if p0 == 0 else a[:, ::-1]
    if effect == 2:
        return a[p1: p1 + dh, p0: p0 + dw]
    fy, fx = a.shape[0] // dh, a.shape[1] // dw
    return a[np.arange(dh) * fy][:, np.arange(dw) * fx]


@pytest.mark.parametrize("fmt", FMTS)
def test_effects_equal_the_reference_remaps(hip_ctx, fmt):
    w, h = 208, 144
    rng = np.random.default_rng(401 + fmt)
    src = Image(fmt, w, h, align=64)
    src.buf[:] = rng.integers(0, 256, src.buf.size, dtype=np.uint8)
    cases = [(0, 90, 0, h, w), (0, 180, 0, w, h), (0, 270, 0, h, w), (1, 0, 0, w, h), (1, 1, 0, w, h), (2, 16, 8, 96, 64), (2, 0, 0, w, h),
             (3, 0, 0, 104, 72), (3, 0, 0, 64, 48), (3, 0, 0, 208, 144)]
    lib = hip_ctx.lib
    for (effect, p0, p1, dw, dh) in cases:
        for dev in (False, True):
            dst = Image(fmt, dw, dh, align=64, device="cuda:0") if dev else Image(fmt, dw, dh, align=64)
            s = src.to("cuda:0") if dev else src
            fn = lib.uhdr_hip_apply_effect_dev if dev else lib.uhdr_hip_apply_effect
            A.check(fn(hip_ctx.handle, effect, p0, p1, C.byref(s.raw), C.byref(dst.raw)))
            hip_ctx.synchronize()
            got = dst.to_host() if dev else dst
            for c in range(len(src.layout)):
                if src.layout[c] is None:
                    continue
                a = _elements(src, c)
                div = src.w // a.shape[1] if fmt != A.UHDR_IMG_FMT_24bppYCbCrP010 or c == 0 else 2  # chroma planes: half size
                if fmt == A.UHDR_IMG_FMT_64bppRGBAHalfFloat:
                    div = 1
                want = _remap(a, effect, p0 // div if effect == 2 else p0, p1 // div if effect == 2 else p1, dw // div, dh // div)
                assert np.array_equal(_elements(got, c), want), (fmt, effect, p0, p1, dw, dh, c, dev)


def test_effect_argument_errors(hip_ctx):
    src = Image(A.UHDR_IMG_FMT_8bppYCbCr400, 64, 32, align=64)
    lib = hip_ctx.lib
    for (effect, p0, p1, dw, dh) in [(0, 45, 0, 64, 32), (0, 90, 0, 64, 32), (1, 2, 0, 64, 32), (2, 40, 0, 32, 32), (2, -1, 0, 8, 8), (7, 0, 0, 64, 32)]:
        dst = Image(A.UHDR_IMG_FMT_8bppYCbCr400, dw, dh, align=64)
        st = lib.uhdr_hip_apply_effect(hip_ctx.handle, effect, p0, p1, C.byref(src.raw), C.byref(dst.raw))
        assert st.error_code == A.UHDR_CODEC_INVALID_PARAM, (effect, p0, p1)


def test_effects_through_the_facade_equal_the_cpu_reference():
    from libultrahdr_amd import facade as F
    from tests import fixture720

    if not F.available():
        pytest.skip("facade not built")
    sdr, hdr = fixture720.inputs()
    hdr.raw.cg = A.UHDR_CG_BT_2100
    jpg = F.encode(hdr, sdr, gpu=False, preset=A.UHDR_USAGE_REALTIME)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    chain = [("rotate", 90), ("mirror", 1), ("crop", 40, 500, 64, 900), ("resize", 230, 418)]
    a = F.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=False, effects=chain)
    b = F.decode(jpg, A.UHDR_CT_LINEAR, f16, gpu=True, effects=chain)
    assert a.shape == b.shape and np.array_equal(a, b)
    # encoder side: the effects run on the raw intents before the encode
    e_cpu = F.encode(hdr, sdr, gpu=False, preset=A.UHDR_USAGE_REALTIME, effects=[("rotate", 180), ("crop", 0, 1024, 0, 512)])
    e_gpu = F.encode(hdr, sdr, gpu=True, preset=A.UHDR_USAGE_REALTIME, effects=[("rotate", 180), ("crop", 0, 1024, 0, 512)])
    assert e_cpu == e_gpu
