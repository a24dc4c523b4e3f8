"""libjpeg_variant = 0 (the libjpeg 6b / libjpeg-turbo constants of ycc_rgb_convert -- the reference pins libjpeg-turbo
3.1.0, CMakeLists.txt:519-521) pinned against a REAL libjpeg-turbo: the one inside Pillow (PIL.features says
libjpeg_turbo), which decodes a 4:4:4 JPEG to RGB with turbo's own JDCT_ISLOW IDCT and colour conversion.  The oracle's
restatement (dequantize + islow IDCT + ycc->rgb, variant 0) must reproduce Pillow's pixels exactly; on a GPU box the
device's fused map decode (uhdr_hip_idct_dequant_rgb_dev, variant 0) must too.  oracle/_ref links IJG libjpeg 9d, whose
green-term constants differ for 59 of the 65536 (Cb, Cr) pairs -- that library pins variant 1 (tests/test_oracle_vs_ref.py)."""
import ctypes as C
import io

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from oracle import loader as L

PIL = pytest.importorskip("PIL")
from PIL import Image as PImage, features  # noqa: E402

pytestmark = pytest.mark.skipif(not features.check_feature("libjpeg_turbo"), reason="this Pillow is not built on libjpeg-turbo")


def _jpeg_444(rng, w, h, quality):
    """A 4:4:4 JPEG of noisy, saturated content (out-of-gamut YCbCr triples exercise the clamps) written by Pillow."""
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    a[: h // 2] = (a[: h // 2].astype(np.int32) // 64 * 85).astype(np.uint8)  # flat saturated patches
    buf = io.BytesIO()
    PImage.fromarray(a, "RGB").save(buf, format="JPEG", quality=quality, subsampling=0)
    return buf.getvalue()


def _coefficients(jpeg: bytes, ref):
    data = np.frombuffer(jpeg, dtype=np.uint8)
    qt = np.zeros((3, 64), dtype=np.uint16)
    bw, bh, nc = (C.c_int * 3)(), (C.c_int * 3)(), C.c_int(0)
    null = (C.c_void_p * 3)(None, None, None)
    assert ref.ref_jpeg_read_coefficients(data.ctypes.data, data.size, null, qt.ctypes.data, bw, bh, C.byref(nc)) == 0
    coefs = [np.zeros((bh[c], bw[c], 64), dtype=np.int16) for c in range(3)]
    ptrs = (C.c_void_p * 3)(*[c.ctypes.data for c in coefs])
    assert ref.ref_jpeg_read_coefficients(data.ctypes.data, data.size, ptrs, qt.ctypes.data, bw, bh, C.byref(nc)) == 0
    return coefs, qt


@pytest.mark.parametrize("quality", [95, 75, 30])
def test_oracle_variant0_equals_pillows_libjpeg_turbo(ref, quality):
    rng = np.random.default_rng(307 + quality)
    w, h = 200, 136
    jpeg = _jpeg_444(rng, w, h, quality)
    want = np.asarray(PImage.open(io.BytesIO(jpeg)).convert("RGB"))  # libjpeg-turbo: islow IDCT + ycc_rgb_convert
    coefs, qt = _coefficients(jpeg, ref)  # entropy decode only (any libjpeg gives the same coefficients)
    planes = [L.idct_dequant_port(coefs[c], qt[c])[:h, :w] for c in range(3)]
    got = L.jpeg_ycc_to_rgb_port(*[np.ascontiguousarray(p) for p in planes], out_bpp=3, variant=0).reshape(h, w, 3)
    assert np.array_equal(got, want), int((got != want).sum())
    other = L.jpeg_ycc_to_rgb_port(*[np.ascontiguousarray(p) for p in planes], out_bpp=3, variant=1).reshape(h, w, 3)
    assert np.abs(other.astype(int) - want.astype(int)).max() <= 1  # the IJG 9 constants differ by at most one code, rarely


@pytest.mark.gpu
@pytest.mark.parametrize("quality", [95, 50])
def test_device_map_decode_variant0_equals_pillows_libjpeg_turbo(hip_ctx, quality):
    import torch

    from libultrahdr_amd.ultrahdr import UltraHdr

    ref = L.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    u = UltraHdr(ctx=hip_ctx)
    rng = np.random.default_rng(311 + quality)
    w, h = 512, 256
    jpeg = _jpeg_444(rng, w, h, quality)
    want = np.asarray(PImage.open(io.BytesIO(jpeg)).convert("RGB"))
    coefs, qt = _coefficients(jpeg, ref)
    dev = [torch.from_numpy(c).to("cuda:0") for c in coefs]
    out = u.idct_dequant_rgb(dev, qt[0], qt[1], w, h, A.UHDR_IMG_FMT_24bppRGB888, 0)
    hip_ctx.synchronize()
    got = out.to_host().valid(0).view(np.uint8).reshape(h, -1)[:, : w * 3].reshape(h, w, 3)
    assert np.array_equal(got, want), int((got != want).sum())
