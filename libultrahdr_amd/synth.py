"""Deterministic synthetic inputs of SURVEY.md section 8(d): a smooth luminance field plus noise,
rendered as an HDR image (P010 limited-range BT.2100 HLG/PQ, or RGBA1010102 PQ) and an SDR image
(YCbCr 4:2:0 full-range BT.709 sRGB).  numpy only (seed 1234, PCG64) so the same bytes are
produced in the build container and on the GPU box."""
from __future__ import annotations

import numpy as np

from . import capi as A
from .images import Image

SEED = 1234


def _field(w, h, rng, noise=0.02):
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    f = 0.5 + 0.5 * np.sin(x / 97.0) * np.cos(y / 61.0)
    if noise:
        f = f + rng.normal(0.0, noise, size=(h, w))
    return np.clip(f, 0.0, 1.0)


def _chroma(w, h, phase):
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    return 0.5 + 0.12 * np.sin(x / 53.0 + phase) * np.sin(y / 71.0 + 2 * phase)


def make_sdr_yuv420(w, h, seed=SEED, cg=A.UHDR_CG_BT_709, align=64, noise=0.02) -> Image:
    rng = np.random.default_rng(seed)
    luma = _field(w, h, rng, noise)
    img = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, cg, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align)
    img.valid(0)[:] = np.clip(np.rint(255.0 * luma ** 0.8), 0, 255).astype(np.uint8)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    img.valid(1)[:] = np.clip(np.rint(255.0 * _chroma(cw, ch, 0.3)), 0, 255).astype(np.uint8)
    img.valid(2)[:] = np.clip(np.rint(255.0 * _chroma(cw, ch, 1.1)), 0, 255).astype(np.uint8)
    return img


def make_hdr_p010(w, h, seed=SEED, ct=A.UHDR_CT_HLG, cg=A.UHDR_CG_BT_2100, align=64, noise=0.02,
                  rng_range=A.UHDR_CR_LIMITED_RANGE) -> Image:
    rng = np.random.default_rng(seed)
    luma = _field(w, h, rng, noise)
    img = Image(A.UHDR_IMG_FMT_24bppYCbCrP010, w, h, cg, ct, rng_range, align)
    if rng_range == A.UHDR_CR_LIMITED_RANGE:
        yv = 64 + np.rint(876.0 * luma)
        cscale, coff = 896.0, 64.0
    else:
        yv = np.rint(1023.0 * luma)
        cscale, coff = 1023.0, 0.0
    img.valid(0)[:] = (np.clip(yv, 0, 1023).astype(np.uint16) << 6)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    u = np.clip(coff + np.rint(cscale * _chroma(cw, ch, 0.7)), 0, 1023).astype(np.uint16) << 6
    v = np.clip(coff + np.rint(cscale * _chroma(cw, ch, 1.9)), 0, 1023).astype(np.uint16) << 6
    uv = img.valid(1)
    uv[:, 0::2] = u
    uv[:, 1::2] = v
    return img


def make_hdr_rgba1010102(w, h, seed=SEED, ct=A.UHDR_CT_PQ, cg=A.UHDR_CG_BT_2100, align=64, noise=0.02) -> Image:
    rng = np.random.default_rng(seed)
    base = _field(w, h, rng, noise)
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    r = np.clip(base * (0.85 + 0.15 * np.sin(x / 41.0)), 0, 1)
    g = np.clip(base * (0.85 + 0.15 * np.cos(y / 37.0)), 0, 1)
    b = np.clip(base * (0.85 + 0.15 * np.sin((x + y) / 59.0)), 0, 1)
    q = lambda c: np.rint(1023.0 * c).astype(np.uint32)
    img = Image(A.UHDR_IMG_FMT_32bppRGBA1010102, w, h, cg, ct, A.UHDR_CR_FULL_RANGE, align)
    img.valid(0)[:] = q(r) | (q(g) << 10) | (q(b) << 20) | (np.uint32(3) << 30)
    return img


def make_sdr_rgba8888(w, h, seed=SEED, cg=A.UHDR_CG_BT_709, align=64, noise=0.02) -> Image:
    rng = np.random.default_rng(seed)
    base = _field(w, h, rng, noise) ** 0.8
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    q = lambda c: np.clip(np.rint(255.0 * c), 0, 255).astype(np.uint32)
    r = q(base * (0.9 + 0.1 * np.sin(x / 43.0)))
    g = q(base * (0.9 + 0.1 * np.cos(y / 31.0)))
    b = q(base * (0.9 + 0.1 * np.sin((x - y) / 67.0)))
    img = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w, h, cg, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align)
    img.valid(0)[:] = r | (g << 8) | (b << 16) | (np.uint32(255) << 24)
    return img


def make_gainmap(w, h, channels=1, alpha=False, seed=SEED + 1, cg=A.UHDR_CG_UNSPECIFIED, align=64) -> Image:
    """A plausible gain map: smooth field + noise, 8 bit, Y400 / RGB888 / RGBA8888."""
    rng = np.random.default_rng(seed)
    if channels == 1:
        img = Image(A.UHDR_IMG_FMT_8bppYCbCr400, w, h, cg, align=align)
        img.valid(0)[:] = np.clip(np.rint(255.0 * _field(w, h, rng, 0.03)), 0, 255).astype(np.uint8)
        return img
    planes = [np.clip(np.rint(255.0 * _field(w, h, np.random.default_rng(seed + k), 0.03) * (1.0 - 0.1 * k)), 0, 255)
              .astype(np.uint8) for k in range(3)]
    if alpha:
        img = Image(A.UHDR_IMG_FMT_32bppRGBA8888, w, h, cg, align=align)
        v = planes[0].astype(np.uint32) | (planes[1].astype(np.uint32) << 8) | (planes[2].astype(np.uint32) << 16) | (np.uint32(255) << 24)
        img.valid(0)[:] = v
    else:
        img = Image(A.UHDR_IMG_FMT_24bppRGB888, w, h, cg, align=align)
        out = img.valid(0)
        out[:, 0::3], out[:, 1::3], out[:, 2::3] = planes
    return img


def default_metadata(max_boost=4.926108, min_boost=1.0, gamma=1.0, offset=1e-7, use_base_cg=1,
                     per_channel=False) -> A.GainmapMetadata:
    md = A.GainmapMetadata()
    for i in range(3):
        k = (1.0 - 0.07 * i) if per_channel else 1.0
        md.max_content_boost[i] = max_boost * k
        md.min_content_boost[i] = min_boost
        md.gamma[i] = gamma
        md.offset_sdr[i] = offset
        md.offset_hdr[i] = offset
    md.hdr_capacity_min = 1.0
    md.hdr_capacity_max = max_boost
    md.use_base_cg = use_base_cg
    return md


def checksum(img: Image) -> int:
    """Order-dependent 64-bit checksum of the valid samples (for 'checksum of checksums' tests)."""
    import zlib

    acc = 0
    for p in img.to_host().planes_valid():
        acc = zlib.crc32(np.ascontiguousarray(p).tobytes(), acc)
    return acc


# ---- the remaining input formats the encode operators accept (gainmapmath.cpp:398-492) -----------------
def make_sdr_planar(fmt, w, h, seed=SEED, cg=A.UHDR_CG_BT_709, align=64, noise=0.02) -> Image:
    """YCbCr 4:2:2 / 4:4:4 (or 4:2:0) SDR rendition of the same field as make_sdr_yuv420."""
    rng = np.random.default_rng(seed)
    luma = _field(w, h, rng, noise)
    img = Image(fmt, w, h, cg, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align)
    img.valid(0)[:] = np.clip(np.rint(255.0 * luma ** 0.8), 0, 255).astype(np.uint8)
    ch, cw = img.valid(1).shape
    sx, sy = w / cw, h / ch
    x = (np.arange(cw, dtype=np.float64) * sx / 2.0)[None, :]
    y = (np.arange(ch, dtype=np.float64) * sy / 2.0)[:, None]
    for pl, ph in ((1, 0.3), (2, 1.1)):
        c = 0.5 + 0.12 * np.sin(x / 53.0 + ph) * np.sin(y / 71.0 + 2 * ph)
        img.valid(pl)[:] = np.clip(np.rint(255.0 * c + rng.normal(0.0, 255.0 * noise / 2, c.shape)), 0, 255).astype(np.uint8)
    return img


def make_hdr_yuv444_10bit(w, h, seed=SEED, ct=A.UHDR_CT_PQ, cg=A.UHDR_CG_BT_2100, align=64, noise=0.02,
                          rng_range=A.UHDR_CR_LIMITED_RANGE) -> Image:
    """UHDR_IMG_FMT_30bppYCbCr444: three planes of 10-bit samples in the LOW bits of uint16 (gainmapmath.cpp:398-420)."""
    rng = np.random.default_rng(seed)
    luma = _field(w, h, rng, noise)
    img = Image(A.UHDR_IMG_FMT_30bppYCbCr444, w, h, cg, ct, rng_range, align)
    if rng_range == A.UHDR_CR_LIMITED_RANGE:
        yv, cscale, coff = 64 + np.rint(876.0 * luma), 896.0, 64.0
    else:
        yv, cscale, coff = np.rint(1023.0 * luma), 1023.0, 0.0
    img.valid(0)[:] = np.clip(yv, 0, 1023).astype(np.uint16)
    for pl, ph in ((1, 0.7), (2, 1.9)):
        c = _chroma(w, h, ph) + rng.normal(0.0, noise / 2, (h, w))
        img.valid(pl)[:] = np.clip(coff + np.rint(cscale * c), 0, 1023).astype(np.uint16)
    return img


def make_hdr_rgba_f16(w, h, seed=SEED, cg=A.UHDR_CG_BT_2100, align=64, noise=0.02, peak=20.0, specials=True) -> Image:
    """UHDR_IMG_FMT_64bppRGBAHalfFloat, linear light (1.0 = SDR white).  specials: a sprinkle of +inf, -inf, NaN,
    negative, sub-normal and over-range values -- what getRgbaF16Pixel's sanitizePixel exists for (gainmapmath.cpp:483-492)."""
    rng = np.random.default_rng(seed)
    base = _field(w, h, rng, noise) ** 2.2 * peak
    x = np.arange(w, dtype=np.float64)[None, :]
    y = np.arange(h, dtype=np.float64)[:, None]
    px = np.empty((h, w, 4), dtype=np.float16)
    px[..., 0] = base * (0.85 + 0.15 * np.sin(x / 41.0))
    px[..., 1] = base * (0.85 + 0.15 * np.cos(y / 37.0))
    px[..., 2] = base * (0.85 + 0.15 * np.sin((x + y) / 59.0))
    px[..., 3] = 1.0
    if specials:
        bits = px.view(np.uint16)
        n = max(8, (w * h) // 97)
        ys, xs, cs = rng.integers(0, h, n), rng.integers(0, w, n), rng.integers(0, 3, n)
        vals = np.array([0x7C00, 0xFC00, 0x7E00, 0xFE01, 0xBC00, 0x8001, 0x0001, 0x03FF, 0x7BFF, 0x5640, 0x0000, 0x8000], dtype=np.uint16)
        bits[ys, xs, cs] = vals[rng.integers(0, vals.size, n)]
    img = Image(A.UHDR_IMG_FMT_64bppRGBAHalfFloat, w, h, cg, A.UHDR_CT_LINEAR, A.UHDR_CR_FULL_RANGE, align)
    img.valid(0)[:] = px.view(np.uint64).reshape(h, w)
    return img
