"""Raw-image containers: the Python-side twin of ``uhdr_raw_image_ext_t``
(``/root/reference/lib/src/ultrahdr_api.cpp:55-117``): one allocation, planes back to back,
stride aligned to ``align`` pixels.  Backed by numpy (host) or a torch CUDA tensor (device)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi as A

_BPS = {  # bytes per sample
    A.UHDR_IMG_FMT_24bppYCbCrP010: 2,
    A.UHDR_IMG_FMT_30bppYCbCr444: 2,
    A.UHDR_IMG_FMT_24bppRGB888: 3,
    A.UHDR_IMG_FMT_32bppRGBA8888: 4,
    A.UHDR_IMG_FMT_32bppRGBA1010102: 4,
    A.UHDR_IMG_FMT_64bppRGBAHalfFloat: 8,
}


def bytes_per_sample(fmt: int) -> int:
    return _BPS.get(fmt, 1)


def _align(x: int, m: int) -> int:
    return ((x + m - 1) // m) * m


def plane_layout(fmt: int, w: int, h: int, align: int = 64):
    """[(rows, stride_px, valid_width_px)] per plane, following uhdr_raw_image_ext's rules."""
    aw = _align(w, align)
    if fmt == A.UHDR_IMG_FMT_24bppYCbCrP010:
        return [(h, aw, w), ((h + 1) // 2, aw, ((w + 1) // 2) * 2), None]
    if fmt == A.UHDR_IMG_FMT_12bppYCbCr420:
        # the reference uses aligned_width / 2; round up so odd unaligned widths keep their last sample
        return [(h, aw, w), ((h + 1) // 2, (aw + 1) // 2, (w + 1) // 2), ((h + 1) // 2, (aw + 1) // 2, (w + 1) // 2)]
    if fmt == A.UHDR_IMG_FMT_16bppYCbCr422:
        return [(h, aw, w), (h, (aw + 1) // 2, (w + 1) // 2), (h, (aw + 1) // 2, (w + 1) // 2)]
    if fmt in (A.UHDR_IMG_FMT_24bppYCbCr444, A.UHDR_IMG_FMT_30bppYCbCr444):
        return [(h, aw, w)] * 3
    return [(h, aw, w), None, None]


class Image:
    """A raw image whose memory we own.  ``.raw`` is the ctypes ``uhdr_raw_image_t`` to pass to
    the C ABI; ``.plane(i)`` gives a 2-D numpy view (host images) of ``rows x stride`` samples."""

    def __init__(self, fmt, w, h, cg=A.UHDR_CG_UNSPECIFIED, ct=A.UHDR_CT_UNSPECIFIED,
                 rng=A.UHDR_CR_UNSPECIFIED, align=64, device=None, fill=None):
        self.fmt, self.w, self.h, self.align = fmt, w, h, align
        self.layout = plane_layout(fmt, w, h, align)
        bps = bytes_per_sample(fmt)
        self.offsets, total = [], 0
        for pl in self.layout:
            self.offsets.append(total)
            if pl is not None:
                total += _align(pl[0] * pl[1] * bps, 256)
        self.nbytes = max(total, 256)
        self.device = device
        if device is None:
            self.buf = np.zeros(self.nbytes, dtype=np.uint8)
            if fill is not None:
                self.buf[:] = fill
            base = self.buf.ctypes.data
        else:
            import torch

            self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
            if fill is not None:
                self.buf.fill_(fill)
            # torch's fill runs asynchronously on torch's stream; the library works on its own
            # (non-blocking) stream, so make the memory quiescent before anybody else touches it
            torch.cuda.current_stream(self.buf.device).synchronize()
            base = self.buf.data_ptr()
        self.raw = A.RawImage()
        self.raw.fmt, self.raw.cg, self.raw.ct, self.raw.range = fmt, cg, ct, rng
        self.raw.w, self.raw.h = w, h
        for i, pl in enumerate(self.layout):
            self.raw.planes[i] = (base + self.offsets[i]) if pl is not None else None
            self.raw.stride[i] = pl[1] if pl is not None else 0

    # ---- host-side views -----------------------------------------------------------------
    def _dtype(self):
        bps = bytes_per_sample(self.fmt)
        return {1: np.uint8, 2: np.uint16, 3: np.uint8, 4: np.uint32, 8: np.uint64}[bps]

    def plane(self, i) -> np.ndarray:
        assert self.device is None, "plane() views are for host images; use .to_host() first"
        rows, stride, _ = self.layout[i]
        bps = bytes_per_sample(self.fmt)
        raw = self.buf[self.offsets[i]: self.offsets[i] + rows * stride * bps]
        if bps == 3:
            return raw.view(np.uint8).reshape(rows, stride * 3)
        return raw.view(self._dtype()).reshape(rows, stride)

    def valid(self, i) -> np.ndarray:
        """Plane i cropped to the samples the image actually contains (no stride padding)."""
        rows, _, wv = self.layout[i]
        k = 3 if bytes_per_sample(self.fmt) == 3 else 1
        return self.plane(i)[:rows, : wv * k]

    def planes_valid(self):
        return [self.valid(i) for i, pl in enumerate(self.layout) if pl is not None]

    def plane_tensor(self, i):
        """Device images: plane i as a (rows, stride * bytes_per_sample) uint8 torch view (no copy)."""
        assert self.device is not None
        rows, stride, _ = self.layout[i]
        bps = bytes_per_sample(self.fmt)
        return self.buf[self.offsets[i]: self.offsets[i] + rows * stride * bps].view(rows, stride * bps)

    # ---- movement ----------------------------------------------------------------------------
    def to(self, device):
        out = Image(self.fmt, self.w, self.h, self.raw.cg, self.raw.ct, self.raw.range, self.align, device)
        import torch

        if self.device is None:
            out.buf.copy_(torch.from_numpy(self.buf))
        else:
            out.buf.copy_(self.buf)
        torch.cuda.current_stream(out.buf.device).synchronize()
        return out

    def to_host(self):
        if self.device is None:
            return self
        out = Image(self.fmt, self.w, self.h, self.raw.cg, self.raw.ct, self.raw.range, self.align, None)
        out.buf[:] = self.buf.cpu().numpy()
        return out

    def sync_meta_from_raw(self):
        """After a call that rewrote fmt/w/h/colour aspects in ``raw`` (generateGainMap, toneMap)."""
        if (self.raw.fmt, self.raw.w, self.raw.h) != (self.fmt, self.w, self.h):
            self.fmt, self.w, self.h = self.raw.fmt, self.raw.w, self.raw.h

    def clone(self):
        out = Image(self.fmt, self.w, self.h, self.raw.cg, self.raw.ct, self.raw.range, self.align, self.device)
        if self.device is None:
            out.buf[:] = self.buf
        else:
            import torch

            out.buf.copy_(self.buf)
            torch.cuda.current_stream(out.buf.device).synchronize()
        return out


def stripe_view(img: Image, row0: int, rows: int) -> A.RawImage:
    """A ``uhdr_raw_image_t`` describing rows [row0, row0+rows) of ``img`` (no copy).  row0 and
    rows must respect the chroma subsampling of the format (even for 4:2:0 / P010)."""
    r = A.RawImage()
    C.memmove(C.byref(r), C.byref(img.raw), C.sizeof(A.RawImage))
    r.h = rows
    bps = bytes_per_sample(img.fmt)
    sub = img.fmt in (A.UHDR_IMG_FMT_24bppYCbCrP010, A.UHDR_IMG_FMT_12bppYCbCr420)
    if sub:
        assert row0 % 2 == 0 and rows % 2 == 0
    for i, pl in enumerate(img.layout):
        if pl is None:
            continue
        prow = row0 // 2 if (sub and i > 0) else row0
        r.planes[i] = img.raw.planes[i] + prow * pl[1] * bps
    return r
