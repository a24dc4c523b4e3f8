"""TEST INFRASTRUCTURE ONLY: ctypes access to the two CPU checkers.

* ``port()``  -> oracle/libuhdr_oracle.so, the plain-C restatement (always buildable: `make -C oracle port`)
* ``ref()``   -> oracle/_ref/libuhdr_ref.so, the REAL reference compiled from /root/reference by
                 oracle/Makefile (`make -C oracle ref`; only where /root/reference exists -- the
                 prebuilt .so travels to the GPU box with the gpurun snapshot).  ``None`` if absent.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (libultrahdr_amd) never does.  The ctypes structs are shared with the product
binding because all three libraries use the reference's public struct layouts.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from libultrahdr_amd import capi as A
from libultrahdr_amd.images import Image

HERE = os.path.dirname(os.path.abspath(__file__))
PORT_PATH = os.path.join(HERE, "libuhdr_oracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libuhdr_ref.so")
_JPEG_CANDIDATES = ["/opt/conda/lib/libjpeg.so.9"]

_port = None
_ref = None
_ref_tried = False
_P = C.POINTER


def build_port():
    subprocess.check_call(["make", "-s", "-C", HERE, "port"])


class ScanDesc(C.Structure):  # uo_scan_t
    _fields_ = [("ncomp", C.c_int), ("coef", C.c_void_p * 3), ("bw", C.c_int * 3), ("bh", C.c_int * 3), ("hs", C.c_int * 3),
                ("vs", C.c_int * 3), ("w", C.c_uint), ("h", C.c_uint), ("restart_interval", C.c_int)]


def port() -> C.CDLL:
    global _port
    if _port is not None:
        return _port
    if not os.path.exists(PORT_PATH):
        build_port()
    lib = C.CDLL(PORT_PATH)
    lib.uo_apply_gainmap.restype = C.c_int
    lib.uo_apply_gainmap.argtypes = [_P(A.RawImage), _P(A.RawImage), _P(A.GainmapMetadata), C.c_int, C.c_int, C.c_float, _P(A.RawImage)]
    lib.uo_generate_gainmap.restype = C.c_int
    lib.uo_generate_gainmap.argtypes = [_P(A.RawImage), _P(A.RawImage), _P(A.EncodeCfg), _P(A.GainmapMetadata), _P(A.RawImage)]
    lib.uo_generate_gainmap_pass1.restype = C.c_int
    lib.uo_generate_gainmap_pass1.argtypes = [_P(A.RawImage), _P(A.RawImage), _P(A.EncodeCfg), C.c_void_p, _P(C.c_float), _P(C.c_int)]
    lib.uo_generate_gainmap_pass2.restype = None
    lib.uo_generate_gainmap_pass2.argtypes = [C.c_void_p, _P(C.c_float), C.c_float, C.c_int, C.c_uint, C.c_uint, C.c_void_p, C.c_size_t]
    lib.uo_tone_map.restype = C.c_int
    lib.uo_tone_map.argtypes = [_P(A.RawImage), _P(A.RawImage)]
    lib.uo_convert_yuv.restype = C.c_int
    lib.uo_convert_yuv.argtypes = [_P(A.RawImage), C.c_int, C.c_int]
    lib.uo_convert_raw_input_to_ycbcr.restype = C.c_int
    lib.uo_convert_raw_input_to_ycbcr.argtypes = [_P(A.RawImage), C.c_int, _P(A.RawImage)]
    lib.uo_copy_raw_image.restype = C.c_int
    lib.uo_copy_raw_image.argtypes = [_P(A.RawImage), _P(A.RawImage)]
    lib.uo_jpeg_quant_table.restype = None
    lib.uo_jpeg_quant_table.argtypes = [C.c_int, C.c_int, _P(C.c_uint16)]
    lib.uo_fdct_quant_plane.restype = None
    lib.uo_fdct_quant_plane.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, _P(C.c_uint16), C.c_void_p]
    lib.uo_jpeg_rgb_to_ycc.restype = None
    lib.uo_jpeg_rgb_to_ycc.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.uo_huffman_encode_scan.restype = C.c_size_t
    lib.uo_huffman_encode_scan.argtypes = [_P(ScanDesc), C.c_void_p, C.c_size_t]
    lib.uo_jpeg_assemble.restype = C.c_size_t
    lib.uo_jpeg_assemble.argtypes = [_P(ScanDesc), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.uo_huffman_decode_scan.restype = C.c_int
    lib.uo_huffman_decode_scan.argtypes = [_P(ScanDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, _P(C.c_void_p)]
    lib.uo_std_huff_table.restype = None
    lib.uo_std_huff_table.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, _P(C.c_int)]
    lib.uo_idct_dequant_plane.restype = None
    lib.uo_idct_dequant_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, _P(C.c_uint16), C.c_void_p, C.c_size_t]
    lib.uo_jpeg_ycc_to_rgb.restype = None
    lib.uo_jpeg_ycc_to_rgb.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_int]
    _common_scalar_sigs(lib, "uo_")
    lib.uo_oetf_code.restype = None
    lib.uo_oetf_code.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.uo_lut.restype = None
    lib.uo_lut.argtypes = [C.c_int, C.c_void_p]
    _port = lib
    return lib


def _common_scalar_sigs(lib, pfx):
    getattr(lib, pfx + "eval").restype = C.c_int
    getattr(lib, pfx + "eval").argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
    getattr(lib, pfx + "float_to_half").restype = None
    getattr(lib, pfx + "float_to_half").argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    getattr(lib, pfx + "color_to_rgba1010102").restype = C.c_uint32
    getattr(lib, pfx + "color_to_rgba1010102").argtypes = [C.c_float] * 3
    getattr(lib, pfx + "color_to_rgbaf16").restype = C.c_uint64
    getattr(lib, pfx + "color_to_rgbaf16").argtypes = [C.c_float] * 3
    getattr(lib, pfx + "compute_gain").restype = C.c_float
    getattr(lib, pfx + "compute_gain").argtypes = [C.c_float] * 2
    getattr(lib, pfx + "affine_map_gain").restype = C.c_uint8
    getattr(lib, pfx + "affine_map_gain").argtypes = [C.c_float] * 4
    getattr(lib, pfx + "encode_gain").restype = C.c_uint8
    getattr(lib, pfx + "encode_gain").argtypes = [C.c_float] * 5
    getattr(lib, pfx + "apply_gain").restype = None
    getattr(lib, pfx + "apply_gain").argtypes = [_P(C.c_float), C.c_float, _P(A.GainmapMetadata), C.c_float, C.c_int, _P(C.c_float)]
    getattr(lib, pfx + "idw_weights").restype = None
    getattr(lib, pfx + "idw_weights").argtypes = [C.c_int, C.c_int, C.c_void_p]
    getattr(lib, pfx + "color_fn").restype = None
    getattr(lib, pfx + "color_fn").argtypes = [C.c_int, _P(C.c_float), _P(C.c_float)]


def ref():
    """The real reference, or None when oracle/_ref was not built / cannot load on this box."""
    global _ref, _ref_tried
    if _ref_tried:
        return _ref
    _ref_tried = True
    if not os.path.exists(REF_PATH):
        return None
    try:
        for j in _JPEG_CANDIDATES:  # no RPATH in the .so: bring libjpeg in by absolute path first
            if os.path.exists(j):
                C.CDLL(j, mode=C.RTLD_GLOBAL)
                break
        lib = C.CDLL(REF_PATH)
    except OSError:
        return None
    lib.ref_apply_gainmap.restype = C.c_int
    lib.ref_apply_gainmap.argtypes = [_P(A.RawImage), _P(A.RawImage), _P(A.GainmapMetadata), C.c_int, C.c_int, C.c_float, _P(A.RawImage), C.c_char_p]
    lib.ref_generate_gainmap.restype = C.c_int
    lib.ref_generate_gainmap.argtypes = [_P(A.RawImage), _P(A.RawImage), _P(A.EncodeCfg), _P(A.GainmapMetadata), _P(A.RawImage), C.c_char_p]
    lib.ref_tone_map.restype = C.c_int
    lib.ref_tone_map.argtypes = [_P(A.RawImage), _P(A.RawImage), C.c_char_p]
    lib.ref_convert_yuv.restype = C.c_int
    lib.ref_convert_yuv.argtypes = [_P(A.RawImage), C.c_int, C.c_int, C.c_char_p]
    lib.ref_copy_raw_image.restype = C.c_int
    lib.ref_copy_raw_image.argtypes = [_P(A.RawImage), _P(A.RawImage)]
    lib.ref_convert_raw_input_to_ycbcr.restype = C.c_int
    lib.ref_convert_raw_input_to_ycbcr.argtypes = [_P(A.RawImage), C.c_int, _P(A.RawImage)]
    lib.ref_jpeg_compress.restype = C.c_long
    lib.ref_jpeg_compress.argtypes = [_P(A.RawImage), C.c_int, C.c_void_p, C.c_size_t]
    lib.ref_jpeg_read_coefficients.restype = C.c_int
    lib.ref_jpeg_read_coefficients.argtypes = [C.c_void_p, C.c_size_t, _P(C.c_void_p), C.c_void_p, _P(C.c_int), _P(C.c_int), _P(C.c_int)]
    lib.ref_jpeg_decompress.restype = C.c_int
    lib.ref_jpeg_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _P(A.RawImage), C.c_void_p, C.c_size_t]
    lib.ref_info.restype = C.c_char_p
    lib.ref_info.argtypes = []
    lib.ref_uhdr_encode.restype = C.c_long
    lib.ref_uhdr_encode.argtypes = [_P(A.RawImage), _P(A.RawImage), C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.ref_uhdr_decode.restype = C.c_int
    lib.ref_uhdr_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, _P(C.c_int), _P(C.c_int)]
    _common_scalar_sigs(lib, "ref_")
    _ref = lib
    return lib


# ---- convenience wrappers shared by tests / smoke / bench ---------------------------------------
FN = dict(srgb_inv=0, srgb_inv_lut=1, srgb_oetf=2, hlg_oetf=3, hlg_oetf_lut=4, hlg_inv=5, hlg_inv_lut=6,
          pq_oetf=7, pq_oetf_lut=8, pq_inv=9, pq_inv_lut=10, half_to_float=11, hlg_ootf=12, hlg_inv_ootf=13,
          log2_f64=14)  # 14: port only ((float)log2((double)x), the encodeGain / computeGain call)


def eval_fn(lib, pfx, name, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    rc = getattr(lib, pfx + "eval")(FN[name], x.ctypes.data, out.ctypes.data, x.size)
    assert rc == 0
    return out


def apply_gainmap(lib_kind, sdr: Image, gm: Image, md, out_ct, max_display_boost=A.FLT_MAX, dest_align=1) -> Image:
    """Run applyGainMap on 'port' or 'ref'; returns the destination image."""
    fmt = A.UHDR_IMG_FMT_64bppRGBAHalfFloat if out_ct == A.UHDR_CT_LINEAR else A.UHDR_IMG_FMT_32bppRGBA1010102
    dest = Image(fmt, sdr.w, sdr.h, align=dest_align)
    if lib_kind == "port":
        rc = port().uo_apply_gainmap(C.byref(sdr.raw), C.byref(gm.raw), C.byref(md), out_ct, fmt, max_display_boost, C.byref(dest.raw))
        detail = b""
    else:
        buf = C.create_string_buffer(256)
        rc = ref().ref_apply_gainmap(C.byref(sdr.raw), C.byref(gm.raw), C.byref(md), out_ct, fmt, max_display_boost, C.byref(dest.raw), buf)
        detail = buf.value
    if rc != 0:
        raise A.UhdrError(rc, detail.decode("utf-8", "replace"))
    return dest


def generate_gainmap(lib_kind, sdr: Image, hdr: Image, cfg: A.EncodeCfg):
    s = cfg.map_dimension_scale_factor
    mw, mh = sdr.w // s, sdr.h // s
    if mw == 0 or mh == 0:
        s2 = min(sdr.w, sdr.h)
        s2 = s2 // 8 if s2 >= 8 else 1
        mw, mh = sdr.w // s2, sdr.h // s2
    fmt = A.UHDR_IMG_FMT_24bppRGB888 if cfg.use_multi_channel_gainmap else A.UHDR_IMG_FMT_8bppYCbCr400
    gm = Image(fmt, mw, mh, align=64)
    md = A.GainmapMetadata()
    if lib_kind == "port":
        rc = port().uo_generate_gainmap(C.byref(sdr.raw), C.byref(hdr.raw), C.byref(cfg), C.byref(md), C.byref(gm.raw))
        detail = b""
    else:
        buf = C.create_string_buffer(256)
        rc = ref().ref_generate_gainmap(C.byref(sdr.raw), C.byref(hdr.raw), C.byref(cfg), C.byref(md), C.byref(gm.raw), buf)
        detail = buf.value
    if rc != 0:
        raise A.UhdrError(rc, detail.decode("utf-8", "replace"))
    gm.sync_meta_from_raw()
    return md, gm


def tone_map(lib_kind, hdr: Image, sdr_fmt=None) -> Image:
    if sdr_fmt is None:
        sdr_fmt = {A.UHDR_IMG_FMT_24bppYCbCrP010: A.UHDR_IMG_FMT_12bppYCbCr420,
                   A.UHDR_IMG_FMT_30bppYCbCr444: A.UHDR_IMG_FMT_24bppYCbCr444}.get(hdr.fmt, A.UHDR_IMG_FMT_32bppRGBA8888)
    sdr = Image(sdr_fmt, hdr.w, hdr.h, align=64)
    if lib_kind == "port":
        rc = port().uo_tone_map(C.byref(hdr.raw), C.byref(sdr.raw))
    else:
        rc = ref().ref_tone_map(C.byref(hdr.raw), C.byref(sdr.raw), None)
    if rc != 0:
        raise A.UhdrError(rc, "tone_map")
    return sdr


def convert_yuv(lib_kind, img: Image, src, dst) -> Image:
    out = img.clone()
    if lib_kind == "port":
        rc = port().uo_convert_yuv(C.byref(out.raw), src, dst)
    else:
        rc = ref().ref_convert_yuv(C.byref(out.raw), src, dst, None)
    if rc != 0:
        raise A.UhdrError(rc, "convert_yuv")
    return out


def convert_raw_input_to_ycbcr(lib_kind, src: Image, chroma: bool) -> Image:
    ten = src.fmt == A.UHDR_IMG_FMT_32bppRGBA1010102
    if ten:
        fmt = A.UHDR_IMG_FMT_24bppYCbCrP010 if chroma else A.UHDR_IMG_FMT_30bppYCbCr444
    else:
        fmt = A.UHDR_IMG_FMT_12bppYCbCr420 if chroma else A.UHDR_IMG_FMT_24bppYCbCr444
    dst = Image(fmt, src.w, src.h, align=64)
    if lib_kind == "port":
        rc = port().uo_convert_raw_input_to_ycbcr(C.byref(src.raw), int(chroma), C.byref(dst.raw))
    else:
        rc = ref().ref_convert_raw_input_to_ycbcr(C.byref(src.raw), int(chroma), C.byref(dst.raw))
    if rc != 0:
        raise A.UhdrError(rc, "convert_raw_input_to_ycbcr")
    return dst


def copy_raw_image(lib_kind, src: Image, dst: Image) -> int:
    """copy_raw_image(src, dst) on caller-allocated descriptors; returns the uhdr error code."""
    if lib_kind == "port":
        return port().uo_copy_raw_image(C.byref(src.raw), C.byref(dst.raw))
    return ref().ref_copy_raw_image(C.byref(src.raw), C.byref(dst.raw))


def fdct_quant_port(plane: np.ndarray, stride: int, bw: int, bh: int, qt: np.ndarray) -> np.ndarray:
    out = np.zeros((bh, bw, 64), dtype=np.int16)
    q = (C.c_uint16 * 64)(*[int(v) for v in qt])
    port().uo_fdct_quant_plane(plane.ctypes.data, stride, bw, bh, q, out.ctypes.data)
    return out


def idct_dequant_port(coef: np.ndarray, qt: np.ndarray) -> np.ndarray:
    """coef: (bh, bw, 64) int16 in JBLOCK layout -> (bh*8, bw*8) uint8 plane."""
    coef = np.ascontiguousarray(coef, dtype=np.int16)
    bh, bw = coef.shape[:2]
    out = np.zeros((bh * 8, bw * 8), dtype=np.uint8)
    q = (C.c_uint16 * 64)(*[int(v) for v in qt])
    port().uo_idct_dequant_plane(coef.ctypes.data, bw, bh, q, out.ctypes.data, bw * 8)
    return out


def scan_desc(coefs, w: int, h: int, sampling, restart_interval: int) -> ScanDesc:
    """coefs: list of (bh, bw, 64) int16 arrays (kept alive by the caller); sampling: [(h, v)] per component."""
    sd = ScanDesc()
    sd.ncomp = len(coefs)
    for c, a in enumerate(coefs):
        assert a.dtype == np.int16 and a.flags["C_CONTIGUOUS"]
        sd.coef[c] = a.ctypes.data
        sd.bh[c], sd.bw[c] = a.shape[0], a.shape[1]
        sd.hs[c], sd.vs[c] = sampling[c]
    sd.w, sd.h, sd.restart_interval = w, h, restart_interval
    return sd


def huffman_encode_port(coefs, w: int, h: int, sampling, restart_interval: int = 0) -> bytes:
    """Entropy-coded data of the scan (between the SOS header and EOI), Annex K tables."""
    coefs = [np.ascontiguousarray(c, dtype=np.int16) for c in coefs]
    sd = scan_desc(coefs, w, h, sampling, restart_interval)
    cap = sum(c.size for c in coefs) * 8 + 4096  # worst case: 1660 bits per block, every byte stuffed
    out = np.zeros(cap, dtype=np.uint8)
    n = port().uo_huffman_encode_scan(C.byref(sd), out.ctypes.data, cap)
    assert n > 0
    return out[:n].tobytes()


def std_dht_tables():
    """(bits[4][17], vals[4][256]) uint8 arrays: DC luma, AC luma, DC chroma, AC chroma (Annex K)."""
    bits, vals = np.zeros((4, 17), dtype=np.uint8), np.zeros((4, 256), dtype=np.uint8)
    for t, (ac, chroma) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        nv = C.c_int(0)
        port().uo_std_huff_table(ac, chroma, bits[t].ctypes.data, vals[t].ctypes.data, C.byref(nv))
    return bits, vals


def huffman_decode_port(shapes, w: int, h: int, sampling, restart_interval: int, data: bytes, tables=None):
    """Entropy-coded data -> list of (bh, bw, 64) int16 arrays; shapes = [(bh, bw)] per component."""
    out = [np.zeros((bh, bw, 64), dtype=np.int16) for (bh, bw) in shapes]
    sd = scan_desc(out, w, h, sampling, restart_interval)
    bits, vals = tables if tables is not None else std_dht_tables()
    bits, vals = np.ascontiguousarray(bits, dtype=np.uint8), np.ascontiguousarray(vals, dtype=np.uint8)
    src = np.frombuffer(data, dtype=np.uint8)
    ptrs = (C.c_void_p * 3)(*[out[c].ctypes.data if c < len(out) else None for c in range(3)])
    rc = port().uo_huffman_decode_scan(C.byref(sd), bits.ctypes.data, vals.ctypes.data, src.ctypes.data, src.size, ptrs)
    return rc, out


def jpeg_assemble_port(coefs, w: int, h: int, sampling, restart_interval: int, qt_luma, qt_chroma, scan: bytes) -> bytes:
    """A complete baseline JFIF file around entropy-coded data."""
    coefs = [np.ascontiguousarray(c, dtype=np.int16) for c in coefs]
    sd = scan_desc(coefs, w, h, sampling, restart_interval)
    qt = np.stack([np.asarray(qt_luma, dtype=np.uint16), np.asarray(qt_chroma, dtype=np.uint16)])
    qt = np.ascontiguousarray(qt)
    sc = np.frombuffer(scan, dtype=np.uint8)
    out = np.zeros(len(scan) + 2048, dtype=np.uint8)
    n = port().uo_jpeg_assemble(C.byref(sd), qt.ctypes.data, sc.ctypes.data, sc.size, out.ctypes.data, out.size)
    assert n > 0
    return out[:n].tobytes()


def jpeg_rgb_to_ycc_port(rgb: np.ndarray, stride_px: int, w: int, h: int):
    """rgb: packed RGB888 rows (uint8) -> three (h, w) uint8 planes (libjpeg rgb_ycc_convert)."""
    planes = [np.zeros((h, w), dtype=np.uint8) for _ in range(3)]
    rgb = np.ascontiguousarray(rgb)
    port().uo_jpeg_rgb_to_ycc(rgb.ctypes.data, stride_px, w, h, planes[0].ctypes.data, planes[1].ctypes.data, planes[2].ctypes.data, w)
    return planes


def jpeg_ycc_to_rgb_port(y: np.ndarray, cb: np.ndarray, cr: np.ndarray, out_bpp: int = 3, variant: int = 0) -> np.ndarray:
    """three (h, w) uint8 planes -> (h, w*out_bpp) packed RGB / RGBA (libjpeg ycc_rgb_convert)."""
    y, cb, cr = [np.ascontiguousarray(p) for p in (y, cb, cr)]
    h, w = y.shape
    out = np.zeros((h, w * out_bpp), dtype=np.uint8)
    port().uo_jpeg_ycc_to_rgb(y.ctypes.data, cb.ctypes.data, cr.ctypes.data, w, w, h, out.ctypes.data, w, out_bpp, variant)
    return out


def quant_table_port(quality: int, chroma: bool) -> np.ndarray:
    q = (C.c_uint16 * 64)()
    port().uo_jpeg_quant_table(quality, int(chroma), q)
    return np.frombuffer(q, dtype=np.uint16).copy()


# ---- the reference's whole public API (CPU baseline of uhdr_encode / uhdr_decode) and its JPEG helpers ---------------
def ref_uhdr_encode(hdr: Image, sdr, quality=95, preset=A.UHDR_USAGE_BEST_QUALITY) -> bytes:
    """uhdr_encode (ultrahdr_api.cpp:1200) through the real reference: API-1 (hdr + sdr) or API-0 (sdr None)."""
    lib = ref()
    cap = max(hdr.raw.w * hdr.raw.h * 6, 1 << 16)
    buf = (C.c_uint8 * cap)()
    n = lib.ref_uhdr_encode(C.byref(hdr.raw), C.byref(sdr.raw) if sdr is not None else None, quality, preset, buf, cap)
    if n <= 0:
        raise RuntimeError(f"ref_uhdr_encode failed: {n}")
    return bytes(buf[:n])


def ref_uhdr_decode(jpeg: bytes, out_ct, out_fmt, dest: np.ndarray = None):
    """uhdr_decode (ultrahdr_api.cpp:1918) through the real reference -> (w, h) [+ packed pixels into dest]."""
    lib = ref()
    w, h = C.c_int(0), C.c_int(0)
    src = (C.c_uint8 * len(jpeg)).from_buffer_copy(jpeg)
    rc = lib.ref_uhdr_decode(src, len(jpeg), out_ct, out_fmt, dest.ctypes.data if dest is not None else None,
                             dest.nbytes if dest is not None else 0, C.byref(w), C.byref(h))
    if rc != 0:
        raise RuntimeError(f"ref_uhdr_decode failed: {rc}")
    return w.value, h.value


def ref_jpeg_compress(img: Image, quality: int) -> bytes:
    """JpegEncoderHelper::compressImage (jpegencoderhelper.cpp:101) through the real reference."""
    lib = ref()
    cap = max(img.raw.w * img.raw.h * 4, 1 << 16)
    buf = (C.c_uint8 * cap)()
    n = lib.ref_jpeg_compress(C.byref(img.raw), quality, buf, cap)
    if n <= 0:
        raise RuntimeError(f"ref_jpeg_compress failed: {n}")
    return bytes(buf[:n])


def ref_jpeg_decompress(jpeg: bytes, mode: int):
    """JpegDecoderHelper::decompressImage (jpegdecoderhelper.cpp:169): mode 0 planar YCbCr, 1 the stream's own space."""
    lib = ref()
    src = (C.c_uint8 * len(jpeg)).from_buffer_copy(jpeg)
    cap = 1 << 28
    buf = np.empty(cap, dtype=np.uint8)
    desc = A.RawImage()
    rc = lib.ref_jpeg_decompress(src, len(jpeg), mode, C.byref(desc), buf.ctypes.data, cap)
    if rc != 0:
        raise RuntimeError(f"ref_jpeg_decompress failed: {rc}")
    return desc, buf
