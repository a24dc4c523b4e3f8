"""ctypes binding of the drop-in libuhdr.so (facade/): the reference's own 43-function C API
(ultrahdr_api.h:296-905) with uhdr_enable_gpu_acceleration() routed to libuhdr_hip.so.  Only what the API-level
measurements and tests need is wrapped: encode (API-0 / API-1), decode, the acceleration switch."""
import ctypes as C
import os

import numpy as np

from . import capi as A

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "lib", "libuhdr.so")


class CompressedImage(C.Structure):  # uhdr_compressed_image_t, ultrahdr_api.h:248-256
    _fields_ = [("data", C.c_void_p), ("data_sz", C.c_size_t), ("capacity", C.c_size_t),
                ("cg", C.c_int), ("ct", C.c_int), ("range", C.c_int)]


_lib = None


def available():
    return os.path.isfile(PATH)


def load():
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (libamdhip64 first, see capi.load)

    A.load()  # libuhdr_hip.so, RTLD_GLOBAL order
    jpeg = os.path.join(_HERE, "lib", "libjpeg.so.9")
    if os.path.exists(jpeg):
        C.CDLL(jpeg, mode=C.RTLD_GLOBAL)
    lib = C.CDLL(PATH)
    E, P = A.ErrorInfo, C.c_void_p
    lib.uhdr_create_encoder.restype = P
    lib.uhdr_create_decoder.restype = P
    lib.uhdr_release_encoder.argtypes = [P]
    lib.uhdr_release_decoder.argtypes = [P]
    for name, args in (("uhdr_enc_set_raw_image", [P, C.POINTER(A.RawImage), C.c_int]),
                       ("uhdr_enc_set_quality", [P, C.c_int, C.c_int]),
                       ("uhdr_enc_set_preset", [P, C.c_int]),
                       ("uhdr_enc_set_using_multi_channel_gainmap", [P, C.c_int]),
                       ("uhdr_enc_set_gainmap_scale_factor", [P, C.c_int]),
                       ("uhdr_enable_gpu_acceleration", [P, C.c_int]),
                       ("uhdr_add_effect_mirror", [P, C.c_int]),
                       ("uhdr_add_effect_rotate", [P, C.c_int]),
                       ("uhdr_add_effect_crop", [P, C.c_int, C.c_int, C.c_int, C.c_int]),
                       ("uhdr_add_effect_resize", [P, C.c_int, C.c_int]),
                       ("uhdr_encode", [P]),
                       ("uhdr_dec_set_image", [P, C.POINTER(CompressedImage)]),
                       ("uhdr_dec_set_out_color_transfer", [P, C.c_int]),
                       ("uhdr_dec_set_out_img_format", [P, C.c_int]),
                       ("uhdr_decode", [P])):
        f = getattr(lib, name)
        f.restype, f.argtypes = E, args
    lib.uhdr_get_encoded_stream.restype = C.POINTER(CompressedImage)
    lib.uhdr_get_encoded_stream.argtypes = [P]
    lib.uhdr_get_decoded_image.restype = C.POINTER(A.RawImage)
    lib.uhdr_get_decoded_image.argtypes = [P]
    lib.uhdr_get_decoded_gainmap_image.restype = C.POINTER(A.RawImage)
    lib.uhdr_get_decoded_gainmap_image.argtypes = [P]
    lib.uhdr_reset_decoder.argtypes = [P]
    lib.uhdr_reset_decoder.restype = None
    _lib = lib
    return lib


UHDR_HDR_IMG, UHDR_SDR_IMG, UHDR_BASE_IMG, UHDR_GAIN_MAP_IMG = 0, 1, 2, 3  # uhdr_img_label_t


def _chk(st):
    if st.error_code != 0:
        raise A.UhdrError(st.error_code, st.detail.decode("utf-8", "replace") if st.has_detail else "")


def _add_effects(lib, h, effects):
    """effects: [("rotate", degrees) | ("mirror", direction) | ("crop", left, right, top, bottom) | ("resize", w, h)]"""
    for e in effects or ():
        if e[0] == "rotate":
            _chk(lib.uhdr_add_effect_rotate(h, e[1]))
        elif e[0] == "mirror":
            _chk(lib.uhdr_add_effect_mirror(h, e[1]))
        elif e[0] == "crop":
            _chk(lib.uhdr_add_effect_crop(h, *e[1:5]))
        elif e[0] == "resize":
            _chk(lib.uhdr_add_effect_resize(h, e[1], e[2]))
        else:
            raise ValueError(e)


last_call_seconds = 0.0  # wall time of the last uhdr_encode / uhdr_decode call itself (what bench.py's api_level reports)
last_gainmap_seconds = 0.0  # wall time of the first uhdr_get_decoded_gainmap_image call after it (decode(..., want_gainmap=True))


def encode(hdr, sdr=None, gpu=False, quality=95, preset=A.UHDR_USAGE_BEST_QUALITY, effects=None, multi_channel=None, scale=None) -> bytes:
    """uhdr_encode: API-1 (hdr + sdr raw intents) or API-0 (hdr only); host images (libultrahdr_amd.images.Image).
    multi_channel / scale: uhdr_enc_set_using_multi_channel_gainmap / _gainmap_scale_factor (None: the API's defaults)."""
    global last_call_seconds
    import time

    lib = load()
    h = lib.uhdr_create_encoder()
    try:
        _add_effects(lib, h, effects)
        _chk(lib.uhdr_enc_set_raw_image(h, C.byref(hdr.raw), UHDR_HDR_IMG))
        if sdr is not None:
            _chk(lib.uhdr_enc_set_raw_image(h, C.byref(sdr.raw), UHDR_SDR_IMG))
        _chk(lib.uhdr_enc_set_quality(h, quality, UHDR_BASE_IMG))
        _chk(lib.uhdr_enc_set_preset(h, preset))
        if multi_channel is not None:
            _chk(lib.uhdr_enc_set_using_multi_channel_gainmap(h, int(multi_channel)))
        if scale is not None:
            _chk(lib.uhdr_enc_set_gainmap_scale_factor(h, int(scale)))
        if gpu:
            _chk(lib.uhdr_enable_gpu_acceleration(h, 1))
        t0 = time.perf_counter()
        st = lib.uhdr_encode(h)
        last_call_seconds = time.perf_counter() - t0
        _chk(st)
        o = lib.uhdr_get_encoded_stream(h).contents
        return C.string_at(o.data, o.data_sz)
    finally:
        lib.uhdr_release_encoder(h)


def _packed(o):
    bpp = {A.UHDR_IMG_FMT_64bppRGBAHalfFloat: 8, A.UHDR_IMG_FMT_8bppYCbCr400: 1, A.UHDR_IMG_FMT_24bppRGB888: 3}.get(o.fmt, 4)
    a = np.ctypeslib.as_array(C.cast(o.planes[0], C.POINTER(C.c_uint8)), shape=(o.h, o.stride[0] * bpp))
    return a[:, : o.w * bpp].reshape(o.h, o.w, bpp).copy()


def decode(jpeg: bytes, out_ct, out_fmt, gpu=False, effects=None, want_gainmap=False, decodes=1):
    """uhdr_decode -> packed pixels as a (h, w, bytes-per-pixel) uint8 array; with want_gainmap, (pixels, gain-map image
    as uhdr_get_decoded_gainmap_image hands it out).  decodes > 1: uhdr_reset_decoder + the same decode again on the same
    handle before the results are read (the facade's lazy gain-map download across a reset)."""
    global last_call_seconds, last_gainmap_seconds
    import time

    lib = load()
    h = lib.uhdr_create_decoder()
    try:
        _add_effects(lib, h, effects)
        buf = (C.c_uint8 * len(jpeg)).from_buffer_copy(jpeg)
        ci = CompressedImage(C.cast(buf, C.c_void_p), len(jpeg), len(jpeg), 0, 0, 0)
        _chk(lib.uhdr_dec_set_image(h, C.byref(ci)))
        _chk(lib.uhdr_dec_set_out_color_transfer(h, out_ct))
        _chk(lib.uhdr_dec_set_out_img_format(h, out_fmt))
        for k in range(decodes):
            if k:
                lib.uhdr_reset_decoder(h)
                _add_effects(lib, h, effects)
                _chk(lib.uhdr_dec_set_image(h, C.byref(ci)))
                _chk(lib.uhdr_dec_set_out_color_transfer(h, out_ct))
                _chk(lib.uhdr_dec_set_out_img_format(h, out_fmt))
            if gpu:
                _chk(lib.uhdr_enable_gpu_acceleration(h, 1))
            t0 = time.perf_counter()
            st = lib.uhdr_decode(h)
            last_call_seconds = time.perf_counter() - t0
            _chk(st)
        px = _packed(lib.uhdr_get_decoded_image(h).contents)
        if not want_gainmap:
            return px
        t0 = time.perf_counter()
        g = lib.uhdr_get_decoded_gainmap_image(h)
        last_gainmap_seconds = time.perf_counter() - t0
        if not g:
            raise A.UhdrError(1, "uhdr_get_decoded_gainmap_image returned NULL")
        gm = _packed(g.contents)
        g2 = lib.uhdr_get_decoded_gainmap_image(h)  # asking twice hands out the same image
        assert g2 and np.array_equal(gm, _packed(g2.contents))
        return px, gm
    finally:
        lib.uhdr_release_decoder(h)
