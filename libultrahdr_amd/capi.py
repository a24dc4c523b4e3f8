"""ctypes binding of the C ABI in ``include/uhdr_hip.h`` (``libultrahdr_amd/lib/libuhdr_hip.so``).

This is the *only* way the Python host layer reaches the kernels -- the same entry points a cgo /
JNI / C++ caller binds.  There is no CPU fallback: if the shared library is
missing the import of :func:`load` raises, and if no GPU is usable ``uhdr_hip_create`` fails.

Struct layouts mirror the reference's public C structs
(``/root/reference/ultrahdr_api.h:220-283``).
"""
from __future__ import annotations

import ctypes as C
import os

# ---- enums (ultrahdr_api.h:108-207) -----------------------------------------------------------
UHDR_IMG_FMT_UNSPECIFIED = -1
UHDR_IMG_FMT_24bppYCbCrP010 = 0
UHDR_IMG_FMT_12bppYCbCr420 = 1
UHDR_IMG_FMT_8bppYCbCr400 = 2
UHDR_IMG_FMT_32bppRGBA8888 = 3
UHDR_IMG_FMT_64bppRGBAHalfFloat = 4
UHDR_IMG_FMT_32bppRGBA1010102 = 5
UHDR_IMG_FMT_24bppYCbCr444 = 6
UHDR_IMG_FMT_16bppYCbCr422 = 7
UHDR_IMG_FMT_24bppRGB888 = 11
UHDR_IMG_FMT_30bppYCbCr444 = 12

UHDR_CG_UNSPECIFIED, UHDR_CG_BT_709, UHDR_CG_DISPLAY_P3, UHDR_CG_BT_2100 = -1, 0, 1, 2
UHDR_CT_UNSPECIFIED, UHDR_CT_LINEAR, UHDR_CT_HLG, UHDR_CT_PQ, UHDR_CT_SRGB = -1, 0, 1, 2, 3
UHDR_CR_UNSPECIFIED, UHDR_CR_LIMITED_RANGE, UHDR_CR_FULL_RANGE = -1, 0, 1
UHDR_USAGE_REALTIME, UHDR_USAGE_BEST_QUALITY = 0, 1

UHDR_CODEC_OK = 0
UHDR_CODEC_ERROR = 1
UHDR_CODEC_UNKNOWN_ERROR = 2
UHDR_CODEC_INVALID_PARAM = 3
UHDR_CODEC_MEM_ERROR = 4
UHDR_CODEC_INVALID_OPERATION = 5
UHDR_CODEC_UNSUPPORTED_FEATURE = 6

FLT_MIN = 1.1754943508222875e-38
FLT_MAX = 3.4028234663852886e38


class ErrorInfo(C.Structure):  # uhdr_error_info_t
    _fields_ = [("error_code", C.c_int), ("has_detail", C.c_int), ("detail", C.c_char * 256)]


class RawImage(C.Structure):  # uhdr_raw_image_t
    _fields_ = [
        ("fmt", C.c_int),
        ("cg", C.c_int),
        ("ct", C.c_int),
        ("range", C.c_int),
        ("w", C.c_uint),
        ("h", C.c_uint),
        ("planes", C.c_void_p * 3),
        ("stride", C.c_uint * 3),
    ]


class GainmapMetadata(C.Structure):  # uhdr_gainmap_metadata_t
    _fields_ = [
        ("max_content_boost", C.c_float * 3),
        ("min_content_boost", C.c_float * 3),
        ("gamma", C.c_float * 3),
        ("offset_sdr", C.c_float * 3),
        ("offset_hdr", C.c_float * 3),
        ("hdr_capacity_min", C.c_float),
        ("hdr_capacity_max", C.c_float),
        ("use_base_cg", C.c_int),
    ]

    def as_dict(self):
        return {
            "max_content_boost": list(self.max_content_boost),
            "min_content_boost": list(self.min_content_boost),
            "gamma": list(self.gamma),
            "offset_sdr": list(self.offset_sdr),
            "offset_hdr": list(self.offset_hdr),
            "hdr_capacity_min": self.hdr_capacity_min,
            "hdr_capacity_max": self.hdr_capacity_max,
            "use_base_cg": self.use_base_cg,
        }


class EncodeCfg(C.Structure):  # uhdr_hip_encode_cfg_t
    _fields_ = [
        ("map_dimension_scale_factor", C.c_int),
        ("use_multi_channel_gainmap", C.c_int),
        ("gamma", C.c_float),
        ("preset", C.c_int),
        ("min_content_boost", C.c_float),
        ("max_content_boost", C.c_float),
        ("target_disp_peak_nits", C.c_float),
        ("sdr_is_601", C.c_int),
        ("use_luminance", C.c_int),
    ]


class JpegCoefficients(C.Structure):  # uhdr_hip_jpeg_coefficients_t
    _fields_ = [
        ("coef", C.c_void_p * 3),
        ("blocks_w", C.c_int * 3),
        ("blocks_h", C.c_int * 3),
        ("qtable", (C.c_uint16 * 64) * 3),
    ]


class JpegScan(C.Structure):  # uhdr_hip_jpeg_scan_t
    _fields_ = [
        ("num_components", C.c_int),
        ("coef", C.c_void_p * 3),
        ("blocks_w", C.c_int * 3),
        ("blocks_h", C.c_int * 3),
        ("h_samp", C.c_int * 3),
        ("v_samp", C.c_int * 3),
        ("w", C.c_uint),
        ("h", C.c_uint),
        ("restart_interval", C.c_int),
    ]


class HuffTables(C.Structure):  # uhdr_hip_huff_tables_t
    _fields_ = [("bits", (C.c_uint8 * 17) * 4), ("vals", (C.c_uint8 * 256) * 4)]


class JpegHeader(C.Structure):  # uhdr_hip_jpeg_header_t
    _fields_ = [("scan", JpegScan), ("tables", HuffTables), ("qtable", (C.c_uint16 * 64) * 3), ("scan_offset", C.c_size_t),
                ("scan_bytes", C.c_size_t)]


def default_encode_cfg(**kw) -> EncodeCfg:
    """C-API defaults (ultrahdrcommon.h:422-446): scale 1, multichannel, gamma 1, two-pass."""
    cfg = EncodeCfg(1, 1, 1.0, UHDR_USAGE_BEST_QUALITY, FLT_MIN, FLT_MAX, -1.0, 0, 1)
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise AttributeError(k)
        setattr(cfg, k, v)
    return cfg


class UhdrError(RuntimeError):
    def __init__(self, code: int, detail: str):
        super().__init__(f"uhdr error {code}: {detail}")
        self.code = code
        self.detail = detail


def check(st: ErrorInfo):
    if st.error_code != UHDR_CODEC_OK:
        raise UhdrError(st.error_code, st.detail.decode("utf-8", "replace") if st.has_detail else "")


# uhdr_hip_comm_ops_t: a caller-provided transport for the exchange steps (include/uhdr_hip.h)
ALL_REDUCE_MIN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
GATHER_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_int, C.c_void_p)


class Api1Blocks(C.Structure):  # uhdr_hip_api1_blocks_t
    _fields_ = [("base_coef", C.c_void_p * 3), ("map_coef", C.c_void_p * 3)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_ulonglong) for n in ("entropy_decode_parallel", "entropy_decode_intervals", "entropy_decode_single_lane",
                                              "entropy_decode_declined", "entropy_encode_stream", "entropy_encode_intervals", "resident_hits",
                                              "generate_channels_tabled", "generate_channels_per_sample", "lazy_downloads_skipped",
                                              "lazy_downloads_done", "last_jpeg_decode_scan_ns", "last_encode_api1_scans_ns")]


class SeamStage(C.Structure):  # uhdr_hip_seam_stage_t
    _fields_ = [("name", C.c_char * 40), ("device_calls", C.c_ulonglong), ("reference_calls", C.c_ulonglong), ("first_seq", C.c_ulonglong),
                ("device_ms", C.c_double), ("last_ms", C.c_double)]


class CommOps(C.Structure):
    _fields_ = [("user", C.c_void_p), ("all_reduce_min_f32", ALL_REDUCE_MIN_FN), ("all_gather", ALL_GATHER_FN), ("gather_v", GATHER_V_FN)]


LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
LIB_PATH = os.path.join(LIB_DIR, "libuhdr_hip.so")

# every symbol include/uhdr_hip.h declares: (restype, argtypes)
_P = C.POINTER
_SIGS = {
    "uhdr_hip_version": (C.c_char_p, []),
    "uhdr_hip_device_count": (C.c_int, []),
    "uhdr_hip_create": (C.c_void_p, [C.c_int, _P(ErrorInfo)]),
    "uhdr_hip_destroy": (None, [C.c_void_p]),
    "uhdr_hip_set_stream": (ErrorInfo, [C.c_void_p, C.c_void_p]),
    "uhdr_hip_get_stream": (C.c_void_p, [C.c_void_p]),
    "uhdr_hip_synchronize": (ErrorInfo, [C.c_void_p]),
    "uhdr_hip_apply_gainmap": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(GainmapMetadata), C.c_int, C.c_int, C.c_float, _P(RawImage)]),
    "uhdr_hip_apply_gainmap_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(GainmapMetadata), C.c_int, C.c_int, C.c_float, _P(RawImage), C.c_uint, C.c_uint]),
    "uhdr_hip_apply_gainmap_batch_dev": (ErrorInfo, [C.c_void_p, C.c_uint, _P(RawImage), _P(RawImage), _P(GainmapMetadata), C.c_int, C.c_int, C.c_float, _P(RawImage)]),
    "uhdr_hip_generate_gainmap": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(EncodeCfg), _P(GainmapMetadata), _P(RawImage)]),
    "uhdr_hip_generate_gainmap_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(EncodeCfg), _P(GainmapMetadata), _P(RawImage)]),
    "uhdr_hip_generate_gainmap_pass1_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(EncodeCfg), C.c_void_p, C.c_void_p, _P(C.c_int)]),
    "uhdr_hip_generate_gainmap_finalize": (ErrorInfo, [_P(EncodeCfg), C.c_int, C.c_int, _P(C.c_float), _P(GainmapMetadata)]),
    "uhdr_hip_generate_gainmap_pass2_dev": (ErrorInfo, [C.c_void_p, C.c_void_p, _P(C.c_float), _P(EncodeCfg), _P(RawImage)]),
    "uhdr_hip_tone_map": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_tone_map_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_convert_yuv": (ErrorInfo, [C.c_void_p, _P(RawImage), C.c_int, C.c_int]),
    "uhdr_hip_convert_yuv_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), C.c_int, C.c_int]),
    "uhdr_hip_convert_raw_input_to_ycbcr": (ErrorInfo, [C.c_void_p, _P(RawImage), C.c_int, _P(RawImage)]),
    "uhdr_hip_convert_raw_input_to_ycbcr_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), C.c_int, _P(RawImage)]),
    "uhdr_hip_jpeg_quant_table": (None, [C.c_int, C.c_int, _P(C.c_uint16)]),
    "uhdr_hip_oetf_code_thresholds": (C.c_int, [C.c_int, _P(C.c_float)]),
    "uhdr_hip_exact_math_eval": (C.c_int, [C.c_int, _P(C.c_float), _P(C.c_float), C.c_size_t]),
    "uhdr_hip_encode_api1_fused_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(EncodeCfg), C.c_int, C.c_void_p, C.c_void_p, _P(Api1Blocks),
                                                   _P(GainmapMetadata), _P(RawImage)]),
    "uhdr_hip_huffman_encode2_dev": (ErrorInfo, [C.c_void_p, _P(JpegScan), C.c_void_p, C.c_size_t, _P(C.c_size_t), _P(JpegScan), C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "uhdr_hip_huffman_decode2_dev": (ErrorInfo, [C.c_void_p, _P(JpegScan), _P(HuffTables), C.c_void_p, C.c_size_t, _P(JpegScan), _P(HuffTables), C.c_void_p, C.c_size_t]),
    "uhdr_hip_encode_api1_scans": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(EncodeCfg), C.c_int, C.c_void_p, C.c_void_p, _P(GainmapMetadata),
                                               _P(RawImage), C.c_void_p, C.c_size_t, _P(C.c_size_t), C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "uhdr_hip_encode_api0_scans": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(EncodeCfg), C.c_void_p, C.c_void_p, _P(GainmapMetadata), _P(RawImage), _P(C.c_int),
                                               C.c_void_p, C.c_size_t, _P(C.c_size_t), C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "uhdr_hip_encode_api1_scans_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(EncodeCfg), C.c_int, C.c_void_p, C.c_void_p, _P(GainmapMetadata),
                                                   _P(RawImage), C.c_void_p, C.c_size_t, _P(C.c_size_t), C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "uhdr_hip_decode_api1_scans_dev": (ErrorInfo, [C.c_void_p, _P(JpegHeader), C.c_void_p, C.c_size_t, C.c_int, _P(JpegHeader), C.c_void_p, C.c_size_t, C.c_int,
                                                   C.c_int, _P(GainmapMetadata), C.c_int, C.c_int, C.c_float, _P(RawImage)]),
    "uhdr_hip_comm_all_reduce_min_dev": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "uhdr_hip_selftest": (ErrorInfo, [C.c_void_p, C.c_int, C.c_uint, C.c_uint, C.c_uint, _P(C.c_float), _P(C.c_ulonglong)]),
    "uhdr_hip_fdct_quant": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, _P(C.c_uint16), C.c_void_p]),
    "uhdr_hip_fdct_quant_dev": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, _P(C.c_uint16), C.c_void_p]),
    "uhdr_hip_encode_api0_fused_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(EncodeCfg), _P(RawImage), _P(RawImage), _P(GainmapMetadata), _P(RawImage)]),
    "uhdr_hip_copy_raw_image_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_fdct_quant_rgb_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(C.c_uint16), _P(C.c_uint16), C.c_void_p, C.c_void_p, C.c_void_p]),
    "uhdr_hip_idct_dequant": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _P(C.c_uint16), C.c_void_p, C.c_size_t]),
    "uhdr_hip_idct_dequant_dev": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, _P(C.c_uint16), C.c_void_p, C.c_size_t]),
    "uhdr_hip_jpeg_rgb_to_ycc": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_jpeg_rgb_to_ycc_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_jpeg_ycc_to_rgb": (ErrorInfo, [C.c_void_p, _P(RawImage), C.c_int, _P(RawImage)]),
    "uhdr_hip_jpeg_ycc_to_rgb_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), C.c_int, _P(RawImage)]),
    "uhdr_hip_idct_dequant_rgb_dev": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _P(C.c_uint16), _P(C.c_uint16),
                                                  C.c_int, _P(RawImage)]),
    "uhdr_hip_apply_gainmap_coef_dev": (ErrorInfo, [C.c_void_p, _P(JpegCoefficients), C.c_uint, C.c_uint, C.c_int, _P(RawImage), _P(GainmapMetadata),
                                                    C.c_int, C.c_int, C.c_float, _P(RawImage)]),
    "uhdr_hip_huffman_encode_dev": (ErrorInfo, [C.c_void_p, _P(JpegScan), C.c_void_p, C.c_size_t, _P(C.c_size_t)]),
    "uhdr_hip_huffman_decode_dev": (ErrorInfo, [C.c_void_p, _P(JpegScan), _P(HuffTables), C.c_void_p, C.c_size_t]),
    "uhdr_hip_jpeg_parse": (C.c_int, [C.c_void_p, C.c_size_t, _P(JpegHeader)]),
    "uhdr_hip_jpeg_decode_scan": (ErrorInfo, [C.c_void_p, _P(JpegHeader), C.c_void_p, C.c_size_t, C.c_int, C.c_int, _P(C.c_void_p), _P(C.c_uint), _P(C.c_uint)]),
    "uhdr_hip_jpeg_encode_scan": (ErrorInfo, [C.c_void_p, _P(JpegScan), C.c_void_p, _P(C.c_void_p), _P(C.c_uint), C.c_int, C.c_void_p, C.c_size_t,
                                              _P(C.c_size_t)]),
    "uhdr_hip_jpeg_encode_image": (ErrorInfo, [C.c_void_p, _P(JpegScan), C.c_void_p, _P(C.c_void_p), _P(C.c_uint), C.c_int, C.c_void_p, C.c_size_t,
                                              _P(C.c_size_t)]),
    "uhdr_hip_jpeg_assemble": (C.c_size_t, [_P(JpegScan), _P(C.c_uint16), _P(C.c_uint16), C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "uhdr_hip_apply_effect": (ErrorInfo, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_apply_effect_dev": (ErrorInfo, [C.c_void_p, C.c_int, C.c_int, C.c_int, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_step_table_eval": (C.c_int, [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "uhdr_hip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "uhdr_hip_comm_init": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "uhdr_hip_comm_destroy": (None, [C.c_void_p]),
    "uhdr_hip_comm_size": (C.c_int, [C.c_void_p]),
    "uhdr_hip_comm_rank": (C.c_int, [C.c_void_p]),
    "uhdr_hip_comm_init_custom": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "uhdr_hip_comm_all_gather_dev": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "uhdr_hip_comm_gather_dev": (ErrorInfo, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, _P(C.c_size_t), C.c_int]),
    "uhdr_hip_generate_gainmap_striped_dev": (ErrorInfo, [C.c_void_p, _P(RawImage), _P(RawImage), _P(EncodeCfg), _P(GainmapMetadata), _P(RawImage)]),
    "uhdr_hip_get_stats": (None, [C.c_void_p, C.c_void_p]),
    "uhdr_hip_recycle": (C.c_int, [C.c_void_p, C.c_size_t]),
    "uhdr_hip_current_device": (C.c_int, []),
    "uhdr_hip_seam_note": (None, [C.c_char_p, C.c_int, C.c_double]),
    "uhdr_hip_seam_stats": (C.c_int, [C.c_void_p, C.c_int]),
    "uhdr_hip_seam_stats_reset": (None, []),
    "uhdr_hip_resident_begin": (None, [C.c_void_p]),
    "uhdr_hip_resident_end": (None, [C.c_void_p]),
    "uhdr_hip_resident_lazy": (None, [C.c_void_p, C.c_int]),
    "uhdr_hip_resident_flush": (ErrorInfo, [C.c_void_p]),
    "uhdr_hip_resident_adopt": (C.c_int, [C.c_void_p, _P(RawImage), _P(RawImage)]),
    "uhdr_hip_resident_materialize": (ErrorInfo, [C.c_void_p]),
    "uhdr_hip_resident_forget": (None, [C.c_void_p]),
    "uhdr_hip_profile_enable": (None, [C.c_void_p, C.c_int]),
    "uhdr_hip_profile_read": (C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_double), C.c_int]),
    "uhdr_hip_profile_read_list": (C.c_int, [C.c_void_p, C.c_char_p, _P(C.c_double), C.c_int, C.c_int]),
    "uhdr_hip_profile_mark": (None, [C.c_void_p]),
}
ABI_SYMBOLS = tuple(_SIGS)

_lib = None


def load() -> C.CDLL:
    """Load libuhdr_hip.so (built in-tree by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  Device memory
    # handed to us comes from torch, so torch's HIP runtime must be the one this process uses: import
    # it first, then our DT_NEEDED libamdhip64.so.7 binds to the already-loaded copy.  (Loading ours
    # first gives the process two HIP runtimes and torch then reports "No HIP GPUs are available".)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C libultrahdr_amd/csrc).  libultrahdr_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError here == ABI drift
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def seam_stats(reset: bool = False) -> dict:
    """The facade's stage tallies (uhdr_hip_seam_stats): {stage: {"device": n, "reference": n, "device_ms": t, "last_ms": t}} in
    the order the stages were first seen since the last reset."""
    lib = load()
    rows = (SeamStage * 64)()
    n = min(lib.uhdr_hip_seam_stats(rows, 64), 64)
    out = {}
    for r in sorted(rows[:n], key=lambda r: r.first_seq):
        out[r.name.decode()] = {"device": int(r.device_calls), "reference": int(r.reference_calls), "device_ms": float(r.device_ms),
                                "last_ms": float(r.last_ms)}
    if reset:
        lib.uhdr_hip_seam_stats_reset()
    return out
