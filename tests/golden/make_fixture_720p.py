#!/usr/bin/env python
"""Generates tests/golden/fixture_720p.npz from the reference's own raw test images
(/root/reference/tests/data/raw_p010_image.p010 + raw_yuv420_image.yuv420, 1280x720 -- BASELINE config 1's inputs)
and the REAL reference (oracle/_ref).  Run in the build container only:

    python tests/golden/make_fixture_720p.py

Stored: the two input images verbatim (they are smooth synthetic content: ~30 KB compressed) and what the reference's
API-1 stages produce for them with ultrahdr_app's defaults (hdr: P3 / HLG / narrow range, sdr: BT.709; C-API encoder
defaults: scale 1, multi-channel map, best quality, gamma 1):
  generateGainMap -> map bytes + metadata        (jpegr.cpp:255-258)
  convertYuv(sdr, bt709 -> P3/601)               (jpegr.cpp:281)
  libjpeg coefficients of the base image (q95) and of the map (q95)    (compressImage / compressGainMap)
  applyGainMap(sdr, map, metadata) -> linear F16 (crc32) / HLG / PQ 1010102 (crc32)  (decode direction, jpegr.cpp:1533)
tests/test_golden.py::test_fixture_720p_* checks the C oracle against these on any machine; tests/test_gpu_fullframe.py
checks the HIP path against them on the GPU box, where /root/reference does not exist."""
import ctypes as C
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from libultrahdr_amd import capi as A  # noqa: E402
from libultrahdr_amd.images import Image  # noqa: E402
from oracle import loader as L  # noqa: E402

W, H = 1280, 720
DATA = "/root/reference/tests/data"


def load_inputs(p010_words, yuv_bytes):
    hdr = Image(A.UHDR_IMG_FMT_24bppYCbCrP010, W, H, A.UHDR_CG_DISPLAY_P3, A.UHDR_CT_HLG, A.UHDR_CR_LIMITED_RANGE, align=64)
    hdr.valid(0)[:] = p010_words[: W * H].reshape(H, W)
    hdr.valid(1)[:] = p010_words[W * H:].reshape(H // 2, W)
    sdr = Image(A.UHDR_IMG_FMT_12bppYCbCr420, W, H, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=64)
    sdr.valid(0)[:] = yuv_bytes[: W * H].reshape(H, W)
    sdr.valid(1)[:] = yuv_bytes[W * H: W * H + W * H // 4].reshape(H // 2, W // 2)
    sdr.valid(2)[:] = yuv_bytes[W * H + W * H // 4:].reshape(H // 2, W // 2)
    return sdr, hdr


def crc(a):
    return np.array([zlib.crc32(np.ascontiguousarray(a).tobytes())], dtype=np.uint32)


def main():
    assert L.ref() is not None, "oracle/_ref is not built"
    from test_oracle_vs_ref import _read_coefficients

    p010 = np.fromfile(os.path.join(DATA, "raw_p010_image.p010"), dtype=np.uint16)
    yuv = np.fromfile(os.path.join(DATA, "raw_yuv420_image.yuv420"), dtype=np.uint8)
    sdr, hdr = load_inputs(p010, yuv)
    out = {"p010": p010, "yuv420": yuv, "_info": np.frombuffer(L.ref().ref_info(), dtype=np.uint8)}
    cfg = A.default_encode_cfg()
    md, gm = L.generate_gainmap("ref", sdr, hdr, cfg)
    out["gainmap"] = gm.valid(0).copy()
    out["metadata"] = np.array([list(md.max_content_boost), list(md.min_content_boost), list(md.gamma), list(md.offset_sdr), list(md.offset_hdr),
                                [md.hdr_capacity_min, md.hdr_capacity_max, float(md.use_base_cg)]], dtype=np.float32)
    conv = L.convert_yuv("ref", sdr, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3)
    for c in range(3):
        out[f"sdr601_{c}"] = conv.valid(c).copy()
    buf = np.zeros(8 << 20, dtype=np.uint8)
    n = L.ref().ref_jpeg_compress(C.byref(conv.raw), 95, buf.ctypes.data, buf.size)
    coefs, qt = _read_coefficients(L.ref(), buf[:n].tobytes())
    for c in range(3):
        out[f"base_coef{c}"], out[f"base_qt{c}"] = coefs[c], qt[c]
    n = L.ref().ref_jpeg_compress(C.byref(gm.raw), 95, buf.ctypes.data, buf.size)
    coefs, qt = _read_coefficients(L.ref(), buf[:n].tobytes())
    for c in range(3):
        out[f"map_coef{c}"], out[f"map_qt{c}"] = coefs[c], qt[c]
    for name, ct in (("linear", A.UHDR_CT_LINEAR), ("hlg", A.UHDR_CT_HLG), ("pq", A.UHDR_CT_PQ)):
        res = L.apply_gainmap("ref", sdr, gm, md, ct)
        out[f"apply_{name}_crc"] = crc(res.valid(0))
        out[f"apply_{name}_rows"] = res.valid(0)[::45].copy()  # 16 full rows for a readable diff when the crc fails
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fixture_720p.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
