#!/usr/bin/env python
"""Generates tests/golden/hotpath_golden.npz from the REAL reference (oracle/_ref, built from
/root/reference by `make -C oracle ref`).  Run in the build container only:

    python tests/golden/make_golden.py

Every case stores what is needed to re-create the inputs (they come from libultrahdr_amd.synth with
fixed seeds, or are stored verbatim when random) plus the reference's output planes.  The test
suite (tests/test_golden.py) checks the C oracle against these on any machine and the HIP path on
the GPU box, where /root/reference does not exist."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from libultrahdr_amd import capi as A  # noqa: E402
from oracle import loader as L  # noqa: E402
import golden_cases as G  # noqa: E402


def main():
    assert L.ref() is not None, "oracle/_ref is not built"
    out = {"_info": np.frombuffer(L.ref().ref_info(), dtype=np.uint8)}
    for name, case in G.cases().items():
        res = G.run(case, "ref")
        for k, v in res.items():
            out[f"{name}/{k}"] = v
    # JPEG stage: quantized coefficients straight from libjpeg (jpeg_read_coefficients)
    from test_oracle_vs_ref import _read_coefficients

    for q in (95, 50):
        img = G.jpeg_image()
        buf = np.zeros(1 << 20, dtype=np.uint8)
        n = L.ref().ref_jpeg_compress(C.byref(img.raw), q, buf.ctypes.data, buf.size)
        coefs, qt = _read_coefficients(L.ref(), buf[:n].tobytes())
        for c in range(3):
            out[f"jpeg_q{q}/coef{c}"] = coefs[c]
            out[f"jpeg_q{q}/qt{c}"] = qt[c]
        # entropy stage: the reference encoder's own entropy-coded bytes (Annex K tables, no restart markers)
        from test_oracle_vs_ref import _scan_data

        out[f"jpeg_q{q}/scan"] = np.frombuffer(_scan_data(buf[:n].tobytes()), dtype=np.uint8).copy()
        # decode stage: the planes JpegDecoderHelper (libjpeg, JDCT_ISLOW, raw-data mode) returns for that JPEG
        from test_oracle_vs_ref import _decode_with_reference

        dst, store = _decode_with_reference(L.ref(), buf[:n].tobytes(), 0)
        off = 0
        for c in range(3):
            pw, ph = (G.W, G.H) if c == 0 else (G.W // 2, G.H // 2)
            st = dst.stride[c]
            out[f"jpeg_q{q}/dec{c}"] = store[off: off + st * ph].reshape(ph, st)[:, :pw].copy()
            off += st * ph
        # 3-channel gain map: JCS_RGB in, RGB scanlines out (IJG 9 colour constants = the library linked here)
        gm = G.jpeg_rgb_map()
        n = L.ref().ref_jpeg_compress(C.byref(gm.raw), q, buf.ctypes.data, buf.size)
        coefs, qt = _read_coefficients(L.ref(), buf[:n].tobytes())
        for c in range(3):
            out[f"jpegrgb_q{q}/coef{c}"] = coefs[c]
            out[f"jpegrgb_q{q}/qt{c}"] = qt[c]
        out[f"jpegrgb_q{q}/scan"] = np.frombuffer(_scan_data(buf[:n].tobytes()), dtype=np.uint8).copy()
        dst, store = _decode_with_reference(L.ref(), buf[:n].tobytes(), 1)
        bpp = 4 if dst.fmt == A.UHDR_IMG_FMT_32bppRGBA8888 else 3
        out[f"jpegrgb_q{q}/dec_rgb"] = store[: dst.stride[0] * bpp * gm.h].reshape(gm.h, dst.stride[0] * bpp)[:, : gm.w * bpp].copy()
        out[f"jpegrgb_q{q}/dec_bpp"] = np.array([bpp])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
