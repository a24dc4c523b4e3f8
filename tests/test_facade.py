"""CPU-side checks of the drop-in libuhdr.so: symbol surface, the unaccelerated path is the
reference's, and asking for acceleration without a GPU fails loudly instead of computing on the CPU."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import facade_util as F
from tests import fixture720

pytestmark = pytest.mark.skipif(not F.built(), reason="facade not built (needs /root/reference at build time)")


def test_facade_exports_exactly_the_43_api_symbols():
    out = subprocess.run(["nm", "-D", "--defined-only", F.FACADE], capture_output=True, text=True, check=True).stdout
    syms = sorted(l.split()[2] for l in out.splitlines() if len(l.split()) == 3 and l.split()[1] == "T")
    assert syms == sorted(F.API_SYMBOLS)


def _write_fixture(d):
    g = fixture720.gold()
    g["p010"].tofile(os.path.join(d, "in.p010"))
    g["yuv420"].tofile(os.path.join(d, "in.yuv420"))
    return os.path.join(d, "in.p010"), os.path.join(d, "in.yuv420")


def test_unaccelerated_facade_is_the_reference_on_config_1():
    """BASELINE config 1 through the relinked sample app without -u: the reference's CPU path, 85 449 bytes."""
    with tempfile.TemporaryDirectory() as d:
        p, y = _write_fixture(d)
        rc, _, err, trace = F.encode_api1(p, y, 1280, 720, "cpu.jpg", False, d)
        assert rc == 0, err
        assert trace == {}  # no stage touched the seam
        assert os.path.getsize(os.path.join(d, "cpu.jpg")) == 85449


def test_acceleration_without_a_gpu_fails_loudly():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with tempfile.TemporaryDirectory() as d:
        p, y = _write_fixture(d)
        rc, _, err, _ = F.encode_api1(p, y, 1280, 720, "gpu.jpg", True, d)
        assert rc != 0
        assert "no CPU fallback" in err
        assert not os.path.exists(os.path.join(d, "gpu.jpg"))


def test_unaccelerated_decode_hands_out_the_gainmap_image_as_before():
    """The round-4 seam hooks in decodeJPEGR / uhdr_get_decoded_gainmap_image / uhdr_reset_decoder (lazy download of the decoded
    gain-map image) are inert without uhdr_enable_gpu_acceleration: the image is there after uhdr_decode, the same across a
    reset and a second decode on the handle, the same for SDR output, and mirrored when an effect is queued."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import facade as FA
    from libultrahdr_amd import synth

    w, h = 256, 128
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
    sdr = synth.make_sdr_yuv420(w, h)
    lin, f16 = A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    for multi, scale, shape in ((1, 1, (h, w, 4)), (0, 4, (h // 4, w // 4, 1))):
        jpg = FA.encode(hdr, sdr, gpu=False, preset=A.UHDR_USAGE_REALTIME, multi_channel=multi, scale=scale)
        px, gm = FA.decode(jpg, lin, f16, gpu=False, want_gainmap=True)
        assert gm.shape == shape and gm.any() and px.shape == (h, w, 8)
        if multi:
            assert (gm[..., 3] == 255).all()  # copy_raw_image's RGB888 -> RGBA8888 (IJG libjpeg build) or libjpeg-turbo's RGBA
        px2, gm2 = FA.decode(jpg, lin, f16, gpu=False, want_gainmap=True, decodes=2)
        assert np.array_equal(px, px2) and np.array_equal(gm, gm2)
        _, gm3 = FA.decode(jpg, A.UHDR_CT_SRGB, A.UHDR_IMG_FMT_32bppRGBA8888, gpu=False, want_gainmap=True)
        assert np.array_equal(gm, gm3)
        _, gm4 = FA.decode(jpg, lin, f16, gpu=False, effects=[("mirror", 0)], want_gainmap=True)
        assert np.array_equal(gm4[::-1], gm)


def test_every_encode_variant_compresses_the_gain_map_right_after_generating_it():
    """The generate seam leaves the map in HBM (lazy download) for the compress seam that follows.  That rests on the reference
    calling compressGainMap on the same image directly after generateGainMap in every encodeJPEGR variant, with no host code
    reading `gainmap` in between (jpegr.cpp:211-217, 253-257, 316-320, 377-381).  Pinned here against the reference source the
    facade is built from: a release that puts a reader between the two calls fails this test instead of reading unwritten memory."""
    import os
    import re

    src = "/root/reference/lib/src/jpegr.cpp"
    if not os.path.exists(src):
        pytest.skip("reference source not on this machine")
    lines = open(src).read().split("\n")
    gen = [i for i, l in enumerate(lines) if re.search(r"\bgenerateGainMap\(", l) and "UHDR_ERR_CHECK" in "".join(lines[max(0, i - 1): i + 1])]
    assert len(gen) == 4, gen
    for g in gen:
        j = g
        while not lines[j].rstrip().endswith(";"):
            j += 1
        between = []
        k = j + 1
        while "compressGainMap(" not in lines[k]:
            code = lines[k].split("//")[0].strip()
            if code:
                between.append(code)
            k += 1
            assert k - j < 12, (g, "compressGainMap does not follow generateGainMap")
        assert "compressGainMap(gainmap.get()" in lines[k]
        # what may stand between: the declaration of the JPEG encoder object that compressGainMap fills -- nothing that names the map
        assert all("gainmap" not in c.replace("jpeg_enc_obj_gm", "") for c in between), (g, between)


def test_zero_page_blocks_behave_like_the_containers_they_replace(tmp_path):
    """facade/uhdr_zero_pages.h (the calloc-backed blocks the patch puts behind uhdr_memory_block and
    JpegDecoderHelper::mResultBuffer) against std::vector<uint8_t>: 4000 random clear / resize / write steps, contents equal after
    each -- new elements are zero on fresh, recycled, grown and reallocated blocks."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "zero_pages_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(root, "facade"), os.path.join(root, "tests", "cpp", "zero_pages_check.cpp"), "-o", exe],
                   check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "zero_pages ok" in r.stdout, r.stdout + r.stderr
