"""GPU: the reference's sample app, linked unmodified against the facade libuhdr.so, with -u 1
(uhdr_enable_gpu_acceleration) -- SURVEY.md 8f-3.  Accelerated runs must (a) really go through the device at every
stage the seam covers and (b) give the CPU reference's bytes: the encode-side operators are measured identical to
the reference, FDCT / IDCT / colour conversion are integer exact, so whole files and whole decoded
frames are compared byte for byte; the two float tails with a stated +-1 bar (tone map, generate) get that bar."""
import os
import tempfile

import numpy as np
import pytest

from tests import facade_util as F
from tests import fixture720

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not F.built(), reason="facade not built")]


def _stages(trace, where="device"):
    return trace.on(where)  # tests/facade_util.Stages: the library's own stage table of that process, not stderr text


def _fixture(d):
    g = fixture720.gold()
    g["p010"].tofile(os.path.join(d, "in.p010"))
    g["yuv420"].tofile(os.path.join(d, "in.yuv420"))
    return os.path.join(d, "in.p010"), os.path.join(d, "in.yuv420")


def test_config1_encode_through_the_facade_equals_the_cpu_reference():
    with tempfile.TemporaryDirectory() as d:
        p, y = _fixture(d)
        rc, _, err, _ = F.encode_api1(p, y, 1280, 720, "cpu.jpg", False, d)
        assert rc == 0, err
        rc, _, err, trace = F.encode_api1(p, y, 1280, 720, "gpu.jpg", True, d)
        assert rc == 0, err
        st = _stages(trace)
        # round 5: the seam at JpegR::encodeJPEGR (API-1) runs generateGainMap + compressGainMap + convertYuv + compressImage as ONE
        # device sequence (uhdr_hip_encode_api1_scans): a single stage line, none of the per-stage seams, no libjpeg entropy pass
        assert st == ["encode_api1_fused"], trace
        a, b = F.read(os.path.join(d, "cpu.jpg")), F.read(os.path.join(d, "gpu.jpg"))
        assert a.size == b.size == 85449
        assert np.array_equal(a, b), f"{int((a != b).sum())} differing bytes"
        # the per-stage seams (what the fused seam falls back to for geometries it declines): generateGainMap, convertYuv (BT.709 ->
        # the P3/601 encoding of the base JPEG) and both compressImage calls whole on the device -- the same file
        rc, _, err, trace = F.encode_api1(p, y, 1280, 720, "gpu1.jpg", True, d, env_extra={"UHDR_HIP_SEAM_NO_FUSED_ENCODE": "1"})
        assert rc == 0, err
        st = _stages(trace)
        assert "generate_gainmap" in st and "convert_yuv" in st and "jpeg_encode_scan" in st and "fdct_planes" not in st and "encode_api1_fused" not in st, trace
        assert trace.n("jpeg_encode_scan") == 2, trace
        assert np.array_equal(a, F.read(os.path.join(d, "gpu1.jpg")))
        # the older route (device FDCT, libjpeg's Huffman pass) still gives the same file
        rc, _, err, trace = F.encode_api1(p, y, 1280, 720, "gpu2.jpg", True, d, env_extra={"UHDR_HIP_SEAM_CPU_ENTROPY": "1"})
        assert rc == 0, err
        assert "fdct_planes" in _stages(trace) and "jpeg_encode_scan" not in _stages(trace), trace
        assert np.array_equal(a, F.read(os.path.join(d, "gpu2.jpg")))


def test_4k_encode_through_the_facade_equals_the_cpu_reference_file():
    """API-1 at 4K, three-channel full-resolution map (the C API default): the device-made file -- marker-less Huffman coding
    of 194 400 + 388 800 blocks in wavefront segments -- is the CPU reference's file byte for byte."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth

    w, h = 3840, 2160
    with tempfile.TemporaryDirectory() as d:
        hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
        sdr = synth.make_sdr_yuv420(w, h)
        np.concatenate([hdr.valid(0).ravel(), hdr.valid(1).ravel()]).tofile(os.path.join(d, "in.p010"))
        np.concatenate([sdr.valid(c).ravel() for c in range(3)]).tofile(os.path.join(d, "in.yuv420"))
        rc, _, err, _ = F.encode_api1("in.p010", "in.yuv420", w, h, "cpu.jpg", False, d)
        assert rc == 0, err
        rc, _, err, trace = F.encode_api1("in.p010", "in.yuv420", w, h, "gpu.jpg", True, d)
        assert rc == 0, err
        assert _stages(trace) == ["encode_api1_fused"], trace  # one device sequence for the whole call (round 5)
        a, b = F.read(os.path.join(d, "cpu.jpg")), F.read(os.path.join(d, "gpu.jpg"))
        assert a.size == b.size, (a.size, b.size)
        assert np.array_equal(a, b), f"{int((a != b).sum())} differing bytes"
        rc, _, err, trace = F.encode_api1("in.p010", "in.yuv420", w, h, "gpu1.jpg", True, d, env_extra={"UHDR_HIP_SEAM_NO_FUSED_ENCODE": "1"})
        assert rc == 0, err
        assert trace.n("jpeg_encode_scan") == 2 and "fdct_planes" not in _stages(trace), trace
        assert np.array_equal(a, F.read(os.path.join(d, "gpu1.jpg")))


@pytest.mark.parametrize("ct,fmt,bpp", [(0, 4, 8), (1, 5, 4), (2, 5, 4)])  # linear F16, HLG 1010102, PQ 1010102
def test_config1_decode_through_the_facade_equals_the_cpu_reference(ct, fmt, bpp):
    with tempfile.TemporaryDirectory() as d:
        p, y = _fixture(d)
        rc, _, err, _ = F.encode_api1(p, y, 1280, 720, "in.jpg", False, d)
        assert rc == 0, err
        rc, _, err, _ = F.decode("in.jpg", ct, fmt, "cpu.raw", False, d)
        assert rc == 0, err
        rc, _, err, trace = F.decode("in.jpg", ct, fmt, "gpu.raw", True, d)
        assert rc == 0, err
        st = _stages(trace)
        # the base image and the gain map are decoded on the device from their compressed bytes (entropy decode included)
        assert "apply_gainmap" in st and "jpeg_decode_scan" in st and "idct_planes" not in st, trace
        a, b = F.read(os.path.join(d, "cpu.raw")), F.read(os.path.join(d, "gpu.raw"))
        assert a.size == b.size == 1280 * 720 * bpp
        assert np.array_equal(a, b), f"{int((a != b).sum())} differing bytes"


def test_4k_decode_through_the_facade_equals_the_cpu_reference():
    """BASELINE config 2 at the API level: a 4K UltraHDR JPEG (encoded by the CPU reference path) -> RGBA_F16."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth

    w, h = 3840, 2160
    with tempfile.TemporaryDirectory() as d:
        hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
        sdr = synth.make_sdr_yuv420(w, h)
        np.concatenate([hdr.valid(0).ravel(), hdr.valid(1).ravel()]).tofile(os.path.join(d, "in.p010"))
        np.concatenate([sdr.valid(c).ravel() for c in range(3)]).tofile(os.path.join(d, "in.yuv420"))
        # realtime preset keeps the CPU encode of the test input short; the decode under test does not depend on it
        rc, _, err, _ = F.encode_api1("in.p010", "in.yuv420", w, h, "in.jpg", False, d, extra=("-D", 0))
        assert rc == 0, err
        rc, _, err, _ = F.decode("in.jpg", 0, 4, "cpu.raw", False, d)
        assert rc == 0, err
        rc, _, err, trace = F.decode("in.jpg", 0, 4, "gpu.raw", True, d)
        assert rc == 0, err
        assert "apply_gainmap" in _stages(trace), trace
        a, b = F.read(os.path.join(d, "cpu.raw")), F.read(os.path.join(d, "gpu.raw"))
        assert a.size == b.size == w * h * 8
        assert np.array_equal(a, b), f"{int((a != b).sum())} differing bytes"
        assert trace.n("jpeg_decode_scan") == 2, trace  # base image and gain map, from their compressed bytes


def test_a_scan_larger_than_the_first_guess_is_coded_again_not_handed_back():
    """The fused encode seams size their output buffers at one byte per coefficient; a busier stream makes the entropy coder report the size it needs
    (UHDR_CODEC_MEM_ERROR + the byte counts).  The seam then runs the device chain once more with room for that -- it used to fall to the per-stage seams,
    i.e. every intermediate over PCIe (round-5 advice).  UHDR_HIP_SEAM_TEST_SMALL_CAP shrinks the first guess so that an ordinary frame takes that route."""
    with tempfile.TemporaryDirectory() as d:
        p, y = _fixture(d)
        rc, _, err, _ = F.encode_api1(p, y, 1280, 720, "cpu.jpg", False, d)
        assert rc == 0, err
        rc, _, err, trace = F.encode_api1(p, y, 1280, 720, "gpu.jpg", True, d, env_extra={"UHDR_HIP_SEAM_TEST_SMALL_CAP": "256"})
        assert rc == 0, err
        assert _stages(trace) == ["encode_api1_fused"] and trace.n("encode_api1_fused") == 2 and trace.n("encode_api1_fused", "reference") == 0, trace
        assert np.array_equal(F.read(os.path.join(d, "cpu.jpg")), F.read(os.path.join(d, "gpu.jpg")))


def test_api0_encode_through_the_facade():
    """API-0 (HDR only): toneMap + generateGainMap + convert_raw_input_to_ycbcr on the device.  toneMap's sRGB OETF is
    correctly rounded on the device and faithfully rounded in glibc: the files may differ in the rare
    +-1 sample, so the check is on decoded pixels."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth

    w, h = 1280, 720
    with tempfile.TemporaryDirectory() as d:
        hdr = synth.make_hdr_rgba1010102(w, h, ct=A.UHDR_CT_PQ)
        hdr.valid(0).tofile(os.path.join(d, "in.raw"))
        args = ["-m", 0, "-p", "in.raw", "-w", w, "-h", h, "-a", 5, "-C", 2, "-t", 2, "-R", 1]
        rc, _, err, _ = F.run_app(args + ["-z", "cpu.jpg"], False, d)
        assert rc == 0, err
        rc, _, err, trace = F.run_app(args + ["-z", "gpu.jpg"], True, d)
        assert rc == 0, err
        # round 6: the seam at JpegR::encodeJPEGR API-0 -- tone map + one-pass gain map + RGB -> YCbCr 4:4:4 fused, FDCTs, both scans
        # Huffman-coded (uhdr_hip_encode_api0_scans) -- is ONE stage for the input BASELINE config 3 names (RGBA1010102) ...
        assert _stages(trace) == ["encode_api0_fused"], trace
        # ... and writes the file the five per-stage seams write (what it falls back to for P010 intents, scale factors other than 1, ...)
        rc, _, err, trace = F.run_app(args + ["-z", "gpu1.jpg"], True, d, env_extra={"UHDR_HIP_SEAM_NO_FUSED_ENCODE": "1"})
        assert rc == 0, err
        st = _stages(trace)
        assert "tone_map" in st and "generate_gainmap" in st and "convert_raw_input_to_ycbcr" in st and trace.n("jpeg_encode_scan") == 2, trace
        assert np.array_equal(F.read(os.path.join(d, "gpu.jpg")), F.read(os.path.join(d, "gpu1.jpg")))
        for name in ("cpu", "gpu"):
            rc, _, err, _ = F.decode(name + ".jpg", 0, 4, name + ".raw", False, d)
            assert rc == 0, err
        a = np.fromfile(os.path.join(d, "cpu.raw"), dtype=np.float16).astype(np.float32)
        b = np.fromfile(os.path.join(d, "gpu.raw"), dtype=np.float16).astype(np.float32)
        assert a.size == b.size == w * h * 4
        # a +-1 8-bit sample before the JPEG DCT moves a handful of decoded pixels slightly
        assert (a != b).mean() < 1e-3 and np.abs(a - b).max() < 0.25


def test_api0_half_float_and_declined_shapes_through_the_facade():
    """The fused API-0 seam also takes RGBA half float; a P010 intent and a scale factor other than 1 go through the per-stage seams.  Each
    against the CPU reference's decoded pixels with the bar of test_api0_encode_through_the_facade."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth

    w, h = 640, 360
    with tempfile.TemporaryDirectory() as d:
        synth.make_hdr_rgba_f16(w, h, specials=False).valid(0).tofile(os.path.join(d, "f16.raw"))
        hp = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
        np.concatenate([hp.valid(0).ravel(), hp.valid(1).ravel()]).tofile(os.path.join(d, "in.p010"))
        synth.make_hdr_rgba1010102(w, h, ct=A.UHDR_CT_PQ).valid(0).tofile(os.path.join(d, "in.raw"))
        cases = (("f16", ["-m", 0, "-p", "f16.raw", "-w", w, "-h", h, "-a", 4, "-C", 2, "-t", 0, "-R", 1], ["encode_api0_fused"]),
                 ("p010", ["-m", 0, "-p", "in.p010", "-w", w, "-h", h, "-a", 0, "-C", 2, "-t", 1, "-R", 0], None),
                 ("s2", ["-m", 0, "-p", "in.raw", "-w", w, "-h", h, "-a", 5, "-C", 2, "-t", 2, "-R", 1, "-s", 2], None))
        for name, args, want in cases:
            rc, _, err, _ = F.run_app(args + ["-z", name + "_cpu.jpg"], False, d)
            assert rc == 0, (name, err)
            rc, _, err, trace = F.run_app(args + ["-z", name + "_gpu.jpg"], True, d)
            assert rc == 0, (name, err)
            st = _stages(trace)
            if want is not None:
                assert st == want, (name, trace)
            else:
                assert "encode_api0_fused" not in st and "tone_map" in st and "generate_gainmap" in st and trace.n("jpeg_encode_scan") == 2, (name, trace)
            for side in ("cpu", "gpu"):
                rc, _, err, _ = F.decode(f"{name}_{side}.jpg", 0, 4, f"{name}_{side}.raw", False, d)
                assert rc == 0, (name, err)
            a = np.fromfile(os.path.join(d, name + "_cpu.raw"), dtype=np.float16).astype(np.float32)
            b = np.fromfile(os.path.join(d, name + "_gpu.raw"), dtype=np.float16).astype(np.float32)
            assert a.size == b.size == w * h * 4
            assert (a != b).mean() < 1e-3 and np.abs(a - b).max() < 0.25, (name, float((a != b).mean()), float(np.abs(a - b).max()))


@pytest.mark.parametrize("api", [2, 3])
def test_api2_and_api3_encode_through_the_facade(api):
    """API-2 (raw HDR + raw SDR + compressed SDR) and API-3 (raw HDR + compressed SDR), jpegr.cpp:294-384.  API-2: generateGainMap on the two raw
    intents (the compressed one is only parsed); API-3: the compressed SDR intent is decoded on the device from its bytes and generateGainMap
    runs with sdr_is_601 = true.  Either way the map is compressed on the device and the caller's JPEG is passed through as the base image.  Files are compared with the CPU reference's: the base JPEG is the caller's, the gain map is the device's, whose bytes equal
    the reference's -- whole files byte for byte."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth
    from oracle import loader as L

    w, h = 1280, 720
    with tempfile.TemporaryDirectory() as d:
        hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
        sdr = synth.make_sdr_yuv420(w, h)
        np.concatenate([hdr.valid(0).ravel(), hdr.valid(1).ravel()]).tofile(os.path.join(d, "in.p010"))
        np.concatenate([sdr.valid(c).ravel() for c in range(3)]).tofile(os.path.join(d, "in.yuv420"))
        # the compressed SDR intent: the SDR rendition as a BT.601 JPEG (what an application's camera pipeline would hand over)
        if L.ref() is None:
            pytest.skip("oracle/_ref not built")
        open(os.path.join(d, "sdr.jpg"), "wb").write(L.ref_jpeg_compress(L.convert_yuv("ref", sdr, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3), 95))
        args = ["-m", 0, "-p", "in.p010", "-i", "sdr.jpg", "-w", w, "-h", h, "-a", 0, "-C", 2, "-c", 0, "-t", 1, "-R", 0]
        if api == 2:
            args += ["-y", "in.yuv420", "-b", 1]
        rc, _, err, _ = F.run_app(args + ["-z", "cpu.jpg"], False, d)
        assert rc == 0, err
        rc, _, err, trace = F.run_app(args + ["-z", "gpu.jpg"], True, d)
        assert rc == 0, err
        st = _stages(trace)
        assert "generate_gainmap" in st and trace.n("jpeg_encode_scan") == 1, trace  # the map's compressImage; the base image is the caller's JPEG
        if api == 3:
            assert trace.n("jpeg_decode_scan") == 1, trace  # the compressed SDR intent, decoded on the device
        a, b = F.read(os.path.join(d, "cpu.jpg")), F.read(os.path.join(d, "gpu.jpg"))
        assert a.size == b.size and np.array_equal(a, b), f"{int((a != b).sum()) if a.size == b.size else (a.size, b.size)} differing bytes"


@pytest.mark.parametrize("multi", [False, True])
def test_encode_with_device_entropy_coding_decodes_to_the_same_pixels(multi):
    """UHDR_HIP_SEAM_RESTART_INTERVAL=max (opt-in): the device's Huffman pass with restart markers,
    one restart interval per wavefront.  The file differs from the reference's by the DRI segments and RSTn markers only:
    decoded by the CPU reference (no acceleration) it gives exactly the pixels of the CPU-encoded file, and the accelerated
    decoder -- whose entropy stage then takes the one-lane-per-interval kernel -- agrees."""
    with tempfile.TemporaryDirectory() as d:
        p, y = _fixture(d)
        extra = ("-M", 1, "-s", 1) if multi else ()  # 3-channel map at full resolution: the packed-RGB route
        rc, _, err, _ = F.encode_api1(p, y, 1280, 720, "cpu.jpg", False, d, extra=extra)
        assert rc == 0, err
        rc, _, err, trace = F.encode_api1(p, y, 1280, 720, "dev.jpg", True, d, extra=extra, env_extra={"UHDR_HIP_SEAM_RESTART_INTERVAL": "max"})
        assert rc == 0, err
        assert trace.n("jpeg_encode_scan") == 2 and "fdct_planes" not in _stages(trace), trace  # base image and gain map
        a, b = F.read(os.path.join(d, "cpu.jpg")), F.read(os.path.join(d, "dev.jpg"))
        assert b.size > a.size and bytes(b[:2]) == b"\xff\xd8"
        assert bytes([0xff, 0xdd]) in b.tobytes() and bytes([0xff, 0xdd]) not in a.tobytes()[:2000]
        for name in ("cpu", "dev"):
            rc, _, err, _ = F.decode(name + ".jpg", 0, 4, name + ".raw", False, d)
            assert rc == 0, err
        pa, pb = F.read(os.path.join(d, "cpu.raw")), F.read(os.path.join(d, "dev.raw"))
        assert pa.size == pb.size == 1280 * 720 * 8
        assert np.array_equal(pa, pb), f"{int((pa != pb).sum())} differing bytes"
        rc, _, err, trace = F.decode("dev.jpg", 0, 4, "dev_gpu.raw", True, d)
        assert rc == 0, err
        assert "jpeg_decode_scan" in _stages(trace), trace
        assert np.array_equal(F.read(os.path.join(d, "dev_gpu.raw")), pa)


@pytest.mark.parametrize("w,h,extra,what", [
    (1920, 1080, (), "1080p: 960 x 540 chroma planes, a dummy luma block row"),
    (3840, 2160, ("-M", 0, "-s", 4), "4K with the Android-style scale-4 Y400 map: 960 x 540"),
    (1920, 1080, ("-M", 1, "-s", 1), "1080p, three-channel full-resolution map (packed RGB route)"),
    (1000, 562, ("-M", 1, "-s", 2), "odd everything: 500 x 281 RGB map, 500 x 281 chroma"),
])
def test_partial_block_encodes_run_on_the_device(w, h, extra, what):
    """Round 4: planes that are not whole 8 x 8 blocks no longer fall back to libjpeg -- the FDCT pads edge blocks on the device
    by JpegEncoderHelper::compressYCbCr's own rules (jpegencoderhelper.cpp:246-309) / libjpeg's edge replication for packed
    RGB, the entropy coder adds the dummy blocks of edge MCUs.  Both compressImage calls of the encode must show up as
    device stages and the file must be the CPU reference's byte for byte."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth

    with tempfile.TemporaryDirectory() as d:
        hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
        sdr = synth.make_sdr_yuv420(w, h)
        np.concatenate([hdr.valid(0).ravel(), hdr.valid(1).ravel()]).tofile(os.path.join(d, "in.p010"))
        np.concatenate([sdr.valid(c).ravel() for c in range(3)]).tofile(os.path.join(d, "in.yuv420"))
        rc, _, err, _ = F.encode_api1("in.p010", "in.yuv420", w, h, "cpu.jpg", False, d, extra=extra)
        assert rc == 0, err
        rc, _, err, trace = F.encode_api1("in.p010", "in.yuv420", w, h, "gpu.jpg", True, d, extra=extra)
        assert rc == 0, err
        assert trace.n("jpeg_encode_scan") == 2, (what, trace)
        assert trace.n("jpeg_encode_scan", "reference") == 0, (what, trace)
        a, b = F.read(os.path.join(d, "cpu.jpg")), F.read(os.path.join(d, "gpu.jpg"))
        assert a.size == b.size and np.array_equal(a, b), (what, a.size, b.size)


@pytest.mark.parametrize("w,h", [(1000, 562), (642, 362)])
def test_sizes_with_partial_blocks_through_the_facade(w, h):
    """Dimensions that are not multiples of the 16 x 16 MCU: the compress seam pads edge blocks on the device (round 4), the
    decode seam takes them too (dummy blocks are dropped on the device, only the visible samples come
    back).  Files and decoded frames equal the CPU reference's byte for byte."""
    from libultrahdr_amd import capi as A
    from libultrahdr_amd import synth

    with tempfile.TemporaryDirectory() as d:
        hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
        sdr = synth.make_sdr_yuv420(w, h)
        np.concatenate([hdr.valid(0).ravel(), hdr.valid(1).ravel()]).tofile(os.path.join(d, "in.p010"))
        np.concatenate([sdr.valid(c).ravel() for c in range(3)]).tofile(os.path.join(d, "in.yuv420"))
        rc, _, err, _ = F.encode_api1("in.p010", "in.yuv420", w, h, "cpu.jpg", False, d)
        assert rc == 0, err
        rc, _, err, trace = F.encode_api1("in.p010", "in.yuv420", w, h, "gpu.jpg", True, d)
        assert rc == 0, err
        assert "generate_gainmap" in _stages(trace), trace
        assert trace.n("jpeg_encode_scan") == 2, trace
        a, b = F.read(os.path.join(d, "cpu.jpg")), F.read(os.path.join(d, "gpu.jpg"))
        assert np.array_equal(a, b), f"{int((a != b).sum()) if a.size == b.size else 'size'} differing bytes"
        rc, _, err, _ = F.decode("cpu.jpg", 0, 4, "cpu.raw", False, d)
        assert rc == 0, err
        rc, _, err, trace = F.decode("cpu.jpg", 0, 4, "gpu.raw", True, d)
        assert rc == 0, err
        assert "apply_gainmap" in _stages(trace) and "jpeg_decode_scan" in _stages(trace), trace
        pa, pb = F.read(os.path.join(d, "cpu.raw")), F.read(os.path.join(d, "gpu.raw"))
        assert pa.size == pb.size == w * h * 8
        assert np.array_equal(pa, pb), f"{int((pa != pb).sum())} differing bytes"


@pytest.mark.parametrize("w,h", [(1280, 720), (1000, 562)])
def test_gainmap_image_is_downloaded_only_when_asked_for(w, h):
    """uhdr_decode leaves both decoded images on the device and defers copy_raw_image(&gainmap, gainmap_img) (jpegr.cpp:1490);
    uhdr_get_decoded_gainmap_image (ultrahdr_api.cpp:2032-2043) downloads the image when -- and only when -- it is called.  What
    it hands out, and the decoded frame, are the CPU reference's bytes: RGBA and single-channel maps, across a uhdr_reset_decoder,
    for SDR output (no applyGainMap at all), and with an effect queued (apply_effects reads the image on the host: no deferral)."""
    import subprocess
    import sys

    import json

    env = dict(os.environ)
    env.pop("UHDR_HIP_SEAM_TRACE", None)
    env["PYTHONPATH"] = F.ROOT + os.pathsep + env.get("PYTHONPATH", "")
    probe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "facade_lazy_probe.py")
    p = subprocess.run([sys.executable, probe, str(w), str(h)], cwd=F.ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    results = [l for l in p.stdout.splitlines() if l.startswith("===")]
    assert len(results) == 10 and all("MATCH" in l for l in results), p.stdout
    # the probe prints the library's stage table (uhdr_hip_seam_stats) of every section as one "--- name {json}" line
    sections = {}
    for l in p.stdout.splitlines():
        if l.startswith("--- "):
            name, _, js = l[4:].partition(" {")
            sections[name] = F.Stages(json.loads("{" + js))
    for name in ("rgb_map", "luma_map_scale4"):
        asked, not_asked, reset, sdr_out, effects = (sections[f"{name} {k}"] for k in ("asked", "not_asked", "reset", "sdr_out", "effects"))
        for sec in (asked, not_asked, reset):
            assert sec.n("jpeg_decode_scan") == 2 * (2 if sec is reset else 1), sec
            assert sec.n("apply_gainmap") >= 1, sec
        assert asked.n("gainmap_copy_deferred") == 1 and asked.n("gainmap_image_asked_for") == 2, asked
        assert not_asked.n("gainmap_copy_deferred") == 1 and not_asked.n("gainmap_image_asked_for") == 0, not_asked
        assert reset.n("gainmap_copy_deferred") == 2, reset
        assert sdr_out.n("gainmap_copy_deferred") == 1 and sdr_out.n("apply_gainmap") == 0 and sdr_out.n("apply_gainmap", "reference") == 0, sdr_out
        assert effects.n("gainmap_copy_deferred") == 0, effects
