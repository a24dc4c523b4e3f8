"""GPU: uhdr_hip_jpeg_decode_scan -- JpegDecoderHelper::decompressImage (jpegdecoderhelper.cpp:169-535) for baseline files
with every stage on the device (entropy decode, dequantization, JDCT_ISLOW IDCT, ycc_rgb_convert).  Checked bit for bit
against (a) the oracle's IDCT / colour conversion of the coefficients that went into the file, for every sampling layout,
with and without restart markers, sizes with partial blocks, and (b) the real reference's decompressImage on files the
real reference wrote (planar and RGB modes)."""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from oracle import loader as L

from test_gpu_parity import _random_coefs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def uhdr(hip_ctx):
    from libultrahdr_amd.ultrahdr import UltraHdr

    return UltraHdr(ctx=hip_ctx)


S420, S444, S422, GRAY = [(2, 2), (1, 1), (1, 1)], [(1, 1)] * 3, [(2, 1), (1, 1), (1, 1)], [(1, 1)]
CASES = [(640, 480, S420, 0), (333, 211, S420, 0), (500, 300, S444, 0), (401, 203, S422, 0), (1000, 400, GRAY, 0),
         (640, 480, S420, 4), (333, 211, S444, 7), (1000, 400, GRAY, 50), (64, 48, S420, 0), (1920, 1080, S420, 0)]


def _file(rng, w, h, sampling, ri, quality=90, kind="sparse"):
    coefs = _random_coefs(rng, w, h, sampling, kind)
    scan = L.huffman_encode_port(coefs, w, h, sampling, ri)
    ql, qc = L.quant_table_port(quality, False), L.quant_table_port(quality, True)
    return coefs, (ql, qc), L.jpeg_assemble_port(coefs, w, h, sampling, ri, ql, qc, scan)


@pytest.mark.parametrize("w,h,sampling,ri", CASES)
def test_planes_equal_the_oracles_idct_of_the_coefficients(uhdr, w, h, sampling, ri):
    rng = np.random.default_rng(1000 + w + ri)
    coefs, (ql, qc), jpeg = _file(rng, w, h, sampling, ri)
    got = uhdr.jpeg_decode(jpeg)
    assert len(got) == len(coefs)
    for c in range(len(coefs)):
        want = L.idct_dequant_port(coefs[c], ql if c == 0 else qc)
        assert got[c].shape == want.shape, c
        assert np.array_equal(got[c], want), (c, int((got[c] != want).sum()))


@pytest.mark.parametrize("channels,variant", [(3, 0), (3, 1), (4, 0), (4, 1)])
def test_rgb_output_equals_the_oracles_colour_conversion(uhdr, channels, variant):
    rng = np.random.default_rng(77 + channels * 2 + variant)
    for (w, h, ri) in ((500, 300, 0), (333, 211, 7), (1280, 720, 0)):
        coefs, (ql, qc), jpeg = _file(rng, w, h, S444, ri, quality=95)
        planes = [L.idct_dequant_port(coefs[c], ql if c == 0 else qc)[:h, :w] for c in range(3)]
        want = L.jpeg_ycc_to_rgb_port(*planes, out_bpp=channels, variant=variant).reshape(h, w, channels)
        got = uhdr.jpeg_decode(jpeg, channels, variant)
        assert np.array_equal(got, want), (w, h, ri, int((got != want).sum()))


@pytest.mark.parametrize("ri", [0, 4])
def test_the_end_of_the_scan_is_found_on_the_device_or_by_the_walk(uhdr, ri):
    """Round 5: uhdr_hip_jpeg_decode_scan takes the EOI marker from the buffer's last two bytes and lets the device report any
    OTHER marker inside what it took for entropy-coded data (a pinned status word); only then does it walk the scan on the host
    (T.81 B.1.1.2) and decode the prefix.  All four situations give the planes of the plain file: (a) the file as it is (EOI last:
    the guess holds); (b) bytes behind EOI that do not end in FF D9 (no guess: the walk); (c) bytes behind EOI that DO end in FF D9
    (the guess is wrong, the device sees the real EOI as a stray marker, the walk cuts there); (d) UHDR_HIP_JPEG_WALK semantics are
    those of (b): the same code path as rounds 1-4."""
    rng = np.random.default_rng(4242 + ri)
    w, h = 640, 480
    coefs, (ql, qc), jpeg = _file(rng, w, h, S420, ri)
    want = uhdr.jpeg_decode(jpeg)
    st0 = A.Stats()
    uhdr.lib.uhdr_hip_get_stats(uhdr.ctx.handle, C.byref(st0))
    for tail in (b"\x00\x11\x22\x33", b"\x12\x34\x56\xff\xd9", b"\xff\xd9", b"\xff\xe1\x00\x04ab\xff\xd9"):
        got = uhdr.jpeg_decode(jpeg + tail)
        for c in range(3):
            assert np.array_equal(got[c], want[c]), (tail, c)
    st1 = A.Stats()
    uhdr.lib.uhdr_hip_get_stats(uhdr.ctx.handle, C.byref(st1))
    assert st1.entropy_decode_declined == st0.entropy_decode_declined  # nothing was handed back to the caller on the way


def test_rgb_output_of_a_subsampled_file_is_refused(uhdr):
    rng = np.random.default_rng(5)
    _, _, jpeg = _file(rng, 640, 480, S420, 0)
    with pytest.raises(A.UhdrError) as e:
        uhdr.jpeg_decode(jpeg, 3)
    assert e.value.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE


def test_truncated_data_is_reported_and_the_context_stays_usable(uhdr):
    rng = np.random.default_rng(6)
    coefs, (ql, qc), jpeg = _file(rng, 640, 480, S420, 0)
    hdr = uhdr.jpeg_parse(jpeg)
    cut = jpeg[: hdr.scan_offset + hdr.scan_bytes // 2] + b"\xff\xd9"
    with pytest.raises(A.UhdrError) as e:
        uhdr.jpeg_decode(cut)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM
    got = uhdr.jpeg_decode(jpeg)
    assert np.array_equal(got[0], L.idct_dequant_port(coefs[0], ql))


def _ref_decode(jpeg, mode):
    ref = L.ref()
    buf = np.frombuffer(jpeg, dtype=np.uint8)
    dst = A.RawImage()
    store = np.zeros(1 << 26, dtype=np.uint8)
    assert ref.ref_jpeg_decompress(buf.ctypes.data, buf.size, mode, C.byref(dst), store.ctypes.data, store.size) == 0
    return dst, store


@pytest.mark.parametrize("w,h", [(1280, 720), (3840, 2160), (1366, 768)])
def test_reference_written_base_image_decodes_like_the_reference(uhdr, w, h):
    """4:2:0 q95 file from JpegEncoderHelper::compressImage; planes vs JpegDecoderHelper::decompressImage(DECODE_TO_YCBCR_CS)."""
    if L.ref() is None:
        pytest.skip("oracle/_ref not built")
    img = synth.make_sdr_yuv420(w, h, align=8)
    jpeg = L.ref_jpeg_compress(img, 95)
    dst, store = _ref_decode(jpeg, 0)
    assert dst.fmt == A.UHDR_IMG_FMT_12bppYCbCr420
    got = uhdr.jpeg_decode(jpeg)
    off = 0
    for c in range(3):  # oracle/ref_shim.cpp packs the planes back to back: stride x h rows of luma, stride x h/2 of each chroma
        pw, ph = (w, h) if c == 0 else (w // 2, h // 2)
        stride = dst.stride[c]
        want = store[off: off + stride * ph].reshape(ph, stride)[:, :pw]
        off += stride * ph
        assert np.array_equal(got[c][:ph, :pw], want), c


def test_reference_written_three_channel_map_decodes_like_the_reference(uhdr):
    """3-channel gain map (JCS_RGB in, 4:4:4): RGB output vs JpegDecoderHelper::decompressImage(DECODE_STREAM)."""
    if L.ref() is None:
        pytest.skip("oracle/_ref not built")
    w, h = 1920, 1080
    gm = synth.make_gainmap(w, h, 3)
    jpeg = L.ref_jpeg_compress(gm, 95)
    dst, store = _ref_decode(jpeg, 1)
    bpp = 4 if dst.fmt == A.UHDR_IMG_FMT_32bppRGBA8888 else 3
    want = store[: dst.stride[0] * bpp * h].reshape(h, dst.stride[0] * bpp)[:, : w * bpp].reshape(h, w, bpp)
    got = uhdr.jpeg_decode(jpeg, bpp, 1)  # oracle/_ref links the image's IJG libjpeg 9: the refined green constants
    assert np.array_equal(got, want), int((got != want).sum())


# ---- the encode-side mirror: uhdr_hip_jpeg_encode_scan ----------------------------------------------------------------------
def _planes(rng, w, h, sampling):
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    out = []
    for (hs, vs) in sampling:
        pw, ph = -(-w * hs // hmax), -(-h * vs // vmax)
        pw, ph = -(-pw // 8) * 8, -(-ph // 8) * 8
        yy, xx = np.mgrid[0:ph, 0:pw]
        base = 128 + 90 * np.sin(xx / 37.0) * np.cos(yy / 23.0) + rng.normal(0, 6, (ph, pw))
        out.append(np.clip(base, 0, 255).astype(np.uint8))
    return out


@pytest.mark.parametrize("w,h,sampling", [(640, 480, S420), (1920, 1080, S420), (512, 256, S444), (400, 200, S422), (1000, 400, GRAY), (64, 64, S420)])
def test_device_encoded_file_holds_the_oracles_fdct_coefficients(uhdr, w, h, sampling):
    """samples -> FDCT + quantize + Huffman on the device -> file; libjpeg-style decode of that file (the oracle's entropy
    decoder) gives the oracle's FDCT coefficients of the same samples, and the device decoder inverts it."""
    rng = np.random.default_rng(w * 3 + h)
    planes = _planes(rng, w, h, sampling)
    ql, qc = L.quant_table_port(95, False), L.quant_table_port(95, True)
    jpeg = uhdr.jpeg_encode(planes, w, h, sampling, ql, qc)
    hdr = uhdr.jpeg_parse(jpeg)
    bpm = sum(a * b for a, b in sampling) if len(sampling) > 1 else 1
    assert hdr.scan.restart_interval == 64 // bpm
    want = [L.fdct_quant_port(p, p.shape[1], p.shape[1] // 8, p.shape[0] // 8, ql if c == 0 else qc) for c, p in enumerate(planes)]
    # real component grids (the planes may be padded beyond them up to the MCU grid)
    shapes = [(hdr.scan.blocks_h[c], hdr.scan.blocks_w[c]) for c in range(len(planes))]
    rc, got = L.huffman_decode_port(shapes, w, h, sampling, hdr.scan.restart_interval, jpeg[hdr.scan_offset: hdr.scan_offset + hdr.scan_bytes])
    assert rc == 0
    for c in range(len(planes)):
        bh, bw = shapes[c]
        assert np.array_equal(got[c], want[c][:bh, :bw]), c
    back = uhdr.jpeg_decode(jpeg)
    for c in range(len(planes)):
        bh, bw = shapes[c]
        assert np.array_equal(back[c], L.idct_dequant_port(want[c][:bh, :bw], ql if c == 0 else qc)), c


def test_device_encoded_rgb_map_holds_the_oracles_coefficients(uhdr):
    rng = np.random.default_rng(31)
    w, h = 640, 360
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([np.clip(128 + 80 * np.sin(xx / (30.0 + 7 * k)) * np.cos(yy / 19.0) + rng.normal(0, 5, (h, w)), 0, 255) for k in range(3)], axis=-1).astype(np.uint8)
    ql, qc = L.quant_table_port(95, False), L.quant_table_port(95, True)
    jpeg = uhdr.jpeg_encode(rgb, w, h, S444, ql, qc, rgb_channels=3)
    hdr = uhdr.jpeg_parse(jpeg)
    assert hdr.scan.restart_interval == 21
    ycc = L.jpeg_rgb_to_ycc_port(rgb.reshape(h, w * 3), w, w, h)
    want = [L.fdct_quant_port(ycc[c], w, w // 8, h // 8, ql if c == 0 else qc) for c in range(3)]
    rc, got = L.huffman_decode_port([(h // 8, w // 8)] * 3, w, h, S444, 21, jpeg[hdr.scan_offset: hdr.scan_offset + hdr.scan_bytes])
    assert rc == 0
    for c in range(3):
        assert np.array_equal(got[c], want[c]), c


# ---- partial edge blocks on the device: uhdr_hip_jpeg_encode_image against the REAL reference's files ---------------------------
def _ref_scan_bytes(uhdr, img, quality):
    jpeg = L.ref_jpeg_compress(img, quality)
    hdr = uhdr.jpeg_parse(jpeg)
    return jpeg[hdr.scan_offset: hdr.scan_offset + hdr.scan_bytes], hdr


@pytest.mark.parametrize("w,h", [(1920, 1080), (1000, 562), (333, 217), (24, 10), (8, 8), (17, 9), (640, 476)])
@pytest.mark.parametrize("wide_stride", [True, False])
def test_partial_edge_blocks_420_equal_the_reference_file(uhdr, ref, w, h, wide_stride):
    """JpegEncoderHelper::compressYCbCr's edge rules (jpegencoderhelper.cpp:246-309) on the device: a stride that covers the
    block-aligned width hands libjpeg the caller's own stride bytes and constant rows below the plane; a shorter one goes
    through the helper's scratch MCU rows (constant tail, STALE rows below).  Entropy-coded bytes == the reference's."""
    from libultrahdr_amd.images import Image

    if w % 2 or h % 2:
        pytest.skip("4:2:0 needs even dimensions in the reference's API")
    rng = np.random.default_rng(w * 7 + h + wide_stride)
    align = 64 if wide_stride else 1
    img = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, align=align)
    for i in range(3):
        st = img.plane(i)
        st[...] = rng.integers(0, 256, st.shape, dtype=np.uint8)  # the stride bytes are part of the input: fill them too
    want, hdr = _ref_scan_bytes(uhdr, img, 90)
    planes = [img.plane(i) for i in range(3)]
    ql, qc = hdr_qt(hdr)
    got = uhdr.jpeg_encode_image(planes, w, h, S420, ql, qc)
    assert got == want, (len(got), len(want))


@pytest.mark.parametrize("w,h", [(960, 540), (241, 135), (13, 7), (1000, 562)])
@pytest.mark.parametrize("wide_stride", [True, False])
def test_partial_edge_blocks_y400_map_equal_the_reference_file(uhdr, ref, w, h, wide_stride):
    from libultrahdr_amd.images import Image

    rng = np.random.default_rng(w + h * 5 + wide_stride)
    img = Image(A.UHDR_IMG_FMT_8bppYCbCr400, w, h, align=64 if wide_stride else 1)
    st = img.plane(0)
    st[...] = rng.integers(0, 256, st.shape, dtype=np.uint8)
    want, hdr = _ref_scan_bytes(uhdr, img, 85)
    ql, qc = hdr_qt(hdr)
    got = uhdr.jpeg_encode_image([img.plane(0)], w, h, GRAY, ql, qc)
    assert got == want, (len(got), len(want))


@pytest.mark.parametrize("w,h", [(333, 217), (1000, 562), (8, 8), (21, 3)])
def test_partial_edge_blocks_rgb_map_equal_the_reference_file(uhdr, ref, w, h):
    """A packed RGB gain map goes through jpeg_write_scanlines in the reference: libjpeg replicates the last column / row."""
    from libultrahdr_amd.images import Image

    rng = np.random.default_rng(w * 11 + h)
    img = Image(A.UHDR_IMG_FMT_24bppRGB888, w, h, align=16)
    st = img.plane(0)
    st[...] = rng.integers(0, 256, st.shape, dtype=np.uint8)
    want, hdr = _ref_scan_bytes(uhdr, img, 92)
    ql, qc = hdr_qt(hdr)
    got = uhdr.jpeg_encode_image(img.plane(0).reshape(h, img.raw.stride[0], 3), w, h, S444, ql, qc, rgb_channels=3)
    assert got == want, (len(got), len(want))


def hdr_qt(hdr):
    return np.array(hdr.qtable[0][:], dtype=np.uint16), np.array(hdr.qtable[1][:] if hdr.scan.num_components == 3 else hdr.qtable[0][:], dtype=np.uint16)


def test_damaged_files_never_crash_the_device_path(uhdr):
    """Random damage inside the entropy-coded data of a good file: the decode either succeeds (Huffman streams rarely hold
    undefined codes: the result is then simply what the bits say) or reports UHDR_CODEC_INVALID_PARAM /
    UNSUPPORTED_FEATURE; it never faults or hangs, and the undamaged file decodes correctly afterwards."""
    rng = np.random.default_rng(99)
    coefs, (ql, qc), jpeg = _file(rng, 800, 480, S420, 0)
    hdr = uhdr.jpeg_parse(jpeg)
    lo, hi = hdr.scan_offset + 16, hdr.scan_offset + hdr.scan_bytes - 16
    for trial in range(12):
        bad = bytearray(jpeg)
        for i in rng.integers(lo, hi, 1 + 40 * (trial % 4)):
            if bad[i] != 0xFF and bad[i - 1] != 0xFF and bad[i + 1] != 0xFF:
                bad[i] ^= int(rng.integers(1, 255))
                if bad[i] == 0xFF:
                    bad[i] = 0x7F
        try:
            uhdr.jpeg_decode(bytes(bad))
        except A.UhdrError as err:
            assert err.code in (A.UHDR_CODEC_INVALID_PARAM, A.UHDR_CODEC_UNSUPPORTED_FEATURE), err
    got = uhdr.jpeg_decode(jpeg)
    for c in range(3):
        assert np.array_equal(got[c], L.idct_dequant_port(coefs[c], ql if c == 0 else qc)), c


@pytest.mark.parametrize("subsampling,quality", [(2, 90), (0, 85), (1, 95), (2, 100)])
def test_files_with_optimised_huffman_tables(uhdr, subsampling, quality):
    """Files written by another encoder (Pillow's libjpeg-turbo) with optimize=True: the DHT segments hold per-image tables,
    not Annex K.  The device decode (file tables -> two-level / value-form tables) against the oracle's entropy decoder and
    IDCT on the same file."""
    PILImage = pytest.importorskip("PIL.Image")
    import io

    rng = np.random.default_rng(400 + subsampling + quality)
    w, h = 1024, 640
    yy, xx = np.mgrid[0:h, 0:w]
    rgb = np.stack([np.clip(128 + 90 * np.sin(xx / (23.0 + 9 * k)) * np.cos(yy / (17.0 + 3 * k)) + rng.normal(0, 10, (h, w)), 0, 255) for k in range(3)],
                   axis=-1).astype(np.uint8)
    buf = io.BytesIO()
    PILImage.fromarray(rgb, "RGB").save(buf, "JPEG", quality=quality, optimize=True, subsampling=subsampling)
    jpeg = buf.getvalue()
    hdr = uhdr.jpeg_parse(jpeg)
    sc = hdr.scan
    assert sc.num_components == 3 and sc.restart_interval == 0
    sampling = [(sc.h_samp[c], sc.v_samp[c]) for c in range(3)]
    bits = np.frombuffer(hdr.tables.bits, dtype=np.uint8).reshape(4, 17)
    vals = np.frombuffer(hdr.tables.vals, dtype=np.uint8).reshape(4, 256)
    std_bits, _ = L.std_dht_tables()
    assert not np.array_equal(bits, std_bits), "the encoder was asked for optimised tables"
    shapes = [(sc.blocks_h[c], sc.blocks_w[c]) for c in range(3)]
    rc, coefs = L.huffman_decode_port(shapes, w, h, sampling, 0, jpeg[hdr.scan_offset: hdr.scan_offset + hdr.scan_bytes], tables=(bits, vals))
    assert rc == 0
    import ctypes as C_

    def stats():
        st = A.Stats()
        uhdr.lib.uhdr_hip_get_stats(uhdr.ctx.handle, C_.byref(st))
        return st

    before = stats()
    try:
        got = uhdr.jpeg_decode(jpeg)
        after = stats()
        assert after.entropy_decode_parallel == before.entropy_decode_parallel + 1 and after.entropy_decode_declined == before.entropy_decode_declined
    except A.UhdrError as err:
        # the decline is visible through the C ABI (uhdr_hip_get_stats), not only in the facade's trace
        assert stats().entropy_decode_declined == before.entropy_decode_declined + 1
        # quality 100 on a noisy image: nearly every block runs to coefficient 63 without an EOB, decoders started in different
        # places fall in step only after tens of kilobits, and the parallel schemes give the stream back (the facade then takes
        # libjpeg's decoder).  The plain entry point still decodes it -- on one lane -- to the same coefficients.
        assert quality == 100 and err.code == A.UHDR_CODEC_UNSUPPORTED_FEATURE, err
        _, dev = uhdr.jpeg_to_coefficients(jpeg)
        for c in range(3):
            assert np.array_equal(dev[c].cpu().numpy(), coefs[c]), c
        return
    for c in range(3):
        qt = np.frombuffer(hdr.qtable, dtype=np.uint16).reshape(3, 64)[c]
        assert np.array_equal(got[c], L.idct_dequant_port(coefs[c], qt)), c


def test_decoded_images_stay_on_the_device_for_the_apply_stage(uhdr):
    """uhdr_hip_resident_begin / _end: decodeJPEGR's handoff (jpegr.cpp:1478-1530).  The host variant of applyGainMap, handed the
    very buffers uhdr_hip_jpeg_decode_scan filled, reads the device copies (stats.resident_hits) and produces the bytes of the
    upload path; outside a session, or after the session is reopened, the host planes are what counts."""
    from libultrahdr_amd.images import Image

    w, h = 656, 352  # 656: the device copies have a 64-aligned pitch (704), the host planes do not
    rng = np.random.default_rng(4242)
    _, _, base_jpeg = _file(rng, w, h, S420, 0)
    _, _, map_jpeg = _file(rng, w, h, S444, 0)
    base = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
    gmap = [np.zeros((h, w, 4), np.uint8)]

    def raw(fmt, planes, strides, cg):
        r = A.RawImage()
        r.fmt, r.cg, r.ct, r.range, r.w, r.h = fmt, cg, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, w, h
        for i, p in enumerate(planes):
            r.planes[i], r.stride[i] = p.ctypes.data, strides[i]

        class _Host:
            device = None
        o = _Host()
        o.raw = r
        return o

    sdr_img = raw(A.UHDR_IMG_FMT_12bppYCbCr420, base, [w, w // 2, w // 2], A.UHDR_CG_DISPLAY_P3)
    gm_img = raw(A.UHDR_IMG_FMT_32bppRGBA8888, gmap, [w], A.UHDR_CG_UNSPECIFIED)
    md = synth.default_metadata()

    def stats():
        st = A.Stats()
        uhdr.lib.uhdr_hip_get_stats(uhdr.ctx.handle, C.byref(st))
        return st.resident_hits

    def apply():
        dest = Image(A.UHDR_IMG_FMT_64bppRGBAHalfFloat, w, h)
        uhdr.applyGainMap(sdr_img, gm_img, md, A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat, 4.0, dest)
        return dest.buf.copy()

    def decode_both():
        uhdr.jpeg_decode(base_jpeg, outs=base)
        uhdr.jpeg_decode(map_jpeg, 4, outs=gmap)

    h0 = stats()
    decode_both()
    want = apply()  # no session: both images uploaded
    assert stats() == h0 and want.any()
    uhdr.lib.uhdr_hip_resident_begin(uhdr.ctx.handle)
    try:
        decode_both()
        got = apply()
        assert stats() == h0 + 2, "base and gain map were found on the device"
        assert np.array_equal(got, want)
        # the session is reopened (what the facade does when a stage falls back to the reference's CPU code, which may write in
        # place): the copies are dropped, the host planes count again
        uhdr.lib.uhdr_hip_resident_begin(uhdr.ctx.handle)
        base[0][:] = 255 - base[0]
        changed = apply()
        assert stats() == h0 + 2 and not np.array_equal(changed, want)
        base[0][:] = 255 - base[0]
        # a second file decoded into the same base buffers: the newer copy counts, the older one is dropped
        _, _, base_jpeg2 = _file(rng, w, h, S420, 0)
        uhdr.jpeg_decode(base_jpeg, outs=base)
        uhdr.jpeg_decode(base_jpeg2, outs=base)
        uhdr.jpeg_decode(map_jpeg, 4, outs=gmap)
        newer = apply()
        assert stats() == h0 + 4
        uhdr.lib.uhdr_hip_resident_begin(uhdr.ctx.handle)
        assert np.array_equal(newer, apply()) and not np.array_equal(newer, want)
        # an in-place operator on a resident image (convertYuv) works on the device copy and brings the host planes along
        uhdr.jpeg_decode(base_jpeg, outs=base)
        uhdr.convertYuv(sdr_img, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3)
        assert stats() == h0 + 5
        converted = [p.copy() for p in base]
        uhdr.lib.uhdr_hip_resident_begin(uhdr.ctx.handle)
        uhdr.jpeg_decode(base_jpeg, outs=base)
        uhdr.lib.uhdr_hip_resident_begin(uhdr.ctx.handle)  # dropped: the upload route
        uhdr.convertYuv(sdr_img, A.UHDR_CG_BT_709, A.UHDR_CG_DISPLAY_P3)
        assert stats() == h0 + 5
        for a, b in zip(converted, base):
            assert np.array_equal(a, b)
        uhdr.jpeg_decode(base_jpeg, outs=base)
    finally:
        uhdr.lib.uhdr_hip_resident_end(uhdr.ctx.handle)
    assert np.array_equal(apply(), want) and stats() == h0 + 5  # after _end nothing is kept


def test_lazy_downloads_of_decoded_images(uhdr):
    """uhdr_hip_resident_lazy: inside a session the decoded images are not written to the caller's planes; applyGainMap reads the
    device copies; _flush writes the planes; _adopt + _materialize stand for copy_raw_image of the gain map (jpegr.cpp:1490), also
    after _end; the library writes back by itself before a slot is reused or a host plane would be uploaded."""
    from libultrahdr_amd.images import Image

    w, h = 656, 352
    rng = np.random.default_rng(777)
    _, _, base_jpeg = _file(rng, w, h, S420, 0)
    _, _, map_jpeg = _file(rng, w, h, S444, 0)
    _, _, luma_jpeg = _file(rng, w, h, GRAY, 0)
    lib, ctx = uhdr.lib, uhdr.ctx.handle

    def raw(fmt, planes, strides):
        r = A.RawImage()
        r.fmt, r.cg, r.ct, r.range, r.w, r.h = fmt, A.UHDR_CG_DISPLAY_P3, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, w, h
        for i, p in enumerate(planes):
            r.planes[i], r.stride[i] = p.ctypes.data, strides[i]

        class _Host:
            device = None
        o = _Host()
        o.raw = r
        return o

    def stats():
        st = A.Stats()
        lib.uhdr_hip_get_stats(ctx, C.byref(st))
        return st

    def ok(st):
        assert st.error_code == 0, st.detail

    md = synth.default_metadata()

    def apply(sdr_img, gm_img):
        dest = Image(A.UHDR_IMG_FMT_64bppRGBAHalfFloat, w, h)
        uhdr.applyGainMap(sdr_img, gm_img, md, A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat, 4.0, dest)
        return dest.buf.copy()

    # eager reference
    base_e = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
    gmap_e = [np.zeros((h, w, 4), np.uint8)]
    luma_e = [np.zeros((h, w), np.uint8)]
    uhdr.jpeg_decode(base_jpeg, outs=base_e)
    uhdr.jpeg_decode(map_jpeg, 4, outs=gmap_e)
    uhdr.jpeg_decode(luma_jpeg, outs=luma_e)
    want = apply(raw(A.UHDR_IMG_FMT_12bppYCbCr420, base_e, [w, w // 2, w // 2]), raw(A.UHDR_IMG_FMT_32bppRGBA8888, gmap_e, [w]))

    base = [np.full_like(p, 7) for p in base_e]
    gmap = [np.full_like(gmap_e[0], 7)]
    sdr_img = raw(A.UHDR_IMG_FMT_12bppYCbCr420, base, [w, w // 2, w // 2])
    gm_img = raw(A.UHDR_IMG_FMT_32bppRGBA8888, gmap, [w])
    dst = np.full((h, w + 8, 4), 9, np.uint8)  # a destination with its own stride
    dst_img = raw(A.UHDR_IMG_FMT_32bppRGBA8888, [dst], [w + 8])
    s0 = stats()
    lib.uhdr_hip_resident_lazy(ctx, 1)  # outside a session: no effect
    uhdr.jpeg_decode(base_jpeg, outs=base)
    assert np.array_equal(base[0], base_e[0]) and stats().lazy_downloads_skipped == s0.lazy_downloads_skipped
    for p in base:
        p[:] = 7
    lib.uhdr_hip_resident_begin(ctx)
    try:
        lib.uhdr_hip_resident_lazy(ctx, 1)
        uhdr.jpeg_decode(base_jpeg, outs=base)
        uhdr.jpeg_decode(map_jpeg, 4, outs=gmap)
        lib.uhdr_hip_resident_lazy(ctx, 0)
        assert stats().lazy_downloads_skipped == s0.lazy_downloads_skipped + 2
        assert all((p == 7).all() for p in base) and (gmap[0] == 7).all(), "the caller's planes were left alone"
        assert lib.uhdr_hip_resident_adopt(ctx, C.byref(sdr_img.raw), C.byref(dst_img.raw)) == 0  # three planes: not adoptable
        assert lib.uhdr_hip_resident_adopt(ctx, C.byref(gm_img.raw), C.byref(dst_img.raw)) == 1
        assert lib.uhdr_hip_resident_adopt(ctx, C.byref(gm_img.raw), C.byref(dst_img.raw)) == 0  # one at a time
        got = apply(sdr_img, gm_img)
        assert np.array_equal(got, want) and stats().resident_hits == s0.resident_hits + 2
        assert (gmap[0] == 7).all() and (dst == 9).all()
    finally:
        lib.uhdr_hip_resident_end(ctx)
    assert (gmap[0] == 7).all() and (dst == 9).all(), "_end discards without writing back"
    ok(lib.uhdr_hip_resident_materialize(ctx))
    assert np.array_equal(dst[:, :w], gmap_e[0]) and (dst[:, w:] == 9).all() and (gmap[0] == 7).all()
    assert stats().lazy_downloads_done == s0.lazy_downloads_done + 1
    dst[:] = 9
    ok(lib.uhdr_hip_resident_materialize(ctx))  # nothing pending any more
    assert (dst == 9).all()

    # _flush inside the session: planes and the adopted copy, both; _forget after _end: nothing is written
    lib.uhdr_hip_resident_begin(ctx)
    try:
        lib.uhdr_hip_resident_lazy(ctx, 1)
        uhdr.jpeg_decode(base_jpeg, outs=base)
        uhdr.jpeg_decode(map_jpeg, 4, outs=gmap)
        assert lib.uhdr_hip_resident_adopt(ctx, C.byref(gm_img.raw), C.byref(dst_img.raw)) == 1
        ok(lib.uhdr_hip_resident_flush(ctx))
        for a, b in zip(base, base_e):
            assert np.array_equal(a, b)
        assert np.array_equal(gmap[0], gmap_e[0]) and np.array_equal(dst[:, :w], gmap_e[0]) and (dst[:, w:] == 9).all()
        assert np.array_equal(apply(sdr_img, gm_img), want)  # still resident
        # both images unwritten again; a third lazy decode takes the older slot (the base image's) and the library writes
        # that image back first
        for p in base + gmap:
            p[:] = 7
        dst[:] = 9
        uhdr.jpeg_decode(base_jpeg, outs=base)
        uhdr.jpeg_decode(map_jpeg, 4, outs=gmap)
        assert all((p == 7).all() for p in base + gmap)
        luma = [np.full((h, w), 7, np.uint8)]
        uhdr.jpeg_decode(luma_jpeg, outs=luma)
        for a, b in zip(base, base_e):
            assert np.array_equal(a, b)
        assert (gmap[0] == 7).all() and (luma[0] == 7).all()
        # the base image is no longer kept: applyGainMap uploads its host planes -- and before the library reads ANY host plane it
        # writes the unwritten ones
        assert np.array_equal(apply(sdr_img, gm_img), want)
        assert np.array_equal(gmap[0], gmap_e[0]) and np.array_equal(luma[0], luma_e[0])
        # the session reopened (the facade's hand-back to the reference's CPU code): planes are written first
        for p in gmap:
            p[:] = 7
        uhdr.jpeg_decode(map_jpeg, 4, outs=gmap)
        assert (gmap[0] == 7).all()
        lib.uhdr_hip_resident_begin(ctx)
        assert np.array_equal(gmap[0], gmap_e[0])
        # adopted, then forgotten before the session ends: nothing is copied afterwards
        lib.uhdr_hip_resident_lazy(ctx, 1)
        uhdr.jpeg_decode(map_jpeg, 4, outs=gmap)
        dst[:] = 9
        assert lib.uhdr_hip_resident_adopt(ctx, C.byref(gm_img.raw), C.byref(dst_img.raw)) == 1
    finally:
        lib.uhdr_hip_resident_end(ctx)
    lib.uhdr_hip_resident_forget(ctx)
    ok(lib.uhdr_hip_resident_materialize(ctx))
    assert (dst == 9).all()

    # copy_raw_image's one conversion on this path: an RGB888 image (what a build against IJG libjpeg decodes a three-channel map
    # to) adopted by an RGBA8888 destination -- alpha 255 (gainmapmath.cpp:1566-1587)
    rgb_e = uhdr.jpeg_decode(map_jpeg, 3)
    rgb = [np.full((h, w, 3), 7, np.uint8)]
    rgb_img = raw(A.UHDR_IMG_FMT_24bppRGB888, rgb, [w])
    lib.uhdr_hip_resident_begin(ctx)
    try:
        lib.uhdr_hip_resident_lazy(ctx, 1)
        uhdr.jpeg_decode(map_jpeg, 3, outs=rgb)
        assert lib.uhdr_hip_resident_adopt(ctx, C.byref(rgb_img.raw), C.byref(dst_img.raw)) == 1
    finally:
        lib.uhdr_hip_resident_end(ctx)
    ok(lib.uhdr_hip_resident_materialize(ctx))
    assert (rgb[0] == 7).all() and np.array_equal(dst[:, :w, :3], rgb_e) and (dst[:, :w, 3] == 255).all() and (dst[:, w:] == 9).all()


def test_progressive_files_are_not_for_this_path(uhdr):
    PILImage = pytest.importorskip("PIL.Image")
    import io

    buf = io.BytesIO()
    PILImage.fromarray(np.full((64, 64, 3), 77, dtype=np.uint8), "RGB").save(buf, "JPEG", progressive=True)
    with pytest.raises(ValueError):
        uhdr.jpeg_parse(buf.getvalue())
