"""GPU: baseline Huffman decoding of scans WITHOUT restart markers -- every file the reference writes -- by the
self-synchronising parallel decoder (csrc/huffman_decode_sync.hip), against the coefficients that went in / that libjpeg
reads, for every sampling layout, sizes with dummy blocks, all coefficient statistics, several subsequence sizes, and the
serial kernel it replaces."""
import ctypes as C
import os

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


def _dev(scan: bytes):
    import torch

    return torch.from_numpy(np.frombuffer(scan, dtype=np.uint8).copy()).to("cuda:0")


CASES = [(640, 480, [(2, 2), (1, 1), (1, 1)]), (333, 211, [(2, 2), (1, 1), (1, 1)]), (500, 300, [(1, 1)] * 3), (401, 203, [(2, 1), (1, 1), (1, 1)]),
         (1000, 400, [(1, 1)]), (1920, 1080, [(2, 2), (1, 1), (1, 1)])]


@pytest.mark.parametrize("kind", ["sparse", "dense", "worst", "zero"])
def test_sync_decoder_inverts_the_encoder(uhdr, kind):
    rng = np.random.default_rng(211)
    for (w, h, sampling) in CASES:
        if kind in ("dense", "worst") and w * h > 700_000:
            continue  # the oracle's encoder is the slow part
        coefs = _random_coefs(rng, w, h, sampling, kind)
        scan = L.huffman_encode_port(coefs, w, h, sampling, 0)
        if len(scan) < 4096:
            continue
        got = uhdr.huffman_decode(_dev(scan), [c.shape[:2] for c in coefs], w, h, sampling, 0)
        for c in range(len(coefs)):
            assert np.array_equal(got[c].cpu().numpy(), coefs[c]), (kind, w, h, c)


def _smooth_coefs(rng, w, h, sampling, style):
    """flat: constant blocks under a noisy band (a periodic bit pattern); smooth: a slowly varying DC with the two lowest AC
    terms now and then -- the statistics of a gain map, the content the decoder needs its longest windows for."""
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    out = []
    for hs, vs in sampling:
        cw, ch = -(-w * hs // hmax), -(-h * vs // vmax)
        bw, bh = -(-cw // 8), -(-ch // 8)
        a = np.zeros((bh, bw, 64), np.int16)
        if style == "flat":
            a[:2] = (rng.integers(-40, 41, (2, bw, 64)) * (rng.random((2, bw, 64)) < 0.1)).astype(np.int16)
            a[:2, :, 0] = rng.integers(-500, 501, (2, bw))
            a[2:, :, 0] = 77
        else:
            a[..., 0] = np.clip(np.cumsum(rng.integers(-2, 3, (bh, bw)), axis=1) + 100, -1020, 1020)
            for k in (1, 8):
                a[..., k] = rng.integers(-2, 3, (bh, bw)) * (rng.random((bh, bw)) < 0.3)
        out.append(np.ascontiguousarray(a))
    return out


@pytest.mark.parametrize("style", ["flat", "smooth"])
@pytest.mark.parametrize("sampling", [[(1, 1)] * 3, [(2, 2), (1, 1), (1, 1)], [(2, 1), (1, 1), (1, 1)], [(1, 1)]])
def test_flat_and_smooth_content_takes_the_parallel_decoder(uhdr, style, sampling):
    """Sparse streams (a few short symbols per block) are where a decoder started at the wrong bit takes longest to fall in step:
    a 4K three-channel gain map lost its true path with seven overflow levels at 512 and 1024 bits.  Up to fifteen levels where the
    slots allow (fewer than six blocks per MCU), and 1024-bit subsequences from the start below 64 bits per block: such scans are
    decoded in one attempt -- and whatever the attempt ladder does, the coefficients are the ones that were coded."""
    rng = np.random.default_rng(4100 + len(sampling) + sampling[0][0] * 7 + sampling[0][1])
    w, h = 2048, 1024
    coefs = _smooth_coefs(rng, w, h, sampling, style)
    scan = L.huffman_encode_port(coefs, w, h, sampling, 0)
    assert len(scan) >= 4096

    def stats():
        st = A.Stats()
        uhdr.lib.uhdr_hip_get_stats(uhdr.ctx.handle, C.byref(st))
        return st.entropy_decode_parallel, st.entropy_decode_declined, st.entropy_decode_single_lane

    s0 = stats()
    got = uhdr.huffman_decode(_dev(scan), [c.shape[:2] for c in coefs], w, h, sampling, 0)
    s1 = stats()
    assert (s1[0] - s0[0], s1[1] - s0[1], s1[2] - s0[2]) == (1, 0, 0)
    for c in range(len(coefs)):
        assert np.array_equal(got[c].cpu().numpy(), coefs[c]), (style, sampling, c)
    # the overflow depth is a tuning knob, not part of the result: 1, 7 and the default give the same coefficients
    for levels in ("1", "7"):
        os.environ["UHDR_HIP_HUFF_LEVELS"] = levels
        try:
            again = uhdr.huffman_decode(_dev(scan), [c.shape[:2] for c in coefs], w, h, sampling, 0)
        finally:
            del os.environ["UHDR_HIP_HUFF_LEVELS"]
        for c in range(len(coefs)):
            assert np.array_equal(again[c].cpu().numpy(), coefs[c]), (style, sampling, levels, c)


@pytest.mark.parametrize("sub_bits", ["256", "512", "2048", "4096"])  # the default is 1024
def test_subsequence_size_does_not_change_the_result(uhdr, sub_bits):
    rng = np.random.default_rng(223)
    w, h, sampling = 720, 400, [(2, 2), (1, 1), (1, 1)]
    coefs = _random_coefs(rng, w, h, sampling, "sparse")
    scan = L.huffman_encode_port(coefs, w, h, sampling, 0)
    os.environ["UHDR_HIP_HUFF_SUB_BITS"] = sub_bits
    try:
        got = uhdr.huffman_decode(_dev(scan), [c.shape[:2] for c in coefs], w, h, sampling, 0)
    finally:
        del os.environ["UHDR_HIP_HUFF_SUB_BITS"]
    for c in range(3):
        assert np.array_equal(got[c].cpu().numpy(), coefs[c]), c


def test_sync_decoder_equals_the_serial_kernel_and_libjpeg_on_a_reference_file(uhdr):
    """A 4K q95 4:2:0 file written by the REFERENCE encoder, parsed by the library's host parser: parallel decode ==
    serial single-lane decode == jpeg_read_coefficients()."""
    if L.ref() is None:
        pytest.skip("oracle/_ref not built")
    w, h = 3840, 2160
    img = synth.make_sdr_yuv420(w, h, align=8)
    jpeg = L.ref_jpeg_compress(img, 95)
    hdr, dev = uhdr.jpeg_to_coefficients(jpeg)
    assert hdr.scan.restart_interval == 0
    os.environ["UHDR_HIP_HUFF_SERIAL"] = "1"
    try:
        _, ser = uhdr.jpeg_to_coefficients(jpeg)
    finally:
        del os.environ["UHDR_HIP_HUFF_SERIAL"]
    ref = L.ref()
    buf = np.frombuffer(jpeg, dtype=np.uint8)
    want = [np.zeros(tuple(d.shape), dtype=np.int16) for d in dev]
    ptrs = (C.c_void_p * 3)(*[b.ctypes.data for b in want])
    qt = np.zeros((3, 64), dtype=np.uint16)
    bw, bh, nc = (C.c_int * 3)(), (C.c_int * 3)(), C.c_int(0)
    assert ref.ref_jpeg_read_coefficients(buf.ctypes.data, buf.size, ptrs, qt.ctypes.data, bw, bh, C.byref(nc)) == 0
    for c in range(3):
        g = dev[c].cpu().numpy()
        assert np.array_equal(g, want[c]), c
        assert np.array_equal(g, ser[c].cpu().numpy()), c


def test_sync_decoder_reports_corrupt_and_truncated_scans(uhdr):
    rng = np.random.default_rng(227)
    w, h, sampling = 640, 320, [(2, 2), (1, 1), (1, 1)]
    coefs = _random_coefs(rng, w, h, sampling, "sparse")
    scan = bytearray(L.huffman_encode_port(coefs, w, h, sampling, 0))
    shapes = [c.shape[:2] for c in coefs]
    with pytest.raises(A.UhdrError) as e:  # half the data is missing
        uhdr.huffman_decode(_dev(bytes(scan[: len(scan) // 2])), shapes, w, h, sampling, 0)
    assert e.value.code == A.UHDR_CODEC_INVALID_PARAM
    # random damage either still decodes to SOMETHING (Huffman streams rarely contain undefined codes) or is reported;
    # it must never crash or hang, and undamaged data still decodes afterwards
    bad = bytearray(scan)
    for i in rng.integers(100, len(bad) - 100, 200):
        if bad[i] != 0xFF and bad[i - 1] != 0xFF:
            bad[i] ^= 0x5A
    bad = bytes(bad).replace(b"\xff", b"\xfe")
    try:
        uhdr.huffman_decode(_dev(bad), shapes, w, h, sampling, 0)
    except A.UhdrError as err:
        assert err.code == A.UHDR_CODEC_INVALID_PARAM
    got = uhdr.huffman_decode(_dev(bytes(scan)), shapes, w, h, sampling, 0)
    for c in range(3):
        assert np.array_equal(got[c].cpu().numpy(), coefs[c]), c


def _stats(uhdr):
    st = A.Stats()
    uhdr.lib.uhdr_hip_get_stats(uhdr.ctx.handle, C.byref(st))
    return st


RST_CASES = [(640, 480, [(2, 2), (1, 1), (1, 1)], 10), (333, 211, [(2, 2), (1, 1), (1, 1)], 21), (500, 300, [(1, 1)] * 3, 63), (401, 203, [(2, 1), (1, 1), (1, 1)], 7),
             (1000, 400, [(1, 1)], 125), (1920, 1080, [(2, 2), (1, 1), (1, 1)], 120), (640, 480, [(2, 2), (1, 1), (1, 1)], 1199)]


@pytest.mark.parametrize("kind", ["sparse", "dense", "zero"])
def test_restart_marker_files_take_the_parallel_decoder(uhdr, kind):
    """Files WITH restart markers whose intervals are long enough (>= 320 bytes on average) go through the same
    self-synchronising decoder: markers dropped by the unstuff pass, the padding bits hopped over at the flagged interval
    starts, the DC prediction started over per interval.  Result == the coefficients that went in == the interval decoder
    (one lane per interval, UHDR_HIP_HUFF_RST_INTERVALS=1)."""
    rng = np.random.default_rng(307)
    took = 0
    for (w, h, sampling, ri) in RST_CASES:
        if kind == "dense" and w * h > 700_000:
            continue
        coefs = _random_coefs(rng, w, h, sampling, kind)
        scan = L.huffman_encode_port(coefs, w, h, sampling, ri)
        shapes = [c.shape[:2] for c in coefs]
        before = _stats(uhdr)
        got = uhdr.huffman_decode(_dev(scan), shapes, w, h, sampling, ri)
        after = _stats(uhdr)
        for c in range(len(coefs)):
            assert np.array_equal(got[c].cpu().numpy(), coefs[c]), (kind, w, h, ri, c)
        hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
        mcus = -(-w // (8 * hmax)) * -(-h // (8 * vmax)) if len(sampling) > 1 else -(-w // 8) * -(-h // 8)
        nseg = -(-mcus // ri)
        if nseg > 1 and len(scan) >= 4096 and len(scan) // nseg >= 320:
            if kind == "dense" and after.entropy_decode_parallel == before.entropy_decode_parallel:
                # ~1500 bits in every block and no EOBs: paths started in different places may not fall in step within an
                # interval; the decoder then gives the stream to the interval kernel (as it declines such marker-less files)
                assert after.entropy_decode_intervals == before.entropy_decode_intervals + 1
                continue
            assert after.entropy_decode_parallel == before.entropy_decode_parallel + 1, (kind, w, h, ri, len(scan), nseg)
            took += 1
            os.environ["UHDR_HIP_HUFF_RST_INTERVALS"] = "1"
            try:
                old = uhdr.huffman_decode(_dev(scan), shapes, w, h, sampling, ri)
            finally:
                del os.environ["UHDR_HIP_HUFF_RST_INTERVALS"]
            assert _stats(uhdr).entropy_decode_intervals == after.entropy_decode_intervals + 1
            for c in range(len(coefs)):
                assert np.array_equal(old[c].cpu().numpy(), coefs[c]), (kind, w, h, ri, c)
        else:
            assert after.entropy_decode_intervals == before.entropy_decode_intervals + 1, (kind, w, h, ri, len(scan), nseg)
    assert took >= {"zero": 1, "dense": 0, "sparse": 4}[kind], took


@pytest.mark.parametrize("sub_bits", ["1024", "2048", "4096"])  # (smaller ones lose the true path on this random data and end in the interval kernel)
def test_restart_files_subsequence_size_does_not_change_the_result(uhdr, sub_bits):
    rng = np.random.default_rng(311)
    w, h, sampling, ri = 720, 400, [(2, 2), (1, 1), (1, 1)], 9
    coefs = _random_coefs(rng, w, h, sampling, "sparse")
    scan = L.huffman_encode_port(coefs, w, h, sampling, ri)
    os.environ["UHDR_HIP_HUFF_SUB_BITS"] = sub_bits
    try:
        before = _stats(uhdr)
        got = uhdr.huffman_decode(_dev(scan), [c.shape[:2] for c in coefs], w, h, sampling, ri)
        assert _stats(uhdr).entropy_decode_parallel == before.entropy_decode_parallel + 1
    finally:
        del os.environ["UHDR_HIP_HUFF_SUB_BITS"]
    for c in range(3):
        assert np.array_equal(got[c].cpu().numpy(), coefs[c]), c


def test_restart_files_that_are_not_what_the_header_says_are_reported(uhdr):
    """A wrong restart interval in the header, a marker out of sequence, a missing marker, damaged data: the parallel
    decoder notices (marker count, an interval that ends where none is due) and the interval decoder words the error --
    the same errors as for short-interval files (test_gpu_parity.py::test_huffman_decode_error_behaviour)."""
    rng = np.random.default_rng(313)
    w, h, sampling, ri = 640, 480, [(2, 2), (1, 1), (1, 1)], 20
    coefs = _random_coefs(rng, w, h, sampling, "sparse")
    scan = bytearray(L.huffman_encode_port(coefs, w, h, sampling, ri))
    shapes = [c.shape[:2] for c in coefs]
    assert len(scan) // 60 >= 320
    good = uhdr.huffman_decode(_dev(bytes(scan)), shapes, w, h, sampling, ri)
    assert np.array_equal(good[0].cpu().numpy(), coefs[0])

    def fails(buf, ri_=ri):
        with pytest.raises(A.UhdrError) as e:
            uhdr.huffman_decode(_dev(bytes(buf)), shapes, w, h, sampling, ri_)
        assert e.value.code == A.UHDR_CODEC_INVALID_PARAM, e.value

    fails(scan, 40)  # the stream has a marker every 20 MCUs
    fails(scan, 10)
    k = scan.find(b"\xff\xd3")
    swapped = bytearray(scan)
    swapped[k + 1] = 0xD6  # out of sequence
    fails(swapped)
    dropped = bytearray(scan)
    del dropped[k:k + 2]  # one marker missing
    fails(dropped)
    # and the context is still usable
    again = uhdr.huffman_decode(_dev(bytes(scan)), shapes, w, h, sampling, ri)
    for c in range(3):
        assert np.array_equal(again[c].cpu().numpy(), coefs[c]), c


@pytest.mark.parametrize("ri", [0, 6])
def test_two_scans_at_once_equal_two_calls(uhdr, ri):
    """uhdr_hip_huffman_encode2_dev / _decode2_dev (the base image's and the gain map's scans of one file, the second on the context's
    auxiliary stream) give the bytes and coefficients of two single-scan calls -- and of the oracle --, repeatedly, with the roles of
    the scans exchanged, and an error in either scan is the call's error."""
    import torch

    rng = np.random.default_rng(977 + ri)
    wa, ha, sa = 1280, 720, [(2, 2), (1, 1), (1, 1)]
    wb, hb, sb = 640, 360, [(1, 1)] * 3
    ca, cb = _random_coefs(rng, wa, ha, sa, "dense"), _random_coefs(rng, wb, hb, sb, "sparse")
    da, db = [torch.from_numpy(c).to("cuda:0") for c in ca], [torch.from_numpy(c).to("cuda:0") for c in cb]
    one_a = bytes(uhdr.huffman_encode(da, wa, ha, sa, ri).cpu().numpy())
    one_b = bytes(uhdr.huffman_encode(db, wb, hb, sb, ri).cpu().numpy())
    assert one_a == L.huffman_encode_port(ca, wa, ha, sa, ri)
    for rep in range(3):
        ea, eb = uhdr.huffman_encode2(da, wa, ha, sa, db, wb, hb, sb, ri)
        assert bytes(ea.cpu().numpy()) == one_a and bytes(eb.cpu().numpy()) == one_b, rep
        eb2, ea2 = uhdr.huffman_encode2(db, wb, hb, sb, da, wa, ha, sa, ri)
        assert bytes(ea2.cpu().numpy()) == one_a and bytes(eb2.cpu().numpy()) == one_b, rep
        ga, gb = uhdr.huffman_decode2(_dev(one_a), [c.shape[:2] for c in ca], wa, ha, sa, _dev(one_b), [c.shape[:2] for c in cb], wb, hb, sb, ri)
        for c in range(3):
            assert np.array_equal(ga[c].cpu().numpy(), ca[c]) and np.array_equal(gb[c].cpu().numpy(), cb[c]), (rep, c)
    # a scan cut short: the same status whichever slot it is in
    short = _dev(one_b[: len(one_b) // 2])
    with pytest.raises(Exception) as e1:
        uhdr.huffman_decode(short, [c.shape[:2] for c in cb], wb, hb, sb, ri)
    for first in (True, False):
        with pytest.raises(Exception) as e2:
            if first:
                uhdr.huffman_decode2(short, [c.shape[:2] for c in cb], wb, hb, sb, _dev(one_a), [c.shape[:2] for c in ca], wa, ha, sa, ri)
            else:
                uhdr.huffman_decode2(_dev(one_a), [c.shape[:2] for c in ca], wa, ha, sa, short, [c.shape[:2] for c in cb], wb, hb, sb, ri)
        assert str(e2.value) == str(e1.value)
    # and the context still works
    ga, gb = uhdr.huffman_decode2(_dev(one_a), [c.shape[:2] for c in ca], wa, ha, sa, _dev(one_b), [c.shape[:2] for c in cb], wb, hb, sb, ri)
    assert np.array_equal(ga[0].cpu().numpy(), ca[0]) and np.array_equal(gb[2].cpu().numpy(), cb[2])
