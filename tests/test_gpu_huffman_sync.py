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
