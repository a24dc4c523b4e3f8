"""Device chains added after the round's GPU budget was spent: every library call in here is exercised by GPU tests that
did run (tests/test_gpu_parity.py: test_huffman_*, test_apply_gainmap_from_coefficients*), the compositions below were
only checked on the CPU side (host parser, oracle stand-ins).  The file sorts last so that, under `pytest -x`, nothing in it
can keep the validated suite from running."""
import ctypes as C  # noqa: F401
import os

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from oracle import loader as L

import golden_cases as G

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz"))


@pytest.fixture(scope="module")
def uhdr(hip_ctx):
    from libultrahdr_amd.ultrahdr import UltraHdr

    return UltraHdr(ctx=hip_ctx)


def oracle_kind():
    return "ref" if L.ref() is not None else "port"


@pytest.mark.parametrize("q", [95, 50])
def test_hip_entropy_stage_against_reference_vectors(hip_ctx, q):
    """The device decoder reads the reference encoder's bytes (one interval: a single lane) back to libjpeg's
    coefficients; the device encoder's restart-interval stream of those coefficients decodes to them again."""
    import torch

    from libultrahdr_amd.ultrahdr import UltraHdr

    u = UltraHdr(ctx=hip_ctx)
    for tag, w, h, sampling, ri in (("jpeg", G.W, G.H, [(2, 2), (1, 1), (1, 1)], 4), ("jpegrgb", 96, 48, [(1, 1)] * 3, 6)):
        coefs = [np.ascontiguousarray(GOLD[f"{tag}_q{q}/coef{c}"]) for c in range(3)]
        shapes = [c.shape[:2] for c in coefs]
        scan = torch.from_numpy(GOLD[f"{tag}_q{q}/scan"].copy()).to("cuda:0")
        got = u.huffman_decode(scan, shapes, w, h, sampling, 0)
        assert all(np.array_equal(g.cpu().numpy(), c) for g, c in zip(got, coefs)), tag
        dev = [torch.from_numpy(c).to("cuda:0") for c in coefs]
        stream = u.huffman_encode(dev, w, h, sampling, ri)
        assert stream.cpu().numpy().tobytes() == L.huffman_encode_port(coefs, w, h, sampling, ri)
        back = u.huffman_decode(stream.clone(), shapes, w, h, sampling, ri)
        assert all(torch.equal(b, d) for b, d in zip(back, dev)), tag


def test_jpeg_file_to_hdr_pixels_without_the_cpu_decoder(uhdr):
    """The whole device decode chain from file bytes: headers parsed by the library's host parser, entropy-coded data
    decoded per restart interval, the base image's IDCT inside applyGainMap -- against the oracle's decode of the same
    coefficients.  The file comes from the oracle's encoder (restart interval 3, i.e. what uhdr_hip_huffman_encode_dev writes)."""
    w, h, ri = 256, 96, 3
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    sampling = [(2, 2), (1, 1), (1, 1)]
    sdr = synth.make_sdr_yuv420(w, h, align=8, noise=0.05)
    ql, qc = uhdr.quant_table(92, False), uhdr.quant_table(92, True)
    coefs = [L.fdct_quant_port(np.ascontiguousarray(sdr.valid(c)), sdr.valid(c).shape[1], sdr.valid(c).shape[1] // 8, sdr.valid(c).shape[0] // 8,
                               ql if c == 0 else qc) for c in range(3)]
    jpeg = L.jpeg_assemble_port(coefs, w, h, sampling, ri, ql, qc, L.huffman_encode_port(coefs, w, h, sampling, ri))
    hdr, dev = uhdr.jpeg_to_coefficients(jpeg)
    assert hdr.scan.restart_interval == ri and (hdr.scan.w, hdr.scan.h) == (w, h)
    for c in range(3):
        assert np.array_equal(dev[c].cpu().numpy(), coefs[c]), c
    qts = [np.frombuffer(hdr.qtable[c], dtype=np.uint16).copy() for c in range(3)]
    assert np.array_equal(qts[0], ql) and np.array_equal(qts[1], qc)
    gm = synth.make_gainmap(w // 4, h // 4, 1)
    md = synth.default_metadata()
    dest = Image(f16, w, h, align=2, device="cuda:0")
    uhdr.applyGainMapFromCoefficients(dev, qts, w, h, A.UHDR_CG_BT_709, gm.to("cuda:0"), md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest)
    uhdr.ctx.synchronize()
    dec = Image(A.UHDR_IMG_FMT_12bppYCbCr420, w, h, A.UHDR_CG_BT_709, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE, align=2)
    for c in range(3):
        dec.valid(c)[:] = L.idct_dequant_port(coefs[c], ql if c == 0 else qc)[: dec.valid(c).shape[0], : dec.valid(c).shape[1]]
    want = L.apply_gainmap(oracle_kind(), dec, gm, md, A.UHDR_CT_LINEAR)
    assert np.array_equal(dest.to_host().valid(0), want.valid(0))
