"""Stream contract of the Python binding (libultrahdr_amd/ultrahdr.py: Context.ordered): device-buffer calls are
ordered against torch's current stream by events, so inputs still being written by torch kernels are not read early and
returned tensors can be consumed at once -- no ctx.synchronize() anywhere in this file."""
import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from oracle import loader as L

pytestmark = pytest.mark.gpu


def test_inputs_produced_by_torch_and_outputs_consumed_by_torch_without_host_sync(hip_ctx):
    import torch

    from libultrahdr_amd.ultrahdr import UltraHdr

    u = UltraHdr(ctx=hip_ctx)
    rng = np.random.default_rng(5)
    w, h = 2048, 1024
    plane = rng.integers(0, 256, (h, w), dtype=np.uint8)
    qt = u.quant_table(90, False)
    want_coef = L.fdct_quant_port(plane, w, w // 8, h // 8, qt)
    want_plane = L.idct_dequant_port(want_coef, qt)
    for _ in range(5):
        # the input is the END of a chain of torch kernels (a large reduction keeps torch's stream busy first)
        junk = torch.randn(64 << 20, device="cuda:0")
        src = torch.from_numpy(plane).to("cuda:0", non_blocking=True)
        junk2 = (junk * 2.0).sum()
        dplane = (src.to(torch.int16) + (junk2 * 0).to(torch.int16)).to(torch.uint8)
        coef = u.fdct_quant(dplane, w, w // 8, h // 8, qt)
        back = u.idct_dequant(coef, qt)
        assert np.array_equal(coef.cpu().numpy(), want_coef)  # .cpu() on torch's stream, straight after the launch
        assert np.array_equal(back.cpu().numpy(), want_plane)
    # an Image written by a torch kernel right before applyGainMap, destination read right after
    sdr = synth.make_sdr_yuv420(1024, 512)
    gm = synth.make_gainmap(256, 128, 1)
    md = synth.default_metadata()
    want = L.apply_gainmap("port", sdr, gm, md, A.UHDR_CT_LINEAR)
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    dsdr = Image(sdr.fmt, sdr.w, sdr.h, sdr.raw.cg, sdr.raw.ct, sdr.raw.range, device="cuda:0")
    dgm = gm.to("cuda:0")
    host = torch.from_numpy(sdr.buf)
    for _ in range(5):
        junk = torch.randn(64 << 20, device="cuda:0").mul_(3.0)
        dsdr.buf.zero_()
        dsdr.buf.copy_(host.to("cuda:0", non_blocking=True))
        dest = Image(f16, sdr.w, sdr.h, align=2, device="cuda:0", fill=0xAB)
        u.applyGainMap(dsdr, dgm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest)
        assert np.array_equal(dest.to_host().valid(0), want.valid(0))
        del junk
