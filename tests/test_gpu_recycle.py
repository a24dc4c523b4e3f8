"""GPU: uhdr_hip_recycle (round 6) -- what the facade's context pool calls before it parks a context: the next user finds no latched state and no
oversized buffers, and the context still computes the same pixels."""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image
from oracle import loader as L

pytestmark = pytest.mark.gpu


def test_a_recycled_context_forgets_and_still_works():
    import torch

    from libultrahdr_amd.ultrahdr import Context, UltraHdr

    ctx = Context(0)
    u = UltraHdr(ctx=ctx)
    w, h = 1024, 512
    sdr = synth.make_sdr_yuv420(w, h)
    gm = synth.make_gainmap(w // 4, h // 4, 1)
    md = synth.default_metadata()
    f16 = A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    want = L.apply_gainmap("port", sdr, gm, md, A.UHDR_CT_LINEAR)

    def run():
        dest = Image(f16, w, h, align=2)
        u.applyGainMap(sdr, gm, md, A.UHDR_CT_LINEAR, f16, A.FLT_MAX, dest)  # host images: staged through the context's device buffers
        return dest

    assert np.array_equal(run().valid(0), want.valid(0))
    # an entropy round trip leaves hints, counters and scratch behind
    qy, qc = u.quant_table(90, False), u.quant_table(90, True)
    coefs = [u.fdct_quant(sdr.to("cuda:0").plane_tensor(c), sdr.layout[c][1], sdr.layout[c][2] // 8, sdr.layout[c][0] // 8, qy if c == 0 else qc) for c in range(3)]
    S = [(2, 2), (1, 1), (1, 1)]
    stream = u.huffman_encode(coefs, w, h, S, 0)
    back = u.huffman_decode(stream, [tuple(c.shape[:2]) for c in coefs], w, h, S, 0)
    assert all(torch.equal(a, b) for a, b in zip(back, coefs))
    st = A.Stats()
    ctx.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st))
    assert st.entropy_encode_stream == 1 and st.entropy_decode_parallel + st.entropy_decode_single_lane == 1
    free0 = torch.cuda.mem_get_info(0)[0]
    assert ctx.lib.uhdr_hip_recycle(ctx.handle, 0) == 0  # bound to device 0; keep nothing
    free1 = torch.cuda.mem_get_info(0)[0]
    assert free1 > free0, "recycling with keep_bytes = 0 must give device memory back"
    ctx.lib.uhdr_hip_get_stats(ctx.handle, C.byref(st))
    assert st.entropy_encode_stream == 0 and st.entropy_decode_parallel == 0 and st.resident_hits == 0
    assert np.array_equal(run().valid(0), want.valid(0))  # buffers come back on demand
    back = u.huffman_decode(stream, [tuple(c.shape[:2]) for c in coefs], w, h, S, 0)
    assert all(torch.equal(a, b) for a, b in zip(back, coefs))
    assert ctx.lib.uhdr_hip_recycle(ctx.handle, 1 << 30) == 0  # within the budget: nothing to free, still fine
    assert np.array_equal(run().valid(0), want.valid(0))
    assert ctx.lib.uhdr_hip_current_device() == 0
    ctx.close()
