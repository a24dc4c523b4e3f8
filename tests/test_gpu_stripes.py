"""GPU: row-striped two-pass generateGainMap (BASELINE config 4) with the REAL kernels.

(1) two stripes processed one after the other on one GPU (pass 1 per stripe, the merge of jpegr.cpp:932-938, pass 2 per
    stripe) give the whole-image gain map and metadata exactly;
(2) the C++ host layer's striped entry point with an RCCL communicator (one rank here: the collective still executes
    on the hardware) equals the single-device two-pass result, with a single host synchronisation."""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import stripes, synth
from libultrahdr_amd.images import Image
from libultrahdr_amd.ultrahdr import UltraHdr

pytestmark = pytest.mark.gpu


def _md_tuple(md):
    return (list(md.max_content_boost), list(md.min_content_boost), list(md.gamma), list(md.offset_sdr), list(md.offset_hdr),
            md.hdr_capacity_min, md.hdr_capacity_max, md.use_base_cg)


def _slice_rows(img: Image, row0: int, rows: int) -> Image:
    """Host image -> a new host image holding rows [row0, row0 + rows) (chroma planes follow)."""
    out = Image(img.fmt, img.w, rows, img.raw.cg, img.raw.ct, img.raw.range, align=64)
    for c in range(len(img.layout)):
        if img.layout[c] is None:
            continue
        full = img.valid(c)
        f = full.shape[0] / img.h
        out.valid(c)[:] = full[int(row0 * f): int((row0 + rows) * f)]
    return out


@pytest.mark.parametrize("multichannel,scale", [(True, 1), (False, 4), (True, 2)])
def test_two_stripes_on_one_gpu_equal_the_whole_image(hip_ctx, multichannel, scale):
    w, h = 768, 512
    sdr, hdr = synth.make_sdr_yuv420(w, h), synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
    u = UltraHdr(ctx=hip_ctx, mapDimensionScaleFactor=scale, useMultiChannelGainMap=multichannel, preset=A.UHDR_USAGE_BEST_QUALITY)
    cfg = u.encode_cfg()
    md_w, gm_w = u.generateGainMap(sdr.to("cuda:0"), hdr.to("cuda:0"))
    hip_ctx.synchronize()
    whole = gm_w.to_host().valid(0)
    nch = 3 if multichannel else 1
    # split at a granule boundary that is NOT half the image: the stripes differ in size
    rows = [(0, 192), (192, h - 192)]
    import torch

    parts, gains = [], []
    lib, hnd = u.lib, hip_ctx.handle
    for r0, n in rows:
        s_, h_ = _slice_rows(sdr, r0, n).to("cuda:0"), _slice_rows(hdr, r0, n).to("cuda:0")
        g = torch.empty((n // scale) * (w // scale) * nch, dtype=torch.float32, device="cuda:0")
        mm = torch.empty(6, dtype=torch.float32, device="cuda:0")
        ubc = C.c_int(1)
        A.check(lib.uhdr_hip_generate_gainmap_pass1_dev(hnd, C.byref(s_.raw), C.byref(h_.raw), C.byref(cfg), C.c_void_p(g.data_ptr()),
                                                        C.c_void_p(mm.data_ptr()), C.byref(ubc)))
        hip_ctx.synchronize()
        parts.append(mm.cpu().tolist())
        gains.append((g, n, ubc.value))
    fin, md = stripes.finalize_minmax(cfg, hdr.raw.ct, gains[0][2], stripes.merge_minmax(parts))
    assert _md_tuple(md) == _md_tuple(md_w)
    got = []
    for (g, n, _), (r0, _) in zip(gains, rows):
        gm = Image(A.UHDR_IMG_FMT_24bppRGB888 if multichannel else A.UHDR_IMG_FMT_8bppYCbCr400, w // scale, n // scale, align=64, device="cuda:0")
        A.check(lib.uhdr_hip_generate_gainmap_pass2_dev(hnd, C.c_void_p(g.data_ptr()), (C.c_float * 6)(*fin), C.byref(cfg), C.byref(gm.raw)))
        hip_ctx.synchronize()
        got.append(gm.to_host().valid(0))
    assert np.array_equal(np.concatenate(got, axis=0), whole)


@pytest.mark.parametrize("multichannel,scale", [(True, 1), (False, 4)])
def test_striped_entry_point_with_an_rccl_communicator(multichannel, scale):
    """uhdr_hip_generate_gainmap_striped_dev on a context that owns a (one-rank) RCCL communicator: the all-reduce runs
    on the library's stream, the range is finalised on the device, and the result is the single-device two-pass one."""
    from libultrahdr_amd.ultrahdr import Context

    ctx = Context(0)
    try:
        assert stripes.init_comm(ctx) == 1  # ncclCommCount
        w, h = 1024, 512
        sdr, hdr = synth.make_sdr_yuv420(w, h, seed=7).to("cuda:0"), synth.make_hdr_p010(w, h, ct=A.UHDR_CT_PQ, seed=8).to("cuda:0")
        u = UltraHdr(ctx=ctx, mapDimensionScaleFactor=scale, useMultiChannelGainMap=multichannel, preset=A.UHDR_USAGE_BEST_QUALITY)
        md_w, gm_w = u.generateGainMap(sdr, hdr)
        ctx.synchronize()
        gm = Image(A.UHDR_IMG_FMT_24bppRGB888 if multichannel else A.UHDR_IMG_FMT_8bppYCbCr400, w // scale, h // scale, align=64, device="cuda:0")
        md = stripes.generate_gainmap_striped(u, sdr, hdr, u.encode_cfg(), gm)
        assert _md_tuple(md) == _md_tuple(md_w)
        assert np.array_equal(gm.to_host().valid(0), gm_w.to_host().valid(0))
        # user hints take the device finalisation's other branches
        u2 = UltraHdr(ctx=ctx, mapDimensionScaleFactor=scale, useMultiChannelGainMap=multichannel, preset=A.UHDR_USAGE_BEST_QUALITY,
                      minContentBoost=1.25, maxContentBoost=3.0)
        md_w2, gm_w2 = u2.generateGainMap(sdr, hdr)
        ctx.synchronize()
        md2 = stripes.generate_gainmap_striped(u2, sdr, hdr, u2.encode_cfg(), gm)
        assert _md_tuple(md2) == _md_tuple(md_w2)
        assert np.array_equal(gm.to_host().valid(0), gm_w2.to_host().valid(0))
    finally:
        ctx.close()
