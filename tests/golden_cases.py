"""Case table shared by tests/golden/make_golden.py (generator, runs the real reference) and
tests/test_golden.py (checks the oracle / the HIP path against the committed outputs)."""
import ctypes as C

import numpy as np

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image

W, H = 128, 64


def jpeg_image():
    return synth.make_sdr_yuv420(W, H, seed=4242, noise=0.08)


def jpeg_rgb_map():
    return synth.make_gainmap(96, 48, 3, seed=4343)


def cases():
    c = {}
    for tag, (ch, alpha, scale) in {"y400_s4": (1, False, 4), "rgb_s1": (3, False, 1), "rgba_s1": (3, True, 1), "y400_s2": (1, False, 2)}.items():
        for ct_name, ct in (("lin", A.UHDR_CT_LINEAR), ("hlg", A.UHDR_CT_HLG), ("pq", A.UHDR_CT_PQ)):
            c[f"apply_{tag}_{ct_name}"] = dict(op="apply", ch=ch, alpha=alpha, scale=scale, ct=ct, use_base_cg=0)
    c["apply_y400_s4_lin_basecg"] = dict(op="apply", ch=1, alpha=False, scale=4, ct=A.UHDR_CT_LINEAR, use_base_cg=1)
    c["gen_default_hlg"] = dict(op="gen", hdr="p010", ct=A.UHDR_CT_HLG, cfg=dict())
    c["gen_default_pq"] = dict(op="gen", hdr="p010", ct=A.UHDR_CT_PQ, cfg=dict())
    c["gen_rt_s4_1ch"] = dict(op="gen", hdr="p010", ct=A.UHDR_CT_HLG,
                              cfg=dict(map_dimension_scale_factor=4, use_multi_channel_gainmap=0, preset=A.UHDR_USAGE_REALTIME))
    c["gen_api0_1010102"] = dict(op="gen", hdr="1010102", ct=A.UHDR_CT_PQ, cfg=dict(preset=A.UHDR_USAGE_REALTIME, use_luminance=0))
    c["tonemap_p010_hlg"] = dict(op="tonemap", hdr="p010", ct=A.UHDR_CT_HLG)
    c["tonemap_1010102_pq"] = dict(op="tonemap", hdr="1010102", ct=A.UHDR_CT_PQ)
    c["convert_yuv_709_601_420"] = dict(op="convert_yuv", src=0, dst=1)
    c["convert_yuv_2100_709_420"] = dict(op="convert_yuv", src=2, dst=0)
    c["rgb1010102_to_p010"] = dict(op="raw2ycc", fmt=A.UHDR_IMG_FMT_32bppRGBA1010102, chroma=True)
    c["rgba8888_to_444"] = dict(op="raw2ycc", fmt=A.UHDR_IMG_FMT_32bppRGBA8888, chroma=False)
    c["rgba8888_to_420"] = dict(op="raw2ycc", fmt=A.UHDR_IMG_FMT_32bppRGBA8888, chroma=True)
    return c


def inputs(case):
    op = case["op"]
    if op == "apply":
        sdr = synth.make_sdr_yuv420(W, H, seed=11, noise=0.05)
        s = case["scale"]
        gm = synth.make_gainmap(W // s, H // s, case["ch"], case["alpha"], seed=12, cg=A.UHDR_CG_BT_2100)
        md = synth.default_metadata(use_base_cg=case["use_base_cg"], per_channel=(case["ch"] == 3))
        return sdr, gm, md
    if op in ("gen", "tonemap"):
        if case["hdr"] == "p010":
            hdr = synth.make_hdr_p010(W, H, seed=21, ct=case["ct"], noise=0.04)
            sdr = synth.make_sdr_yuv420(W, H, seed=22, noise=0.04)
        else:
            hdr = synth.make_hdr_rgba1010102(W, H, seed=21, ct=case["ct"], noise=0.04)
            sdr = synth.make_sdr_rgba8888(W, H, seed=22, noise=0.04)
        return sdr, hdr
    if op == "convert_yuv":
        img = synth.make_sdr_yuv420(W, H, seed=31, cg=case["src"], noise=0.08)
        return (img,)
    if op == "raw2ycc":
        rng = np.random.default_rng(41)
        img = Image(case["fmt"], W, H, A.UHDR_CG_DISPLAY_P3, A.UHDR_CT_SRGB, A.UHDR_CR_FULL_RANGE)
        img.buf[:] = rng.integers(0, 256, img.buf.size, dtype=np.uint8)
        return (img,)
    raise KeyError(op)


def planes(img: Image):
    return {f"plane{i}": np.ascontiguousarray(p) for i, p in enumerate(img.to_host().planes_valid())}


def run(case, kind):
    """kind: 'ref' | 'port' (oracle/loader) -> dict of output arrays."""
    from oracle import loader as L

    op = case["op"]
    if op == "apply":
        sdr, gm, md = inputs(case)
        return planes(L.apply_gainmap(kind, sdr, gm, md, case["ct"]))
    if op == "gen":
        sdr, hdr = inputs(case)
        md, gm = L.generate_gainmap(kind, sdr, hdr, A.default_encode_cfg(**case["cfg"]))
        out = planes(gm)
        out["metadata"] = np.frombuffer(bytes(md), dtype=np.uint8).copy()
        return out
    if op == "tonemap":
        _, hdr = inputs(case)
        return planes(L.tone_map(kind, hdr))
    if op == "convert_yuv":
        (img,) = inputs(case)
        return planes(L.convert_yuv(kind, img, case["src"], case["dst"]))
    if op == "raw2ycc":
        (img,) = inputs(case)
        return planes(L.convert_raw_input_to_ycbcr(kind, img, case["chroma"]))
    raise KeyError(op)
