"""Run as a subprocess by tests/test_gpu_facade.py: uhdr_decode through the facade with and without
acceleration, the decoded image AND the gain-map image of uhdr_get_decoded_gainmap_image compared; one '=== name: MATCH' line
per case on stdout, and one '--- name {stage table}' line per section: the library's own tallies (uhdr_hip_seam_stats) of the
accelerated calls made in that section."""
import json
import sys

import numpy as np

from libultrahdr_amd import capi as A
from libultrahdr_amd import facade as FA
from libultrahdr_amd import synth


_section = [None]


def mark(s):
    """Close the running section -- print its stage table -- and open the next."""
    if _section[0] is not None:
        print("--- " + _section[0] + " " + json.dumps(A.seam_stats(reset=True)), flush=True)
    else:
        A.seam_stats(reset=True)
    _section[0] = s


def main():
    w, h = int(sys.argv[1]), int(sys.argv[2])
    hdr = synth.make_hdr_p010(w, h, ct=A.UHDR_CT_HLG)
    sdr = synth.make_sdr_yuv420(w, h)
    lin, f16 = A.UHDR_CT_LINEAR, A.UHDR_IMG_FMT_64bppRGBAHalfFloat
    for name, multi, scale in (("rgb_map", 1, 1), ("luma_map_scale4", 0, 4)):
        mark(name + " encode")
        jpg = FA.encode(hdr, sdr, gpu=False, preset=A.UHDR_USAGE_REALTIME, multi_channel=multi, scale=scale)
        mark(name + " cpu")
        px0, gm0 = FA.decode(jpg, lin, f16, gpu=False, want_gainmap=True)
        mark(name + " asked")
        px1, gm1 = FA.decode(jpg, lin, f16, gpu=True, want_gainmap=True)
        ok = np.array_equal(px0, px1) and np.array_equal(gm0, gm1) and gm0.any()
        print(f"=== {name} asked: {'MATCH' if ok else 'DIFFER'} map {gm0.shape}", flush=True)
        mark(name + " not_asked")
        px2 = FA.decode(jpg, lin, f16, gpu=True)
        print(f"=== {name} not_asked: {'MATCH' if np.array_equal(px0, px2) else 'DIFFER'}", flush=True)
        mark(name + " reset")
        px3, gm3 = FA.decode(jpg, lin, f16, gpu=True, want_gainmap=True, decodes=2)
        ok = np.array_equal(px0, px3) and np.array_equal(gm0, gm3)
        print(f"=== {name} reset: {'MATCH' if ok else 'DIFFER'}", flush=True)
        mark(name + " sdr_out")
        s0, g0 = FA.decode(jpg, A.UHDR_CT_SRGB, A.UHDR_IMG_FMT_32bppRGBA8888, gpu=False, want_gainmap=True)
        s1, g1 = FA.decode(jpg, A.UHDR_CT_SRGB, A.UHDR_IMG_FMT_32bppRGBA8888, gpu=True, want_gainmap=True)
        ok = np.array_equal(s0, s1) and np.array_equal(g0, g1) and np.array_equal(g0, gm0)
        print(f"=== {name} sdr_out: {'MATCH' if ok else 'DIFFER'}", flush=True)
        mark(name + " effects")
        fx = [("mirror", 0)]
        e0, m0 = FA.decode(jpg, lin, f16, gpu=False, effects=fx, want_gainmap=True)
        e1, m1 = FA.decode(jpg, lin, f16, gpu=True, effects=fx, want_gainmap=True)
        ok = np.array_equal(e0, e1) and np.array_equal(m0, m1) and not np.array_equal(m0, gm0)
        print(f"=== {name} effects: {'MATCH' if ok else 'DIFFER'}", flush=True)
        mark(name + " end")


if __name__ == "__main__":
    main()
