"""Host-side logic that needs no GPU: image containers, stripe partitioning, synthetic inputs,
and the LUT-index identity the kernels rely on."""
import ctypes as C
import hashlib
import subprocess
import tempfile

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from libultrahdr_amd import synth
from libultrahdr_amd.images import Image, plane_layout, stripe_view
from libultrahdr_amd.stripes import partition_rows, stripe_granule


def test_image_layout_matches_reference_allocator():
    # uhdr_raw_image_ext (ultrahdr_api.cpp:55-117): strides aligned to 64, chroma stride = aligned/2
    im = Image(A.UHDR_IMG_FMT_12bppYCbCr420, 1282, 722)
    assert list(im.raw.stride) == [1344, 672, 672]
    assert im.valid(0).shape == (722, 1282) and im.valid(1).shape == (361, 641)
    p = Image(A.UHDR_IMG_FMT_24bppYCbCrP010, 1280, 720)
    assert list(p.raw.stride) == [1280, 1280, 0] and p.plane(1).dtype == np.uint16 and p.valid(1).shape == (360, 1280)
    r = Image(A.UHDR_IMG_FMT_24bppRGB888, 100, 10)
    assert r.raw.stride[0] == 128 and r.valid(0).shape == (10, 300)
    f = Image(A.UHDR_IMG_FMT_64bppRGBAHalfFloat, 33, 5, align=1)
    assert f.raw.stride[0] == 33 and f.plane(0).dtype == np.uint64
    assert plane_layout(A.UHDR_IMG_FMT_8bppYCbCr400, 10, 4)[1] is None


def test_stripe_view_points_into_parent():
    im = synth.make_sdr_yuv420(64, 64)
    v = stripe_view(im, 16, 32)
    assert v.h == 32 and v.planes[0] == im.raw.planes[0] + 16 * im.raw.stride[0]
    assert v.planes[1] == im.raw.planes[1] + 8 * im.raw.stride[1]
    with pytest.raises(AssertionError):
        stripe_view(im, 3, 32)


@pytest.mark.parametrize("h,world,gran", [(2160, 8, 16), (4320, 8, 16), (16384, 8, 16), (720, 4, 16), (100, 8, 16), (17, 2, 16), (2160, 3, 48)])
def test_partition_rows(h, world, gran):
    parts = partition_rows(h, world, gran)
    assert len(parts) == world and parts[0][0] == 0
    assert sum(n for _, n in parts) == h
    for (r0, n), (r1, _) in zip(parts, parts[1:]):
        assert r0 + n == r1
    for r0, n in parts[:-1]:
        assert r0 % gran == 0 and (n % gran == 0 or r0 + n == h)
    sizes = [n for _, n in parts if n]
    assert max(sizes) - min(sizes) <= gran or sizes[-1] < gran + max(sizes)
    assert stripe_granule(4) == 16 and stripe_granule(3) == 48 and stripe_granule(1, 8) == 8


def test_synthetic_inputs_are_deterministic():
    a, b = synth.make_sdr_yuv420(128, 64), synth.make_sdr_yuv420(128, 64)
    assert synth.checksum(a) == synth.checksum(b)
    assert synth.checksum(a) != synth.checksum(synth.make_sdr_yuv420(128, 64, seed=99))
    p = synth.make_hdr_p010(128, 64)
    y = p.valid(0)
    assert (y & 63).max() == 0 and (y >> 6).min() >= 64 and (y >> 6).max() <= 940
    r = synth.make_hdr_rgba1010102(32, 16).valid(0)
    assert ((r >> 30) == 3).all()
    # canonical generator pinned: checksum of the 1280x720 seed-1234 pair
    h = hashlib.sha256()
    for im in (synth.make_sdr_yuv420(1280, 720), synth.make_hdr_p010(1280, 720)):
        for pl in im.planes_valid():
            h.update(np.ascontiguousarray(pl).tobytes())
    assert h.hexdigest()[:16] == open(__file__.replace("test_host_logic.py", "golden/synth_1280x720.sha256")).read().strip()[:16]


LUT_INDEX_C = r"""
#include <stdint.h>
#include <string.h>
#include <stdio.h>
int main(void) {
  const float Ns[2] = {1023.0f, 65535.0f};
  for (int k = 0; k < 2; k++) {
    unsigned long long bad = 0;
    for (uint32_t u = 0; u <= 0x3F800000u; u += 1) {
      float x; memcpy(&x, &u, 4);
      float f = x * Ns[k];
      if ((int)((double)f + 0.5) != (int)(f + 0.5f)) bad++;
    }
    printf("%llu\n", bad);
  }
  return 0;
}
"""


def test_lut_index_float_equivalence():
    """device_math.h::lut_index_f32 replaces int(double(x*(N-1)) + 0.5) by a float add for
    N = 1024 and 65536; exhaustive over every float in [0, 1] (1.07e9 values x 2, a few seconds)."""
    with tempfile.TemporaryDirectory() as d:
        open(f"{d}/t.c", "w").write(LUT_INDEX_C)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", f"{d}/t", f"{d}/t.c"])
        out = subprocess.check_output([f"{d}/t"], text=True).split()
    assert out == ["0", "0"]


def test_jpeg_rgb_to_ycc_constant_sets_are_equivalent():
    """jccolor.c's rgb_ycc_convert: libjpeg 6b / libjpeg-turbo use 0.16874 / 0.33126 / 0.41869 / 0.08131,
    IJG 9 the longer 0.168735892 ... -- the 16-bit fixed-point results are identical for every 8-bit
    (r, g, b), which is why uhdr_hip_jpeg_rgb_to_ycc has no variant argument (the decode direction does)."""
    fix = lambda x: int(x * 65536.0 + 0.5)
    sets = [(0.29900, 0.58700, 0.11400, 0.16874, 0.33126, 0.50000, 0.41869, 0.08131),
            (0.299, 0.587, 0.114, 0.168735892, 0.331264108, 0.5, 0.418687589, 0.081312411)]
    g, b = np.meshgrid(np.arange(256, dtype=np.int32), np.arange(256, dtype=np.int32), indexing="ij")
    half, off = 1 << 15, 128 << 16
    for r in range(256):  # one red value at a time: 64 K triples per step, no large temporaries
        res = []
        for c in sets:
            y = (fix(c[0]) * r + fix(c[1]) * g + fix(c[2]) * b + half) >> 16
            cb = (-fix(c[3]) * r - fix(c[4]) * g + fix(c[5]) * b + off + half - 1) >> 16
            cr = (fix(c[5]) * r - fix(c[6]) * g - fix(c[7]) * b + off + half - 1) >> 16
            res.append((y, cb, cr))
        for a, bb in zip(*res):
            assert np.array_equal(a, bb), r


def test_half_subnormal_formula():
    """apply_gainmap.hip::half_small_pair: for non-negative v, floatToHalf's sub-normal branch
    (((0x7FF000 + mant(b)) >> (125 - exp(b))) + 1) >> 1 with b = bits + 0x1000 equals (floor(v * 2^25) + 1) >> 1 on the
    ORIGINAL v -- checked against the oracle's bit routine for EVERY float whose rounded exponent is below the
    normal-half range (incl. the flush-to-zero region) and across the boundary."""
    from oracle import loader as L

    port = L.port()
    lo, hi = 96 << 23, (114 << 23) + 4096
    for start in range(lo, hi, 1 << 21):  # 2 M floats per step keeps the temporaries small
        bits = np.arange(start, min(start + (1 << 21), hi), dtype=np.uint32)
        want = np.empty(bits.size, dtype=np.uint16)
        port.uo_float_to_half(bits.view(np.float32).ctypes.data, want.ctypes.data, bits.size)
        b = bits + np.uint32(0x1000)
        sub = (np.floor(bits.view(np.float32).astype(np.float64) * 33554432.0).astype(np.uint32) + 1) >> 1
        e = b >> 23
        nrm = (((e.astype(np.int64) - 112) << 10) | ((b & 0x7FFFFF) >> 13)).astype(np.uint32)  # what v_cvt_pkrtz yields for normal halves
        got = np.where(b < (113 << 23), sub, nrm).astype(np.uint16)
        assert np.array_equal(got, want), hex(start)
    zero = np.zeros(1, np.float32)
    out = np.empty(1, np.uint16)
    port.uo_float_to_half(zero.ctypes.data, out.ctypes.data, 1)
    assert out[0] == 0


def test_idct_rewrites_preserve_every_bit():
    """idct_core.h evaluates jidctint.c with two rewrites: DESCALE's rounding constant is added to the even part's DC
    terms instead of to each output, and the row pass also carries 512 << 18 so that range_limit[(x + 128) & 1023]
    becomes  clamp(((sum >> 18) & 1023) - 384, 0, 255).  Emulated here in wrap-around int32 arithmetic (numpy) and
    compared with the oracle's plain restatement of libjpeg, for ordinary, large and garbage coefficients."""
    from oracle import loader as L

    K = dict(f0298=2446, f0390=3196, f0541=4433, f0765=6270, f0899=7373, f1175=9633, f1501=12299, f1847=15137, f1961=16069,
             f2053=16819, f2562=20995, f3072=25172)

    def w32(x):
        x = np.asarray(x, dtype=np.int64) & 0xFFFFFFFF
        return np.where(x >= 2 ** 31, x - 2 ** 32, x)

    def mul(a, c):
        return w32(a * c)

    def pass_1d(v, fudge):
        i = [v[..., k] for k in range(8)]
        z2, z3 = i[2], i[6]
        z1 = mul(w32(z2 + z3), K["f0541"])
        t2 = w32(z1 + mul(z3, -K["f1847"]))
        t3 = w32(z1 + mul(z2, K["f0765"]))
        z2, z3 = i[0], i[4]
        t0 = w32((w32(z2 + z3) << 13) + fudge)
        t1 = w32((w32(z2 - z3) << 13) + fudge)
        t10, t13, t11, t12 = w32(t0 + t3), w32(t0 - t3), w32(t1 + t2), w32(t1 - t2)
        t0, t1, t2, t3 = i[7], i[5], i[3], i[1]
        z1, z2, z3, z4 = w32(t0 + t3), w32(t1 + t2), w32(t0 + t2), w32(t1 + t3)
        z5 = mul(w32(z3 + z4), K["f1175"])
        t0, t1, t2, t3 = mul(t0, K["f0298"]), mul(t1, K["f2053"]), mul(t2, K["f3072"]), mul(t3, K["f1501"])
        z1, z2, z3, z4 = mul(z1, -K["f0899"]), mul(z2, -K["f2562"]), mul(z3, -K["f1961"]), mul(z4, -K["f0390"])
        z3, z4 = w32(z3 + z5), w32(z4 + z5)
        t0, t1, t2, t3 = w32(t0 + w32(z1 + z3)), w32(t1 + w32(z2 + z4)), w32(t2 + w32(z2 + z3)), w32(t3 + w32(z1 + z4))
        return np.stack([w32(t10 + t3), w32(t11 + t2), w32(t12 + t1), w32(t13 + t0), w32(t13 - t0), w32(t12 - t1), w32(t11 - t2),
                         w32(t10 - t3)], -1)

    def device_formula(block):  # block [..., row, col], dequantized
        ws = np.swapaxes(pass_1d(np.swapaxes(block, -1, -2), 1 << 10) >> 11, -1, -2)
        s = pass_1d(ws, (1 << 17) + (512 << 18))
        return np.clip((((s & 0xFFFFFFFF) >> 18) & 1023) - 384, 0, 255).astype(np.uint8)

    rng = np.random.default_rng(71)
    for lo, hi, q in ((-200, 200, 16), (-1024, 1024, 8), (-32768, 32768, 255), (-32768, 32768, 65535)):
        coef = rng.integers(lo, hi, (6, 40, 64), dtype=np.int64).astype(np.int16)
        qt = rng.integers(1, q + 1, 64).astype(np.uint16)
        want = L.idct_dequant_port(coef, qt)  # [6 * 8, 40 * 8]
        deq = w32(coef.astype(np.int64) * qt.astype(np.int64)).reshape(6, 40, 8, 8)
        got = device_formula(deq)  # [6, 40, 8, 8]
        got = got.transpose(0, 2, 1, 3).reshape(48, 320)
        assert np.array_equal(got, want), (lo, hi, q)
