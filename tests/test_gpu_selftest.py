"""Exhaustive on-device sweeps of the instruction-level shortcuts the encode kernels rely on (csrc/encode_core.h, round 4):
what this part's v_rcp_f32 and v_cvt_rpi_i32_f32 return cannot be established by a CPU test.  Every sweep runs inside the
library (csrc/selftest.hip) through the C ABI and reports counters; the assertions are here.
"""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A

pytestmark = pytest.mark.gpu


def _run(hip_ctx, which, arg0=0, arg1=0, seed=1, mm=None):
    out = (C.c_ulonglong * 8)()
    mmv = (C.c_float * 6)(*(mm if mm is not None else [0.0] * 6))
    A.check(hip_ctx.lib.uhdr_hip_selftest(hip_ctx.handle, which, arg0, arg1, seed, mmv, out))
    return [int(v) for v in out]


def test_round_half_up_conversion_is_exact_for_every_float(hip_ctx):
    # lut_index (gainmapmath.cpp:127-129, 249-251, 321-323: int(double(x * (N - 1)) + 0.5)) and ScaleTo8Bit (jpegr.cpp:1979-1983)
    bad, n = _run(hip_ctx, 0)[:2]
    assert n == 0x4B000000 + 1 and bad == 0


def test_refined_reciprocal_is_correctly_rounded_for_every_float(hip_ctx):
    bad, n, raw_bad = _run(hip_ctx, 1)[:3]
    assert n > 2_000_000_000 and bad == 0
    assert raw_bad > 0  # the bare instruction is NOT correctly rounded: the Newton step is what makes it so


@pytest.mark.parametrize("elo,ehi", [(97, 147), (67, 187)])  # exponents -30 .. 20 (the kernels' operands), -60 .. 60 (the proven range)
def test_markstein_quotient_equals_ieee_division(hip_ctx, elo, ehi):
    bad = pairs = 0
    for seed in range(1, 5):
        o = _run(hip_ctx, 2, elo, ehi, seed * 7919)
        bad += o[0]
        pairs += o[1]
    assert pairs >= 4 * 2048 * 256 * 4096 // 2 and bad == 0


def test_srgb_oetf_direct_table_every_float_of_the_unit_interval(hip_ctx):
    lds_vs_generic, new_vs_old, n = _run(hip_ctx, 3)[:3]
    assert n == 0x3F800000 + 1
    assert lds_vs_generic == 0
    # both evaluations are within 2^-47 of the true power, so they agree except where that lands within 2^-47 (relative) of a
    # float rounding boundary: a handful of arguments out of 10^9 at most
    assert new_vs_old <= 16


@pytest.mark.parametrize("mm", [
    [-2.0, -1.5, -1.0, 4.0, 5.0, 6.0],          # ordinary content
    [0.0, 0.0, 0.0, 0.1, 0.1, 0.1],             # a flat image: the epsilon guard's range (jpegr.cpp:981-985)
    [-14.3, -14.3, -14.3, 15.6, 15.6, 15.6],    # the clamp's whole range
    [-0.3333, 0.25, 1.0, 0.71, 2.3, 1.004],     # narrow ranges
])
def test_two_pass_step_table_equals_the_per_sample_evaluation(hip_ctx, mm):
    for ch in range(3):
        bad, n, entries, no_table = _run(hip_ctx, 4, ch, 3, 1, mm)[:4]
        assert no_table == 0, f"channel {ch}: no table"
        assert 0 < entries <= 1024 and n > 2 ** 21 and bad == 0, (ch, bad, n, entries)


def test_a_range_too_dense_for_a_table_is_flagged(hip_ctx):
    o = _run(hip_ctx, 4, 0, 1, 1, [1.0, 0, 0, 1.0 + 1e-4, 0, 0])
    assert o[3] == 1
