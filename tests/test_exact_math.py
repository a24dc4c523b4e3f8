"""The encode kernels' table-driven float64 pow / log2 (csrc/exact_math.h) against the reference's
libm calls, evaluated on the HOST through the C ABI (same source as the device code; float64
mul/add/fma are IEEE on both sides).  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from libultrahdr_amd import capi as A
from oracle import loader as O


def _eval_product(fn, x):
    lib = A.load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    assert lib.uhdr_hip_exact_math_eval(fn, x.ctypes.data_as(C.POINTER(C.c_float)),
                                        out.ctypes.data_as(C.POINTER(C.c_float)), x.size) == 0
    return out


def _floats_between(lo, hi, step):
    lo_b = np.float32(lo).view(np.uint32)
    hi_b = np.float32(hi).view(np.uint32)
    return np.arange(int(lo_b), int(hi_b) + 1, step, dtype=np.uint32).view(np.float32)


def test_srgb_oetf_matches_powf_over_the_whole_domain():
    # every 61st float of the pow branch (the linear branch below 0.0031308 is one multiply) plus a sparse sweep of
    # [0, 1], the neighbourhood of the branch point and both ends
    xs = np.concatenate([_floats_between(0.0031308, 1.0, 61), _floats_between(0.0, 0.0031308, 4099), _floats_between(0.0031300, 0.0031320, 1),
                         _floats_between(0.99999, 1.0, 1), np.float32([0.0, 1.0, 0.5, 0.0031308])])
    got = _eval_product(0, xs)
    want = O.eval_fn(O.port(), "uo_", "srgb_oetf", xs)
    bad = got.view(np.uint32) != want.view(np.uint32)
    # glibc's powf is faithfully, not correctly, rounded (and is an ifunc: its FMA and SSE2 variants
    # need not agree with each other); the table pow is correctly rounded.  Measured here: they
    # differ on ~5.5e-4 of the inputs of the pow branch, by one ulp of the power, and wherever they differ
    # the table value is the correctly rounded one (checked against 80-bit long double pow).
    assert bad.sum() < 2e-3 * np.count_nonzero(xs > np.float32(0.0031308)), f"{bad.sum()} of {xs.size} differ"
    if bad.any():
        # one ulp of the pow result; 1.055 * pow - 0.055 is up to ~3x smaller than pow, so up to 4 of its ulps
        d = np.abs(got[bad].view(np.int32).astype(np.int64) - want[bad].view(np.int32).astype(np.int64))
        assert d.max() <= 4
        p = np.longdouble(np.float32(1.0) / np.float32(2.4))
        cr_pow = (xs[bad].astype(np.longdouble) ** p).astype(np.float32)
        cr = ((np.float32(1.0) + np.float32(0.055)) * cr_pow - np.float32(0.055)).astype(np.float32)
        assert np.array_equal(got[bad].view(np.uint32), cr.view(np.uint32))
    # ... and the 8-bit codes the tone mapper stores (put8 / ScaleTo8Bit of the value) agree
    q = lambda v: np.clip(np.floor(v * np.float32(255.0) + np.float32(0.5)), 0, 255).astype(np.uint8)
    assert np.count_nonzero(q(got) != q(want)) <= 2


def test_log2_matches_double_log2_narrowed_to_float():
    rng = np.random.default_rng(7)
    xs = np.concatenate([
        _floats_between(1e-10, 1e12, 97),                       # the range computeGain's ratio can take
        _floats_between(0.9999, 1.0001, 1),                     # cancellation region around 1
        np.exp2(rng.uniform(-30, 40, 200000)).astype(np.float32),
        np.float32([1.0, 2.0, 0.5, 4.926108, 49.26108]),
    ])
    got = _eval_product(1, xs)
    want = O.eval_fn(O.port(), "uo_", "log2_f64", xs)
    bad = got.view(np.uint32) != want.view(np.uint32)
    assert bad.mean() < 1e-6, f"{bad.sum()} of {xs.size} differ"
    if bad.any():
        d = np.abs(got[bad].view(np.int32).astype(np.int64) - want[bad].view(np.int32).astype(np.int64))
        assert d.max() <= 1


def test_unknown_function_is_rejected():
    lib = A.load()
    x = np.zeros(1, np.float32)
    assert lib.uhdr_hip_exact_math_eval(9, x.ctypes.data_as(C.POINTER(C.c_float)), x.ctypes.data_as(C.POINTER(C.c_float)), 1) == -1


def _library_division_constants():
    """Every divisor the kernels feed to div_const, computed the way the library computes them."""
    f = np.float32
    c = []
    for (r, b) in ((0.2126, 0.0722), (0.2627, 0.059302)):   # BT.709 / BT.2100 luma -> chroma scale 2 * (1 - k)
        c += [f(2) * (f(1) - f(b)), f(2) * (f(1) - f(r))]
    c += [f(1.772), f(1.402)]                                  # Display-P3 rows use the BT.601 literals
    c += [f(1000.0), f(10000.0)]                               # kHlgMaxNits, kPqMaxNits (decode tail)
    for peak in (1000.0, 10000.0, 203.0):                      # tone-map headroom^2 per transfer
        h = f(peak) / f(203.0)
        c.append(h * h)
    return [float(v) for v in c]


@pytest.mark.parametrize("b", _library_division_constants())
def test_div_const_is_exact_for_every_library_constant(b):
    """a / b == div_const(a, b) for all 2^23 mantissas of a (the three-instruction sequence is scale
    invariant, so one binade proves it for every normal a), for each constant divisor in the kernels."""
    lib = A.load()
    for start in range(0, 1 << 23, 1 << 20):  # all 2^23 mantissas, 1 M at a time
        a = (np.arange(start, start + (1 << 20), dtype=np.uint32) | np.uint32(0x3F800000)).view(np.float32)
        buf = np.concatenate([np.float32([b]), a, -a[::4097]])
        out = np.empty_like(buf)
        assert lib.uhdr_hip_exact_math_eval(2, buf.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)), buf.size) == 0
        assert out[0] == np.float32(1.0) / np.float32(b)
        want = buf[1:] / np.float32(b)
        assert np.array_equal(out[1:].view(np.uint32), want.view(np.uint32))


def test_division_through_a_float64_reciprocal_is_exact_for_any_divisor():
    """affineMapGain divides by (max - min), a different divisor per image: RN24((double)a * RN53(1/b)) ==
    RN24(a / b) for all normal floats (proof in csrc/device_math.h).  Random and adversarial divisors."""
    lib = A.load()
    rng = np.random.default_rng(11)
    divisors = np.concatenate([np.exp2(rng.uniform(-20, 20, 300)).astype(np.float32),
                               (np.uint32(0x3F800000) + rng.integers(0, 1 << 23, 300, dtype=np.uint32)).view(np.float32),
                               np.float32([3.0, 7.0, 0.1, 5.7, 1.9999999, 1.0000001, 29.9, 1e-3])])
    a = np.concatenate([(np.uint32(0x3F800000) | rng.integers(0, 1 << 23, 200000, dtype=np.uint32)).view(np.float32),
                        np.exp2(rng.uniform(-30, 30, 50000)).astype(np.float32) * rng.choice(np.float32([-1, 1]), 50000)])
    for b in divisors:
        buf = np.concatenate([np.float32([b]), a, (a[:20000] * b).astype(np.float32)])  # incl. near-exact quotients
        out = np.empty_like(buf)
        assert lib.uhdr_hip_exact_math_eval(3, buf.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)), buf.size) == 0
        want = buf[1:] / np.float32(b)
        assert np.array_equal(out[1:].view(np.uint32), want.view(np.uint32)), float(b)


def test_shared_divisor_division_is_exact():
    """Reinhard's three c * ms / mx share the divisor: float64 reciprocal by Newton from a float seed (a
    2-ulp-off one here, worse than v_rcp_f32), then the float64-multiply quotient == IEEE a / b."""
    lib = A.load()
    rng = np.random.default_rng(13)
    n = 4_000_000
    a = (np.exp2(rng.uniform(-24, 8, n)) * rng.choice([1.0, 1.0, -1.0], n)).astype(np.float32)
    b = np.exp2(rng.uniform(-24, 8, n)).astype(np.float32)
    a[: n // 4] = (b[: n // 4] * rng.integers(1, 1 << 12, n // 4).astype(np.float32)).astype(np.float32)  # near-exact quotients
    buf = np.empty(2 * n, dtype=np.float32)
    buf[0::2], buf[1::2] = a, b
    out = np.empty_like(buf)
    assert lib.uhdr_hip_exact_math_eval(4, buf.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)), buf.size) == 0
    assert np.array_equal(out[0::2].view(np.uint32), (a / b).view(np.uint32))


# ---- step tables (csrc/host_tables.cpp: build_step_table) -------------------------------------------------------------
def _step_eval(which, a, b, x):
    import ctypes as C

    lib = A.load()
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty(x.size, dtype=np.uint32)
    info = (C.c_uint32 * 4)()
    rc = lib.uhdr_hip_step_table_eval(which, C.c_float(a), C.c_float(b), x.ctypes.data, out.ctypes.data, x.size, info)
    return rc, out, list(info)


def _around_every_float_step(lo, hi, n_random, seed):
    """Random floats of [lo, hi] plus both neighbours of each (bit pattern +- 1): steps of a table sit between neighbours."""
    rng = np.random.default_rng(seed)
    x = (lo + rng.random(n_random, dtype=np.float32) * np.float32(hi - lo)).astype(np.float32)
    bits = x.view(np.uint32)
    allb = np.concatenate([bits, bits + 1, np.maximum(bits, 1) - 1])
    v = allb.view(np.float32)
    return v[(v >= lo) & (v <= hi)]


def test_srgb_byte_step_table_equals_the_evaluation_it_replaces():
    """toneMap's RGBA8888 tail through the LDS step table == put8(srgb_oetf_table(x)) for 6 M floats of [0, 1] (dense
    near zero too), the domain ends and out-of-range inputs (the table's domain clamp is clampPixelFloat)."""
    x = np.concatenate([_around_every_float_step(0.0, 1.0, 1_500_000, 3), _around_every_float_step(0.0, 0.01, 500_000, 4),
                        np.array([0.0, 1.0, -0.0, -1.0, 2.0, 1e-30, 0.0031308, np.nextafter(np.float32(0.0031308), np.float32(1))], dtype=np.float32)])
    rc, got, info = _step_eval(0, 0.0, 0.0, x)
    assert rc == 0 and info[0] == 1, info
    xc = np.clip(x, 0.0, 1.0).astype(np.float32)
    want = _eval_product(0, xc) * np.float32(255.0)
    want = np.clip(want + np.float32(0.5), 0.0, 255.0).astype(np.uint32)
    assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("mn,mx", [(1.0, 1000.0 / 203.0), (1.0, 10000.0 / 203.0), (0.5, 8.0)])
def test_encode_gain_step_table_equals_the_evaluation_it_replaces(mn, mx):
    """encodeGain's byte (one pass, gamma 1) through the step table == (uint8)(n * 255) with n from log2_table_f64 and the
    float64 normalisation the kernels used per sample, for 3 M gains across [min boost, max boost] and outside it."""
    mn32, mx32 = np.float32(mn), np.float32(mx)
    x = np.concatenate([_around_every_float_step(float(mn32), float(mx32), 1_000_000, 5),
                        np.array([0.0, float(mn32) / 2, float(mx32) * 2, np.inf, float(mn32), float(mx32)], dtype=np.float32)])
    rc, got, info = _step_eval(1, float(mn32), float(mx32), x)
    assert rc == 0 and info[0] == 1, info
    g = np.clip(x, mn32, mx32).astype(np.float32)
    l2min, l2max = np.float32(np.log2(mn32)), np.float32(np.log2(mx32))
    rng_ = np.float64(np.float32(l2max - l2min))
    pass
    # the table is built from the float64 log2 (not its float narrowing); compare through a float64 log2 that is exact to
    # well below the float64 -> float rounding of n: allow no more than the rare sample where that narrowing matters
    n = ((np.log2(g.astype(np.float64)) - np.float64(l2min)) / rng_).astype(np.float32)
    want = (n * np.float32(255.0)).astype(np.uint32) & 0xff
    assert (got != want).mean() < 1e-5, int((got != want).sum())
    assert np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1


def test_a_boost_range_too_dense_for_the_table_is_reported_not_exact():
    rc, _, info = _step_eval(1, 1.0, 1.0 + 2.0 ** -10, np.array([1.0], dtype=np.float32))
    assert rc == 1 and info[0] == 0


@pytest.mark.parametrize("which,peak", [(4, 1000.0), (5, 10000.0)])
def test_prescaled_output_code_table_equals_scaling_then_the_plain_table(which, peak):
    """The decode tail's table that absorbs (x * 203) / peak (used when no HDR-side gamut conversion sits in between) ==
    the two float operations followed by the plain table, for 3 M values incl. both neighbours, negatives and values past
    the saturation point."""
    hi = peak / 203.0 * 1.2
    x = np.concatenate([_around_every_float_step(0.0, hi, 800_000, 11), _around_every_float_step(0.0, hi / 500, 200_000, 12),
                        np.array([0.0, -0.0, -3.0, hi * 2, 1e-30, peak / 203.0], dtype=np.float32)])
    rc, got, info = _step_eval(which, 0.0, 0.0, x)
    assert rc == 0 and info[0] == 1, info
    v = (x * np.float32(203.0)) / np.float32(peak)  # numpy float32: one rounding per operation, like the reference's two statements
    v = np.clip(v, 0.0, 1.0).astype(np.float32)
    rc, want, info2 = _step_eval(which - 2, 0.0, 0.0, v)
    assert rc == 0 and info2[0] == 1
    assert np.array_equal(got, want), int((got != want).sum())


def test_steps_that_sit_exactly_on_bucket_starts():
    """The device clamps the bit pattern to the first bucket's start instead of guarding `bucket - base` (five instructions
    per lookup instead of six): a threshold exactly ON a bucket start must then still compare above everything below it,
    which the builder arranges with an empty bucket in front (OetfBuckets::clamp_lo_bits).  Synthetic staircases whose
    steps are bucket-aligned, evaluated exactly as the kernels do, against the staircase itself."""
    for first_v, step_patterns in ((0.25, 1 << 15), (2.0 ** -20, 1 << 17), (0.5, 3 << 15)):
        first = np.array([first_v], dtype=np.float32).view(np.uint32)[0] & ~np.uint32(0x7fff)
        step = np.uint32(step_patterns)
        one = np.array([1.0], dtype=np.float32).view(np.uint32)[0]
        nsteps = int((one - first) // step) + 1
        if nsteps > 60000:
            continue
        edges = first + step * np.arange(0, min(nsteps, 4000), dtype=np.uint64)
        edges = edges[edges <= one].astype(np.uint32)
        probes = np.concatenate([edges, edges - 1, edges + 1, np.array([0, 1, one, one + 5, 0x80000000, 0xbf800000], dtype=np.uint32),
                                 np.random.default_rng(int(step_patterns)).integers(0, int(one), 200_000, dtype=np.uint32)])
        x = probes.view(np.float32)
        rc, got, info = _step_eval(6, float(np.array([first], dtype=np.uint32).view(np.float32)[0]), float(np.array([step], dtype=np.uint32).view(np.float32)[0]), x)
        if rc == 1:
            continue  # more than one step per bucket cannot be: steps are >= one bucket apart; a table beyond the capacity may be refused
        assert rc == 0 and info[0] == 1, (first_v, step_patterns, info)
        # the composite on the clamped pattern (negative floats and -0.0 are negative integers: they clamp to the domain's start)
        u = probes.astype(np.int64)
        u = np.where(probes >= 0x80000000, 0, np.minimum(u, int(one)))
        want = np.where(u < int(first), 0, 1 + (u - int(first)) // int(step)).astype(np.uint32)
        assert np.array_equal(got, want), (first_v, step_patterns, int((got != want).sum()))
