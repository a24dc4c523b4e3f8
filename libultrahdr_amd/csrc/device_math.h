// Device-side building blocks shared by the gfx950 kernels of the gain-map hot path.
//
// Arithmetic contract: every kernel TU is compiled with -ffp-contract=off and without fast-math,
// so each float +,-,*,/ below is one IEEE-754 round-to-nearest operation in the order written --
// the same operations the reference's x86-64 (SSE2, no FMA) build performs.  Reference semantics
// are cited per function (paths under /root/reference).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uhdr {

struct Color3 {
  float r, g, b;
};

// ---- LUT index -------------------------------------------------------------------------------
// Reference: idx = int32( double( x * (N-1) ) + 0.5 ), then clip to [0, N-1]
// (lib/src/gainmapmath.cpp:127-129, 249-251, 321-323; lib/include/ultrahdr/gainmapmath.h:485-487).
// For N = 1024 and N = 65536 and every float x in [0, 1] the float expression
// int(x*(N-1) + 0.5f) is identical (checked exhaustively over all 1.06e9 floats, see
// tests/test_host_logic.py::test_lut_index_float_equivalence); N = 4096 has one exception
// (x*(N-1) == 0.49999997), so the 12-bit tables use the double form.
template <int N>
__device__ __forceinline__ int lut_index_f32(float x) {
  float f = x * (float)(N - 1);
  int i = (int)(f + 0.5f);
  return min(max(i, 0), N - 1);
}
template <int N>
__device__ __forceinline__ int lut_index_f64(float x) {
  float f = x * (float)(N - 1);
  int i = (int)((double)f + 0.5);
  return min(max(i, 0), N - 1);
}

// ---- clamps ----------------------------------------------------------------------------------
// clampPixelFloat (gainmapmath.h:561-563).  v_med3_f32; differs from the compare chain only for
// NaN / -0.0 inputs, neither of which survives to an output (see DESIGN.md "zero signs").
__device__ __forceinline__ float clamp01(float v) { return __builtin_amdgcn_fmed3f(v, 0.0f, 1.0f); }
// clampPixelFloatLinear (gainmapmath.h:570-574)
#define UHDR_MAX_LINEAR (10000.0f / 203.0f)
__device__ __forceinline__ float clamp_linear(float v) {
  return __builtin_amdgcn_fmed3f(v, 0.0f, UHDR_MAX_LINEAR);
}
// clipNegatives (gainmapmath.h): (v < 0) ? 0 : v.  One v_max_f32; it differs from the compare for -0.0 (which it turns into +0.0:
// every consumer adds a positive offset or multiplies by a positive factor next, where the zero's sign is lost) and NaN (no
// input produces one).
__device__ __forceinline__ float clip_neg(float v) { return __builtin_fmaxf(v, 0.0f); }

// ---- float -> half, bit-exact to floatToHalf (gainmapmath.h:160-173) ---------------------------
// That routine adds 0x1000 (round-half-up on the magnitude), then: normal halves keep
// (e-112)<<10 | m>>13, sub-normal halves (e in 102..112) shift the significand with another
// half-up rounding, smaller magnitudes flush to zero, e > 143 saturates to 0x7FFF.  For a
// non-negative input whose rounded exponent is in [113, 143] the whole thing is
// (bits + 0x1000 - (112 << 23)) >> 13: two integer ops.  Everything else takes the general form.
__device__ __forceinline__ uint32_t float_to_half_general(uint32_t bits) {
  const uint32_t b = bits + 0x00001000u;
  const int32_t e = (int32_t)((b & 0x7F800000u) >> 23);
  const uint32_t m = b & 0x007FFFFFu;
  uint32_t out = (b & 0x80000000u) >> 16;
  if (e > 112) out |= (((uint32_t)(e - 112) << 10) & 0x7C00u) | (m >> 13);
  if (e < 113 && e > 101) out |= (((0x007FF000u + m) >> (125 - e)) + 1u) >> 1;
  if (e > 143) out |= 0x7FFFu;
  return out;
}
__device__ __forceinline__ bool half_fast_ok(uint32_t bits) {
  // non-negative, rounded exponent in [113, 143]
  return (bits + 0x00001000u - (113u << 23)) < (31u << 23);
}
__device__ __forceinline__ uint32_t float_to_half_fast(uint32_t bits) {
  return (bits + 0x00001000u - (112u << 23)) >> 13;
}
// three channels -> RGBA-F16 pixel (colorToRgbaF16, gainmapmath.cpp:1286-1289); alpha = half(1.0)
__device__ __forceinline__ uint2 pack_rgba_f16(float r, float g, float b) {
  uint32_t br = __float_as_uint(r), bg = __float_as_uint(g), bb = __float_as_uint(b);
  uint32_t hr, hg, hb;
  bool fast = half_fast_ok(br) && half_fast_ok(bg) && half_fast_ok(bb);
  if (__builtin_amdgcn_ballot_w64(!fast) == 0) {  // wave-uniform: every lane on the 2-op path
    hr = float_to_half_fast(br);
    hg = float_to_half_fast(bg);
    hb = float_to_half_fast(bb);
  } else {
    hr = float_to_half_general(br);
    hg = float_to_half_general(bg);
    hb = float_to_half_general(bb);
  }
  uint2 o;
  o.x = hr | (hg << 16);
  o.y = hb | (0x3C00u << 16);
  return o;
}
// halfToFloat (gainmapmath.h:193-216) + sanitizePixel (gainmapmath.h:580-593)
__device__ __forceinline__ float half_to_float_ref(uint32_t h) {
  uint32_t mant = h & 0x3ffu, ex = (h >> 10) & 0x1fu, sign = (h >> 15) & 1u, o;
  if (ex == 0) {
    const uint32_t magic = 126u << 23;
    o = __float_as_uint(__uint_as_float(magic + mant) - __uint_as_float(magic));
  } else {
    o = mant << 13;
    o |= (ex == 0x1f) ? (255u << 23) : ((127u - 15u + ex) << 23);
  }
  o |= sign << 31;
  return __uint_as_float(o);
}
__device__ __forceinline__ float sanitize_linear(float v) {
  uint32_t a = __float_as_uint(v) & 0x7FFFFFFFu;
  if (a < 0x7F800000u) return (v < 0.0f) ? 0.0f : ((v > UHDR_MAX_LINEAR) ? UHDR_MAX_LINEAR : v);
  if (a == 0x7F800000u) return v > 0 ? UHDR_MAX_LINEAR : 0.0f;
  return 0.0f;
}

// colorToRgba1010102 (gainmapmath.cpp:1279-1284): x*1023 + 0.5, clip, truncate, alpha = 3
__device__ __forceinline__ uint32_t pack_rgba1010102(float r, float g, float b) {
  float fr = r * 1023.0f + 0.5f, fg = g * 1023.0f + 0.5f, fb = b * 1023.0f + 0.5f;
  uint32_t ri = (uint32_t)__builtin_amdgcn_fmed3f(fr, 0.0f, 1023.0f);
  uint32_t gi = (uint32_t)__builtin_amdgcn_fmed3f(fg, 0.0f, 1023.0f);
  uint32_t bi = (uint32_t)__builtin_amdgcn_fmed3f(fb, 0.0f, 1023.0f);
  return ri | (gi << 10) | (bi << 20) | (0x3u << 30);
}

// ---- colour transforms ---------------------------------------------------------------------------
struct Yuv2Rgb {  // coefficients built on the host exactly as the reference's static initialisers
  float cr, gcb, gcr, cb;
};
struct Rgb2Yuv {
  float yr, yg, yb, cb, cr;
  float rcb, rcr;  // 1.0f / cb, 1.0f / cr (correctly rounded, host-computed) for div_const
};

// a / b for a divisor that is a library constant: q0 = a * (1/b), one FMA residual, one FMA
// correction -- three full-rate instructions instead of the ~11 of an IEEE division.  For EVERY
// divisor the library uses this way (the chroma scale factors of the three gamuts, 1000, 10000, the
// squared tone-map headrooms) the result equals the correctly rounded quotient for every float
// mantissa: tests/test_exact_math.py::test_div_const_is_exact_for_every_library_constant runs all
// 2^23 of them per constant through this same source on the host (the sequence is scale invariant,
// so one binade covers all normal operands).  Only the sign of a zero quotient can differ (-0 / b
// gives +0), which no consumer observes.
#if defined(__HIPCC__)
#define UHDR_HD_MATH __host__ __device__ __forceinline__
#else
#define UHDR_HD_MATH inline
#endif
// a / b for ANY float divisor whose float64 reciprocal rbd = 1.0 / (double)b the host (or the caller,
// once per divisor) has computed: RN24((double)a * rbd) == RN24(a / b) for all normal a, b.  Proof: the
// exact quotient of two 24-bit significands is never closer than 2^-49 (relative) to a midpoint
// between adjacent floats (|A 2^k - M B| >= 1 for the integer significands A, B and an odd M), while
// the float64 product is within 2^-52 of the quotient (2^-53 from rbd, 2^-53 from the multiply).
// Three instructions: cvt, v_mul_f64, cvt.
UHDR_HD_MATH float div_by_rcp64(float a, double rbd) { return (float)((double)a * rbd); }

// The float64 reciprocal of a VARIABLE float divisor, good to 2^-52: v_rcp_f32 (1 ulp) refined by two
// Newton steps in float64 (the first residual 1 - b*r0 is exact: both factors are floats).  Worth it
// when several numerators share the divisor (the three channels of the Reinhard curve): 7 instructions
// once, then 3 per quotient through div_by_rcp64, against ~11 per IEEE division.
UHDR_HD_MATH double rcp64_of_f32(float b, float r0 /* ~1/b to within a few float ulps */) {
  const double bd = (double)b;
  double r = (double)r0;
  double e = fma(-bd, r, 1.0);
  r = fma(r, e, r);
  e = fma(-bd, r, 1.0);
  return fma(r, e, r);
}

UHDR_HD_MATH float div_const(float a, float b, float rb) {
  const float q0 = a * rb;
  const float r = __builtin_fmaf(-b, q0, a);
  return __builtin_fmaf(r, rb, q0);
}
// *YuvToRgb (gainmapmath.cpp:107-111, 177-181, 229-233)
__device__ __forceinline__ Color3 yuv_to_rgb(float y, float u, float v, const Yuv2Rgb& k) {
  Color3 o;
  o.r = clamp01(y + k.cr * v);
  o.g = clamp01(y - k.gcb * u - k.gcr * v);
  o.b = clamp01(y + k.cb * u);
  return o;
}
// *RgbToYuv (gainmapmath.cpp:96-99, 166-169, 196-199)
__device__ __forceinline__ Color3 rgb_to_yuv(Color3 e, const Rgb2Yuv& k) {
  float y = k.yr * e.r + k.yg * e.g + k.yb * e.b;
  Color3 o = {y, div_const(e.b - y, k.cb, k.rcb), div_const(e.r - y, k.cr, k.rcr)};
  return o;
}
// ConvertGamut / yuvColorGamutConversion (gainmapmath.cpp:617-621, 676-684): row . (r,g,b)
struct Mat3 {
  float m[9];
};
__device__ __forceinline__ Color3 mat3_apply(Color3 e, const Mat3& k) {
  Color3 o;
  o.r = k.m[0] * e.r + k.m[1] * e.g + k.m[2] * e.b;
  o.g = k.m[3] * e.r + k.m[4] * e.g + k.m[5] * e.b;
  o.b = k.m[6] * e.r + k.m[7] * e.g + k.m[8] * e.b;
  return o;
}
// yuvColorGamutConversion multiplies in the other operand order (e.y * c0 + ...); float multiply
// commutes, so mat3_apply gives identical bits.

// ---- wave reductions ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_min(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace uhdr
