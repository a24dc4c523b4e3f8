// Table-driven float64 evaluations of the two libm calls that sit on the per-pixel ENCODE path:
//
//   srgbOetf        std::pow(e, 1/2.4f)           -> powf   (gainmapmath.cpp:139-148)
//   encodeGain /    log2(gain)                     -> double log2, result narrowed to float
//   computeGain                                       (gainmapmath.cpp:758-782)
//
// glibc's powf / log2 are accurate to well under one unit in the last place of their result but
// are 100+ instruction sequences when translated for the GPU.  Both are evaluated here in float64
// from small tables built on the host (host_tables.cpp: math_tables):
//
//   x = 2^k * m,  m in [1,2),  c_i = 1 + i/M the table point nearest to m,  r = m/c_i - 1
//   x^p     = 2^(k p) * c_i^p * (1+r)^p        (1+r)^p by its binomial series, |r| <= 2^-7
//   log2 x  = k + log2 c_i + log2(1+r)         log2(1+r) by its Taylor series,  |r| <= 2^-8
//
// The float64 result is within ~1e-14 (relative to a float ulp: ~1e-6) of the true value, so after
// narrowing to float it is the correctly rounded result -- the value glibc returns except when
// glibc itself is not correctly rounded.  tests/test_exact_math.py runs the same functions on the
// host (they are __host__ __device__, float64 mul/add/fma are IEEE on both sides) against the
// reference's libm over every float of the domain that matters and counts the differences.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define UHDR_HD __host__ __device__ __forceinline__
#else
#define UHDR_HD inline
#endif

namespace uhdr {

// layout of the table block (doubles)
constexpr int kPowM = 64;                      // pow: intervals per binade
constexpr int kPowMinExp = -15;                // smallest binade covered (x >= 2^-15)
constexpr int kPowIcOff = 0;                   // {1/c_i, c_i^p} pairs, i = 0..kPowM
constexpr int kPowScOff = kPowIcOff + 2 * (kPowM + 1);   // 2^(k p), k = kPowMinExp..0
constexpr int kPowAOff = kPowScOff + (1 - kPowMinExp);   // C(p,1..5)
constexpr int kLogM = 128;                     // log2: intervals per binade
constexpr int kLogIcOff = kPowAOff + 5 + 1;    // {1/c_i, log2 c_i} pairs (16-byte aligned)
constexpr int kLogBOff = kLogIcOff + 2 * (kLogM + 1);    // (-1)^(j+1) / (j ln 2), j = 1..6
constexpr int kMathTabDoubles = kLogBOff + 6;
static_assert(kLogIcOff % 2 == 0, "pair tables are read as 16-byte vectors");

UHDR_HD uint32_t em_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
UHDR_HD float em_float(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// x^p for 2^kPowMinExp <= x <= 1 (p fixed by the table: (double)(1.0f / 2.4f)), narrowed to float
UHDR_HD float pow_table_f32(float x, const double* T) {
  const uint32_t b = em_bits(x);
  const uint32_t i = ((b & 0x7fffffu) + 0x10000u) >> 17;  // nearest c_i, 0..64
  const int k = (int)(b >> 23) - 127;
  const double m = (double)em_float((b & 0x7fffffu) | 0x3f800000u);
  const double r = fma(m, T[kPowIcOff + 2 * i], -1.0);
  const double* a = T + kPowAOff;
  double q = fma(r, a[4], a[3]);
  q = fma(r, q, a[2]);
  q = fma(r, q, a[1]);
  q = fma(r, q, a[0]);
  q = fma(r, q, 1.0);
  return (float)((T[kPowIcOff + 2 * i + 1] * T[kPowScOff + (k - kPowMinExp)]) * q);
}

// log2(x) in float64 for a positive normal float x
UHDR_HD double log2_table_f64(float x, const double* T) {
  const uint32_t b = em_bits(x);
  const uint32_t i = ((b & 0x7fffffu) + 0x8000u) >> 16;  // nearest c_i, 0..128 (c_0 = 1 and c_128 = 2 are exact)
  const int k = (int)(b >> 23) - 127;
  const double m = (double)em_float((b & 0x7fffffu) | 0x3f800000u);
  const double r = fma(m, T[kLogIcOff + 2 * i], -1.0);
  const double* c = T + kLogBOff;
  double q = fma(r, c[5], c[4]);
  q = fma(r, q, c[3]);
  q = fma(r, q, c[2]);
  q = fma(r, q, c[1]);
  q = fma(r, q, c[0]);
  return fma(r, q, (double)k + T[kLogIcOff + 2 * i + 1]);
}

// ---- x^p from a DIRECT table (round 4) ------------------------------------------------------------------------------
// The same identity with the table indexed by the top 16 bits of x itself (sign, exponent, 7 significand bits): one
// entry per 2^16 consecutive bit patterns of [2^-9, 1], holding {1 / c, c^p} for the bucket's MIDPOINT c (exponent
// included).  Then r = x / c - 1 is ONE float64 FMA on (double)x -- no significand extraction, no 2^(k p) table, no
// integer exponent -- with |r| <= 2^-8, and (1 + r)^p needs four more: degree 4 with the degree-5 term of the binomial
// series folded into the r and r^3 coefficients by Chebyshev economisation on [-2^-8, 2^-8] (r^5 ~ (20 a^2 r^3 -
// 5 a^4 r) / 16), truncation error 2^-49 relative -- the accuracy of pow_table_f32 (degree 5 on |r| <= 2^-7: 2^-47.5)
// at 6 float64 operations instead of 9 and without the integer work.  srgbOetf only raises arguments in
// (0.0031308, 1], which is why the table starts at 2^-9.
constexpr int kPowDirShift = 16;
constexpr uint32_t kPowDirFirst = 0x3B000000u >> kPowDirShift;          // bits(2^-9) >> 16
constexpr int kPowDirN = (int)((0x3F800000u >> kPowDirShift) - kPowDirFirst) + 1;  // 1153: the last bucket holds x == 1 only
constexpr int kPowDirDoubles = 2 * kPowDirN;
constexpr int kPowDirOff = (kMathTabDoubles + 1) & ~1;         // the direct table follows the round-1 tables in the device block
constexpr int kMathTabDoublesAll = kPowDirOff + kPowDirDoubles;
// C(p, j) for p = (double)(1.0f / 2.4f) = 0.4166666567325592041015625, economised (host_tables.cpp: pow_direct_table
// recomputes them in long double and tests/test_exact_math.py compares)
constexpr double kPowDirC1 = 0.4166666567303992;
constexpr double kPowDirC2 = -0.12152777694993544;
constexpr double kPowDirC3 = 0.06414022669132594;
constexpr double kPowDirC4 = -0.04142353087261225;

// x^p for 2^-9 <= x <= 1, narrowed to float; tab = kPowDirN pairs {1 / c, c^p}
UHDR_HD float pow_direct_f32(float x, const double* tab) {
  const uint32_t k = em_bits(x) >> kPowDirShift;
  const uint32_t i = k > kPowDirFirst ? k - kPowDirFirst : 0u;  // a saturating subtract: arguments below the table read entry 0
  const double r = fma((double)x, tab[2 * i], -1.0);
  double q = fma(r, kPowDirC4, kPowDirC3);
  q = fma(r, q, kPowDirC2);
  q = fma(r, q, kPowDirC1);
  q = fma(r, q, 1.0);
  return (float)(tab[2 * i + 1] * q);
}
// srgbOetf (gainmapmath.cpp:139-148) through the direct table, branch-free: both segments are evaluated (arguments of the
// linear segment read table entry 0 and the value is dropped by the select).  e in [0, 1].
UHDR_HD float srgb_oetf_direct(float e, const double* tab) {
  const float lin = 12.92f * e;
  const float pw = (1.0f + 0.055f) * pow_direct_f32(e, tab) - 0.055f;
  return (e <= 0.0031308f) ? lin : pw;
}

// srgbOetf (gainmapmath.cpp:139-148) with the table pow
UHDR_HD float srgb_oetf_table(float e, const double* T) {
  if (e <= 0.0031308f) return 12.92f * e;
  return (1.0f + 0.055f) * pow_table_f32(e, T) - 0.055f;
}

// a / b in float64 with b's reciprocal precomputed on the host: one Newton correction makes the
// quotient correctly rounded (up to the last bit in rare cases -- far below float resolution)
UHDR_HD double div_by_const_f64(double a, double b, double rb) {
  const double q0 = a * rb;
  const double rem = fma(-b, q0, a);
  return fma(rem, rb, q0);
}

}  // namespace uhdr
