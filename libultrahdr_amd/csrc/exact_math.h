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
