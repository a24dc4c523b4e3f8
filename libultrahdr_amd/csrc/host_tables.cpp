// Host-side table builders (see host_tables.h).
//
// Float-vs-double bookkeeping.  In the reference (as built by GCC/libstdc++ for x86-64) the
// transfer functions in lib/src/gainmapmath.cpp call pow / log / exp / log2 / exp2 unqualified
// from inside namespace ultrahdr with no using-directive in scope, so the C library's DOUBLE
// overloads are selected and float arguments are promoted; results are narrowed when stored to a
// float.  std::pow(float,float) and explicit powf stay single precision.  The built object's
// import list (pow, log, exp, log2, exp2, powf, sqrtf) confirms this.  Every expression below is
// written with the promotions explicit so that any host compiler produces the same tables.
// This TU is compiled with -ffp-contract=off.
#include "host_tables.h"

#include "exact_math.h"

#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <mutex>

namespace uhdr {
namespace host {

namespace {

float srgb_inv_oetf(float e) {  // gainmapmath.cpp:114-120
  if (e <= 0.04045f) return e / 12.92f;
  return (float)std::pow((double)((e + 0.055f) / 1.055f), (double)2.4f);
}
const float kHlgA = 0.17883277f, kHlgB = 0.28466892f, kHlgC = 0.55991073f;
float hlg_oetf(float e) {  // gainmapmath.cpp:238-244
  if (e <= 1.0f / 12.0f) return sqrtf(3.0f * e);
  return (float)((double)kHlgA * std::log((double)(12.0f * e - kHlgB)) + (double)kHlgC);
}
float hlg_inv_oetf(float e) {  // gainmapmath.cpp:259-265 (pow(x, 2.0) is x*x in double)
  if (e <= 0.5f) return (float)(((double)e * (double)e) / (double)3.0f);
  return (float)((std::exp((double)((e - kHlgC) / kHlgA)) + (double)kHlgB) / (double)12.0f);
}
const float kPqM1 = 2610.0f / 16384.0f, kPqM2 = 2523.0f / 4096.0f * 128.0f;
const float kPqC1 = 3424.0f / 4096.0f, kPqC2 = 2413.0f / 4096.0f * 32.0f, kPqC3 = 2392.0f / 4096.0f * 32.0f;
float pq_oetf(float e) {  // gainmapmath.cpp:313-316
  if (e <= 0.0f) return 0.0f;
  const double pw = std::pow((double)e, (double)kPqM1);
  return (float)std::pow(((double)kPqC1 + (double)kPqC2 * pw) / (1 + (double)kPqC3 * pw), (double)kPqM2);
}
float pq_inv_oetf(float e) {  // gainmapmath.cpp:330-333
  const float val = (float)std::pow((double)e, (double)(1 / kPqM2));
  float num = val - kPqC1;
  if (!(num > 0.0f)) num = 0.0f;
  return (float)std::pow((double)(num / (kPqC2 - kPqC3 * val)), (double)(1 / kPqM1));
}

std::vector<float> make_lut(int n, float (*fn)(float)) {  // LookUpTable, gainmapmath.h:345-357
  std::vector<float> t((size_t)n);
  for (int i = 0; i < n; i++) t[(size_t)i] = fn((float)i / (float)(n - 1));
  return t;
}

const float kSrgbR = 0.212639f, kSrgbG = 0.715169f, kSrgbB = 0.072192f;
const float kP3R = 0.2289746f, kP3G = 0.6917385f, kP3B = 0.0792869f;
const float kP3YR = 0.299f, kP3YG = 0.587f, kP3YB = 0.114f, kP3Cb = 1.772f, kP3Cr = 1.402f;
const float kBt2100R = 0.2627f, kBt2100G = 0.677998f, kBt2100B = 0.059302f;

const float kBt709ToP3[9] = {0.822462f, 0.177537f, 0.000001f, 0.033194f, 0.966807f, -0.000001f, 0.017083f, 0.072398f, 0.91052f};
const float kBt709ToBt2100[9] = {0.627404f, 0.329282f, 0.043314f, 0.069097f, 0.919541f, 0.011362f, 0.016392f, 0.088013f, 0.895595f};
const float kP3ToBt709[9] = {1.22494f, -0.22494f, 0.0f, -0.042057f, 1.042057f, 0.0f, -0.019638f, -0.078636f, 1.098274f};
const float kP3ToBt2100[9] = {0.753833f, 0.198597f, 0.04757f, 0.045744f, 0.941777f, 0.012479f, -0.00121f, 0.017601f, 0.983608f};
const float kBt2100ToBt709[9] = {1.660491f, -0.587641f, -0.07285f, -0.124551f, 1.1329f, -0.008349f, -0.018151f, -0.100579f, 1.11873f};
const float kBt2100ToP3[9] = {1.343578f, -0.282179f, -0.061399f, -0.065298f, 1.075788f, -0.01049f, 0.002822f, -0.019598f, 1.016777f};

const float kYuv709To601[9] = {1.0f, 0.101579f, 0.196076f, 0.0f, 0.989854f, -0.110653f, 0.0f, -0.072453f, 0.983398f};
const float kYuv709To2100[9] = {1.0f, -0.016969f, 0.096312f, 0.0f, 0.995306f, -0.051192f, 0.0f, 0.011507f, 1.002637f};
const float kYuv601To709[9] = {1.0f, -0.118188f, -0.212685f, 0.0f, 1.018640f, 0.114618f, 0.0f, 0.075049f, 1.025327f};
const float kYuv601To2100[9] = {1.0f, -0.128245f, -0.115879, 0.0f, 1.010016f, 0.061592f, 0.0f, 0.086969f, 1.029350f};
const float kYuv2100To709[9] = {1.0f, 0.018149f, -0.095132f, 0.0f, 1.004123f, 0.051267f, 0.0f, -0.011524f, 0.996782f};
const float kYuv2100To601[9] = {1.0f, 0.117887f, 0.105521f, 0.0f, 0.995211f, -0.059549f, 0.0f, -0.084085f, 0.976518f};

void set_mat(Mat3* o, const float* m) {
  for (int i = 0; i < 9; i++) o->m[i] = m[i];
}

}  // namespace

#define UHDR_STATIC_LUT(name, n, fn)                 \
  const std::vector<float>& name() {                 \
    static const std::vector<float> t = make_lut(n, fn); \
    return t;                                        \
  }
UHDR_STATIC_LUT(srgb_inv_oetf_lut, kSrgbN, srgb_inv_oetf)
UHDR_STATIC_LUT(hlg_inv_oetf_lut, kInvOetfN, hlg_inv_oetf)
UHDR_STATIC_LUT(pq_inv_oetf_lut, kInvOetfN, pq_inv_oetf)
// hlgInvOetfLUT followed by hlgOotfApprox (gainmapmath.cpp:271-277, 293-295: std::pow(float, float) =
// powf): both encode loops (jpegr.cpp:768-775, 2160-2163) apply the OOTF to the table's output, so
// the composite is a function of the table index alone.
const std::vector<float>& hlg_inv_oetf_ootf_lut() {
  static const std::vector<float> t = [] {
    std::vector<float> v = hlg_inv_oetf_lut();
    for (float& x : v) x = powf(x, 1.2f);
    return v;
  }();
  return t;
}
UHDR_STATIC_LUT(hlg_oetf_lut, kOetfN, hlg_oetf)
UHDR_STATIC_LUT(pq_oetf_lut, kOetfN, pq_oetf)

// pqOetfLUT followed by colorToRgba1010102's quantisation (gainmapmath.cpp:320-326, 1279-1284), per table
// node: the PQ decode tail gathers the 10-bit code directly (65536 x uint16 = 128 KiB, two codes per float
// of the returned vector).  code = uint32(CLIP3(e * 1023.0f + 0.5f, 0, 1023)), the same float operations.
const std::vector<float>& pq_oetf_code_lut() {
  static const std::vector<float> t = [] {
    const std::vector<float>& lut = pq_oetf_lut();
    std::vector<uint16_t> codes(lut.size());
    for (size_t i = 0; i < lut.size(); i++) {
      float v = lut[i] * 1023.0f + 0.5f;
      v = v < 0.0f ? 0.0f : (v > 1023.0f ? 1023.0f : v);
      codes[i] = (uint16_t)(uint32_t)v;
    }
    std::vector<float> packed(lut.size() / 2);
    memcpy(packed.data(), codes.data(), codes.size() * sizeof(uint16_t));
    return packed;
  }();
  return t;
}

// ---- output-code threshold tables ----------------------------------------------------------------
// applyGainMap's HLG / PQ tail maps a clamped float v in [0,1] to a 10-bit code:
//   HLG: v -> powf(v, 1/1.2f) -> hlgOetfLUT (65536 nodes) -> uint(CLIP(e*1023 + 0.5f))
//   PQ :                          v -> pqOetfLUT  (65536 nodes) -> uint(CLIP(e*1023 + 0.5f))
// (jpegr.cpp:1775-1805, gainmapmath.cpp:248-254, 303-306, 320-326, 1279-1284).  The composite is a
// monotone step function with at most 1024 values, so it is fully described by the 1023 smallest
// inputs at which the code increases.  They are found here by bisection with the HOST's libm --
// the same powf the reference would call on this machine -- and the kernel then needs only an
// approximate code plus three compares, no per-pixel powf and no 256 KiB table gather.
static uint32_t oetf_code_host(int ct, float v) {
  float x = v;
  if (ct == UHDR_CT_HLG) x = powf(v, 1.0f / 1.2f);
  int idx = (int)((double)(x * (float)(kOetfN - 1)) + 0.5);
  idx = idx < 0 ? 0 : (idx > kOetfN - 1 ? kOetfN - 1 : idx);
  const float e = (ct == UHDR_CT_HLG ? hlg_oetf_lut() : pq_oetf_lut())[(size_t)idx];
  float q = e * 1023 + 0.5f;
  q = q < 0.0f ? 0.0f : (q > 1023.0f ? 1023.0f : q);
  return (uint32_t)q;
}
uint32_t oetf_code(int ct, float v) { return oetf_code_host(ct, v); }

static std::vector<float> make_thresholds(int ct) {
  std::vector<float> t((size_t)kOetfThrN, 2.0f);  // "never reached" = above the clamped range
  t[0] = 0.0f;
  auto bits = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  const uint32_t top = bits(1.0f);
  const uint32_t cmax = oetf_code_host(ct, 1.0f);
  for (uint32_t c = 1; c <= cmax && c < 1024; c++) {
    uint32_t lo = 0, hi = top;  // invariant: F(lo) < c <= F(hi); non-negative floats order like their bits
    if (oetf_code_host(ct, 0.0f) >= c) { t[c] = 0.0f; continue; }
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (oetf_code_host(ct, flt(mid)) >= c) hi = mid; else lo = mid;
    }
    t[c] = flt(hi);
  }
  return t;
}
// thresholds followed by the packed bucket entries (see apply_gainmap.hip::oetf_code):
//   E[k] = c_lo | needs_search << 15 | c_hi << 16,  c_lo = F(k << 18), c_hi = F((k + 1) << 18)
// The device estimates the code by linear interpolation between c_lo and c_hi on the low 18 bits and
// settles it with the two thresholds around the estimate, which is right whenever the estimate is
// within one code of the truth.  Both the estimate (monotone inside a bucket) and the truth (a step
// function) are checked here at every point where either can change -- the bucket's first and last
// float and, for every threshold inside it, the threshold and its predecessor; a bucket in which
// any probe fails gets the needs_search flag and the device uses the binary search there.
static uint32_t device_fast_code(const std::vector<float>& t, uint32_t c_lo, uint32_t c_hi, float v) {
  uint32_t bits;
  memcpy(&bits, &v, 4);
  const uint32_t est = c_lo + (((c_hi - c_lo) * (bits & 0x3ffffu) + 0x20000u) >> 18);  // rounded interpolation
  return est + (v >= t[est + 1] ? 1u : 0u) - (v < t[est] ? 1u : 0u);
}
static std::vector<float> make_threshold_block(int ct) {
  std::vector<float> t = make_thresholds(ct);
  auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  auto bits_of = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  const uint32_t one = bits_of(1.0f);
  auto code_at = [&](uint32_t u) -> uint32_t { return oetf_code_host(ct, flt(u < one ? u : one)); };
  std::vector<uint32_t> e((size_t)kOetfEstN);
  std::vector<std::vector<uint32_t>> probes((size_t)kOetfEstN);
  for (uint32_t c = 1; c < 1024; c++) {
    if (!(t[c] <= 1.0f)) break;
    const uint32_t u = bits_of(t[c]);
    probes[u >> 18].push_back(u);
    if (u > 0) probes[(u - 1) >> 18].push_back(u - 1);
  }
  for (uint32_t k = 0; k < (uint32_t)kOetfEstN; k++) {
    const uint32_t first = k << 18, last = first + 0x3ffffu;
    const uint32_t c_lo = code_at(first), c_hi = code_at(first + 0x40000u);
    bool ok = true;
    if (first <= one) {
      probes[k].push_back(first);
      probes[k].push_back(last < one ? last : one);
      for (uint32_t u : probes[k])
        if (u <= one && device_fast_code(t, c_lo, c_hi, flt(u)) != oetf_code_host(ct, flt(u))) ok = false;
    }
    e[k] = c_lo | (ok ? 0u : 0x8000u) | (c_hi << 16);
  }
  t.resize((size_t)kOetfTabFloats, 0.0f);
  memcpy(t.data() + kOetfThrN, e.data(), (size_t)kOetfEstN * sizeof(uint32_t));
  return t;
}
const std::vector<float>& oetf_code_thresholds(int ct) {
  static const std::vector<float> hlg = make_threshold_block(UHDR_CT_HLG);
  static const std::vector<float> pq = make_threshold_block(UHDR_CT_PQ);
  return ct == UHDR_CT_HLG ? hlg : pq;
}

// ---- output-code BUCKET tables (the quad kernel's form of the same step function) -----------------
// Non-negative floats order like their bit patterns, so cutting [0, 1] into buckets of 2^shift consecutive
// bit patterns (shift 15 for HLG, 16 for PQ: 0.2 - 0.4 % of the value) leaves at most ONE point per bucket at
// which the 10-bit code changes -- the thresholds are 0.47 % (HLG, around code 511) / 0.8 % (PQ) apart or
// more, and where the reference's 65536-node LUT index makes the code jump by several steps at once the
// jump sits at a single float.  Entry k describes bucket base + k:
//     .thr  = bit pattern of the threshold inside the bucket (0xFFFFFFFF: none)
//     .code = code below the threshold | code from the threshold on << 16
// and the device evaluates   code(v) = bits(v) >= thr ? hi : lo   -- one 8-byte LDS read, one compare, one
// select per channel.  Everything below the first threshold is bucket 0 (code 0).  The construction is
// CHECKED here: the table lookup is replayed against the composite at every threshold, its predecessor
// and both ends of every bucket; `exact` says whether all of them (and the one-threshold property) held.
// Generic form: any monotone step function code(v) of a float restricted to [lo, hi] (bit patterns lo_bits <= hi_bits
// of non-negative floats), given as a callable on the bit pattern.  Thresholds by bisection, buckets of 2^shift bit
// patterns, first bucket = the one holding the smallest threshold.
OetfBuckets build_step_table(const std::function<uint32_t(uint32_t)>& code_of_bits, uint32_t lo_bits, uint32_t hi_bits,
                             uint32_t shift, uint32_t capacity) {
  OetfBuckets b;
  b.shift = shift;
  b.lo_bits = lo_bits;
  b.hi_bits = hi_bits;
  b.exact = true;
  const uint32_t c_lo = code_of_bits(lo_bits), c_hi = code_of_bits(hi_bits);
  std::vector<uint32_t> thr;  // thr[i]: smallest bit pattern whose code is >= c_lo + 1 + i
  for (uint32_t c = c_lo + 1; c <= c_hi; c++) {
    uint32_t lo = lo_bits, hi = hi_bits;  // invariant: code(lo) < c <= code(hi)
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (code_of_bits(mid) >= c) hi = mid; else lo = mid;
    }
    thr.push_back(hi);
  }
  const uint32_t first = thr.empty() ? hi_bits : thr[0];
  b.base = first >> shift;
  // the device clamps the bit pattern to max(lo_bits, base << shift) instead of guarding the bucket subtraction: everything
  // below the first threshold must then still compare below it, so a threshold that sits exactly on the start of its
  // bucket gets an empty bucket in front of it
  if (b.base > 0 && (b.base << shift) == first && first > lo_bits) b.base--;
  b.clamp_lo_bits = std::max(lo_bits, b.base << shift);
  b.n = (hi_bits >> shift) - b.base + 1;
  if (b.n > capacity) { b.exact = false; b.n = 1; }
  b.entries.assign((size_t)b.n * 2, 0u);
  if (!b.exact) return b;
  std::vector<std::vector<uint32_t>> inside(b.n);
  for (uint32_t u : thr) {
    std::vector<uint32_t>& v = inside[(u >> shift) - b.base];
    if (v.empty() || v.back() != u) v.push_back(u);
  }
  auto clampb = [&](uint32_t u) { return u < lo_bits ? lo_bits : (u > hi_bits ? hi_bits : u); };
  auto lookup = [&](uint32_t u) -> uint32_t {  // the device's evaluation: clamp to [clamp_lo_bits, hi_bits], bucket - base
    u = u < b.clamp_lo_bits ? b.clamp_lo_bits : u;
    const uint32_t k = (u >> shift) - b.base;
    const uint32_t t = b.entries[2 * k], cc = b.entries[2 * k + 1];
    return u >= t ? cc >> 16 : cc & 0xffffu;
  };
  for (uint32_t k = 0; k < b.n; k++) {
    const uint32_t start = (b.base + k) << shift;
    const uint32_t lo = k == 0 ? c_lo : code_of_bits(clampb(start));  // bucket 0 also stands for everything below it
    uint32_t t = 0xFFFFFFFFu, hi = lo;
    if (!inside[k].empty()) {
      if (inside[k].size() > 1) b.exact = false;
      t = inside[k][0];
      hi = code_of_bits(inside[k].back());
    }
    b.entries[2 * k] = t;
    b.entries[2 * k + 1] = lo | (hi << 16);
  }
  // replay: both ends of every bucket, every threshold and its predecessor, and the ends of the domain
  for (uint32_t k = 0; k < b.n && b.exact; k++) {
    const uint32_t start = (b.base + k) << shift;
    uint32_t probes[6] = {start, start + (1u << shift) - 1, 0, 0, 0, 0};
    int np = 2;
    for (uint32_t u : inside[k]) {
      if (np < 6) probes[np++] = u;
      if (np < 6 && u > 0) probes[np++] = u - 1;
    }
    for (int i = 0; i < np; i++) {
      const uint32_t u = clampb(probes[i]);
      if (lookup(u) != code_of_bits(u)) b.exact = false;
    }
  }
  for (uint32_t u : {lo_bits, clampb(lo_bits + 1), clampb(first > 0 ? first - 1 : 0u), clampb(first), hi_bits})
    if (lookup(u) != code_of_bits(u)) b.exact = false;
  return b;
}
// prescaled: the table takes the value BEFORE the decode tail's nit scaling, (x * 203.0f) / peak -- two float roundings
// as written in jpegr.cpp:1775-1805 -- so that the kernel can skip those operations when no gamut conversion sits between
// them and the OETF (both are monotone, the composite stays a step function of x).
static OetfBuckets make_bucket_table(int ct, bool prescaled) {
  auto flt = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  auto bits_of = [](float f) { uint32_t u; memcpy(&u, &f, 4); return u; };
  const float peak = ct == UHDR_CT_HLG ? 1000.0f : 10000.0f;
  auto code = [&](uint32_t u) -> uint32_t {
    float v = flt(u);
    if (prescaled) {
      v = (v * 203.0f) / peak;
      v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    }
    return oetf_code_host(ct, v);
  };
  const uint32_t hi = prescaled ? bits_of(peak / 203.0f * 1.03125f) : bits_of(1.0f);  // beyond: the clamp, code stays
  OetfBuckets b = build_step_table(code, 0u, hi, ct == UHDR_CT_HLG ? kOetfBucketShiftHlg : kOetfBucketShiftPq,
                                   ct == UHDR_CT_HLG ? kOetfBucketsHlg : kOetfBucketsPq);
  if (prescaled && b.exact && code(hi) != code(bits_of(1.0e6f))) b.exact = false;  // the domain must reach the saturated code
  return b;
}
const OetfBuckets& oetf_code_buckets(int ct, bool prescaled) {
  static const OetfBuckets hlg = make_bucket_table(UHDR_CT_HLG, false), hlg_pre = make_bucket_table(UHDR_CT_HLG, true);
  static const OetfBuckets pq = make_bucket_table(UHDR_CT_PQ, false), pq_pre = make_bucket_table(UHDR_CT_PQ, true);
  return ct == UHDR_CT_HLG ? (prescaled ? hlg_pre : hlg) : (prescaled ? pq_pre : pq);
}

// ---- step tables of the ENCODE side ------------------------------------------------------------------------------------
// toneMap's last stage for RGBA8888 output: clampPixelFloat -> srgbOetf -> putRgba8888Pixel's byte
// (jpegr.cpp:1976-1977, gainmapmath.cpp:139-148, 538-552) is a monotone step function of the linear value with 255
// steps.  The composite is evaluated here through exact_math.h's srgb_oetf_direct -- the very function the kernels run
// per channel where no table applies (float64) -- so the table changes no result, only its cost.
const OetfBuckets& srgb_code8_buckets() {
  static const OetfBuckets t = [] {
    const std::vector<double>& T = math_tables();
    auto code = [&](uint32_t u) -> uint32_t {
      float x;
      memcpy(&x, &u, 4);
      float v = srgb_oetf_direct(x, T.data() + kPowDirOff) * 255.0f;
      v += 0.5f;
      v = (v < 0.0f) ? 0.0f : ((v > 255.0f) ? 255.0f : v);
      return (uint32_t)v;
    };
    uint32_t one;
    const float onef = 1.0f;
    memcpy(&one, &onef, 4);
    for (uint32_t sh : {16u, 15u, 14u}) {
      OetfBuckets b = build_step_table(code, 0u, one, sh, kStepTabMax);
      if (b.exact) return b;
    }
    OetfBuckets none;
    none.exact = false;
    return none;
  }();
  return t;
}
// computeGain's dark-pixel cap (gainmapmath.cpp:773-782: `if (sdr < 2 / 255) gain = min(gain, 2.3f)` on the float log2) in
// the RATIO domain, where two-pass generation keeps its samples since round 4: the largest float q with
// (float)log2(q) <= 2.3f.  min((float)log2(q), 2.3f) == (float)log2(min(q, cap)) for every q needs that bound to be
// ATTAINED, (float)log2(cap) == 2.3f -- it is (near 4.92 a float ulp moves log2 by 1.4e-7, less than the 2.4e-7 ulp of 2.3f);
// the function returns 0 otherwise and the host layer refuses two-pass generation rather than guess.
float gain_cap_ratio() {
  static const float cap = [] {
    const std::vector<double>& T = math_tables();
    auto lg = [&](uint32_t u) { float q; memcpy(&q, &u, 4); return (float)log2_table_f64(q, T.data()); };
    uint32_t lo = 0x40800000u, hi = 0x41000000u;  // 4.0 (log2 = 2 <= 2.3), 8.0 (3 > 2.3)
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (lg(mid) <= 2.3f) lo = mid; else hi = mid;
    }
    float q;
    memcpy(&q, &lo, 4);
    return lg(lo) == 2.3f ? q : 0.0f;
  }();
  return cap;
}
// the fused API-0 front end reads its own quantised SDR bytes back (generateGainMap would: getRgba8888Pixel's byte / 255.0f,
// then the sRGB inverse-OETF table lookup, gainmapmath.cpp:126-131): both steps per byte value
const std::vector<float>& srgb_inv_oetf_of_byte() {
  static const std::vector<float> t = [] {
    std::vector<float> v(256);
    const std::vector<float>& lut = srgb_inv_oetf_lut();
    for (int b = 0; b < 256; b++) {
      const float e = (float)b / 255.0f;
      int i = (int)((double)(e * (float)(kSrgbN - 1)) + 0.5);
      i = i < 0 ? 0 : (i > kSrgbN - 1 ? kSrgbN - 1 : i);
      v[b] = lut[i];
    }
    return v;
  }();
  return t;
}
// encodeGain (gainmapmath.cpp:758-771) after its clamp to [min_boost, max_boost], gamma == 1: the byte is a monotone
// step function of the clamped gain, evaluated here through the kernels' own log2_table_f64 / div_by_const_f64.
OetfBuckets gain_code8_buckets(float min_boost, float max_boost, float log2min, double log2_range, double log2_range_rcp) {
  const std::vector<double>& T = math_tables();
  auto code = [&](uint32_t u) -> uint32_t {
    float g;
    memcpy(&g, &u, 4);
    const double lg = log2_table_f64(g, T.data());
    const float n = (float)div_by_const_f64(lg - (double)log2min, log2_range, log2_range_rcp);
    return (uint32_t)(uint8_t)(n * 255.0f);
  };
  uint32_t lo, hi;
  memcpy(&lo, &min_boost, 4);
  memcpy(&hi, &max_boost, 4);
  OetfBuckets none;
  none.exact = false;
  if (!(min_boost > 0.0f) || !(max_boost > min_boost) || !std::isfinite(max_boost) || lo < 0x00800000u) return none;
  if (code(hi) < code(lo)) return none;
  for (uint32_t sh : {16u, 15u, 14u}) {
    OetfBuckets b = build_step_table(code, lo, hi, sh, kStepTabMax);
    if (b.exact) return b;
  }
  return none;
}
// RGBA1010102 HDR input: 10-bit code -> linear value = the 1010102 unpack (code / 1023.0f, gainmapmath.cpp:472-481)
// followed by the HDR inverse-OETF table lookup the kernels did per channel (lut_index of device_math.h), per code
std::vector<float> lin10_table(const float* lut, int n) {
  std::vector<float> t(1024);
  for (int c = 0; c < 1024; c++) {
    const float x = (float)c / 1023.0f;
    if (!lut) { t[c] = x; continue; }
    const float f = x * (float)(n - 1);
    int i = (n == kInvOetfN) ? (int)((double)f + 0.5) : (int)(f + 0.5f);
    i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    t[c] = lut[i];
  }
  return t;
}

// Tables of exact_math.h.  Everything is computed in long double (64-bit significand on x86-64)
// and rounded once to double.
const std::vector<double>& math_tables() {
  static const std::vector<double> tab = [] {
    std::vector<double> t((size_t)kMathTabDoublesAll, 0.0);
    const long double p = (long double)(1.0f / 2.4f);  // the float exponent srgbOetf passes to powf
    // the direct table of pow_direct_f32: one entry per 2^16 bit patterns of [2^-9, 1], {1 / c, c^p} at the bucket midpoint
    for (int i = 0; i < kPowDirN; i++) {
      const uint32_t bits = ((kPowDirFirst + (uint32_t)i) << kPowDirShift) | (1u << (kPowDirShift - 1));
      float cf;
      memcpy(&cf, &bits, 4);
      const long double c = (long double)cf;
      t[kPowDirOff + 2 * i] = (double)(1.0L / c);
      t[kPowDirOff + 2 * i + 1] = (double)powl(c, p);
    }
    for (int i = 0; i <= kPowM; i++) {
      const long double c = 1.0L + (long double)i / kPowM;
      t[kPowIcOff + 2 * i] = (double)(1.0L / c);
      t[kPowIcOff + 2 * i + 1] = (double)powl(c, p);
    }
    for (int k = kPowMinExp; k <= 0; k++) t[kPowScOff + (k - kPowMinExp)] = (double)exp2l((long double)k * p);
    long double binom = 1.0L;
    for (int j = 1; j <= 5; j++) {
      binom = binom * (p - (long double)(j - 1)) / (long double)j;  // C(p, j)
      t[kPowAOff + j - 1] = (double)binom;
    }
    for (int i = 0; i <= kLogM; i++) {
      const long double c = 1.0L + (long double)i / kLogM;
      t[kLogIcOff + 2 * i] = (double)(1.0L / c);
      t[kLogIcOff + 2 * i + 1] = (double)log2l(c);
    }
    const long double ln2 = logl(2.0L);
    for (int j = 1; j <= 6; j++) t[kLogBOff + j - 1] = (double)(((j & 1) ? 1.0L : -1.0L) / ((long double)j * ln2));
    return t;
  }();
  return tab;
}

Yuv2Rgb yuv2rgb_coeffs(int cg) {
  Yuv2Rgb k;
  if (cg == UHDR_CG_BT_709) {
    const float cb = 2 * (1 - kSrgbB), cr = 2 * (1 - kSrgbR);
    k.cb = cb; k.cr = cr;
    k.gcb = kSrgbB * cb / kSrgbG;
    k.gcr = kSrgbR * cr / kSrgbG;
  } else if (cg == UHDR_CG_DISPLAY_P3) {
    k.cb = kP3Cb; k.cr = kP3Cr;
    k.gcb = kP3YB * kP3Cb / kP3YG;
    k.gcr = kP3YR * kP3Cr / kP3YG;
  } else {
    const float cb = 2 * (1 - kBt2100B), cr = 2 * (1 - kBt2100R);
    k.cb = cb; k.cr = cr;
    k.gcb = kBt2100B * cb / kBt2100G;
    k.gcr = kBt2100R * cr / kBt2100G;
  }
  return k;
}
Rgb2Yuv rgb2yuv_coeffs(int cg) {
  Rgb2Yuv k;
  if (cg == UHDR_CG_BT_709) {
    k.yr = kSrgbR; k.yg = kSrgbG; k.yb = kSrgbB;
    k.cb = 2 * (1 - kSrgbB); k.cr = 2 * (1 - kSrgbR);
  } else if (cg == UHDR_CG_DISPLAY_P3) {
    k.yr = kP3YR; k.yg = kP3YG; k.yb = kP3YB; k.cb = kP3Cb; k.cr = kP3Cr;
  } else {
    k.yr = kBt2100R; k.yg = kBt2100G; k.yb = kBt2100B;
    k.cb = 2 * (1 - kBt2100B); k.cr = 2 * (1 - kBt2100R);
  }
  k.rcb = 1.0f / k.cb;
  k.rcr = 1.0f / k.cr;
  return k;
}
void luminance_coeffs(int cg, float out[3]) {
  if (cg == UHDR_CG_BT_709) { out[0] = kSrgbR; out[1] = kSrgbG; out[2] = kSrgbB; }
  else if (cg == UHDR_CG_DISPLAY_P3) { out[0] = kP3R; out[1] = kP3G; out[2] = kP3B; }
  else { out[0] = kBt2100R; out[1] = kBt2100G; out[2] = kBt2100B; }
}

bool gamut_matrix(int dst, int src, Mat3* out, bool* identity) {
  *identity = false;
  if (dst < 0 || dst > 2 || src < 0 || src > 2) return false;
  if (dst == src) { *identity = true; return true; }
  const float* m;
  if (dst == UHDR_CG_BT_709) m = src == UHDR_CG_DISPLAY_P3 ? kP3ToBt709 : kBt2100ToBt709;
  else if (dst == UHDR_CG_DISPLAY_P3) m = src == UHDR_CG_BT_709 ? kBt709ToP3 : kBt2100ToP3;
  else m = src == UHDR_CG_BT_709 ? kBt709ToBt2100 : kP3ToBt2100;
  set_mat(out, m);
  return true;
}

int yuv_encoding_matrix(int src, int dst, Mat3* out) {
  if (src < 0 || src > 2) return -1;
  if (dst < 0 || dst > 2) return -2;
  if (src == dst) return 1;
  const float* c;
  if (src == UHDR_CG_BT_709) c = dst == UHDR_CG_DISPLAY_P3 ? kYuv709To601 : kYuv709To2100;
  else if (src == UHDR_CG_DISPLAY_P3) c = dst == UHDR_CG_BT_709 ? kYuv601To709 : kYuv601To2100;
  else c = dst == UHDR_CG_BT_709 ? kYuv2100To709 : kYuv2100To601;
  set_mat(out, c);
  return 0;
}

float reference_peak_nits(int ct) {
  switch (ct) {
    case UHDR_CT_LINEAR: return 10000.0f;
    case UHDR_CT_HLG: return 1000.0f;
    case UHDR_CT_PQ: return 10000.0f;
    case UHDR_CT_SRGB: return 203.0f;
    default: return -1.0f;
  }
}

bool metadata_channels_identical(const uhdr_gainmap_metadata_t& m) {
  return m.max_content_boost[0] == m.max_content_boost[1] && m.max_content_boost[0] == m.max_content_boost[2] &&
         m.min_content_boost[0] == m.min_content_boost[1] && m.min_content_boost[0] == m.min_content_boost[2] &&
         m.gamma[0] == m.gamma[1] && m.gamma[0] == m.gamma[2] && m.offset_sdr[0] == m.offset_sdr[1] &&
         m.offset_sdr[0] == m.offset_sdr[2] && m.offset_hdr[0] == m.offset_hdr[1] &&
         m.offset_hdr[0] == m.offset_hdr[2];
}

float gainmap_weight(const uhdr_gainmap_metadata_t& m, float max_display_boost) {
  const float display_boost = max_display_boost < m.hdr_capacity_max ? max_display_boost : m.hdr_capacity_max;
  if (display_boost == m.hdr_capacity_max) return 1.0f;
  float w = (log2f(display_boost) - log2f(m.hdr_capacity_min)) /
            (log2f(m.hdr_capacity_max) - log2f(m.hdr_capacity_min));
  return (w < 0.0f) ? 0.0f : ((w > 1.0f) ? 1.0f : w);
}

void fill_idw(float* w, int s, int inc_r, int inc_b) {
  auto dist = [](float x1, float x2, float y1, float y2) {
    return sqrtf(((y2 - y1) * (y2 - y1)) + (x2 - x1) * (x2 - x1));
  };
  for (int y = 0; y < s; y++)
    for (int x = 0; x < s; x++) {
      const float px = ((float)x) / s, py = ((float)y) / s;
      const int cx = (int)floorf(px), cy = (int)floorf(py);
      const int nx = cx + inc_r, ny = cy + inc_b;
      float* o = w + y * s * 4 + x * 4;
      const float d1 = dist(px, (float)cx, py, (float)cy);
      if (d1 == 0) {
        o[0] = 1.f; o[1] = 0.f; o[2] = 0.f; o[3] = 0.f;
        continue;
      }
      const float w1 = 1.f / d1;
      const float w2 = 1.f / dist(px, (float)cx, py, (float)ny);
      const float w3 = 1.f / dist(px, (float)nx, py, (float)cy);
      const float w4 = 1.f / dist(px, (float)nx, py, (float)ny);
      const float tot = w1 + w2 + w3 + w4;
      o[0] = w1 / tot; o[1] = w2 / tot; o[2] = w3 / tot; o[3] = w4 / tot;
    }
}

void build_apply_tables(const uhdr_gainmap_metadata_t& m, float weight, int s, std::vector<float>* out) {
  out->assign((size_t)ApplyTables::floats(s), 0.0f);
  float* t = out->data();
  const std::vector<float>& srgb = srgb_inv_oetf_lut();
  for (int i = 0; i < kSrgbN; i++) t[ApplyTables::kSrgbOff + i] = srgb[(size_t)i];
  // GainLUT (gainmapmath.h:452-470): double log2/exp2, logBoost narrowed to float, float product
  // logBoost * weight fed to the double exp2.
  const bool single = metadata_channels_identical(m);
  for (int c = 0; c < 3; c++) {
    float* g = t + ApplyTables::kGainOff + c * kGainN;
    if (single && c > 0) {
      for (int i = 0; i < kGainN; i++) g[i] = t[ApplyTables::kGainOff + i];
      continue;
    }
    for (int i = 0; i < kGainN; i++) {
      const float value = (float)i / (float)(kGainN - 1);
      const float log_boost = (float)(std::log2((double)m.min_content_boost[c]) * (double)(1.0f - value) +
                                      std::log2((double)m.max_content_boost[c]) * (double)value);
      g[i] = (float)std::exp2((double)(log_boost * weight));
    }
  }
  for (int b = 0; b < 256; b++) t[ApplyTables::kU8fOff + b] = (float)b / 255.0f;  // mapUintToFloat
  // byte -> factor (scale-1 shortcut): getGainFactor (gainmapmath.h:483-489) applied to b/255
  for (int c = 0; c < 3; c++) {
    const int k = single ? 0 : c;
    const float gamma_inv = 1.0f / m.gamma[k];
    for (int b = 0; b < 256; b++) {
      float gain = (float)b / 255.0f;
      if (gamma_inv != 1.0f) gain = (float)std::pow((double)gain, (double)gamma_inv);
      int idx = (int)((double)(gain * (float)(kGainN - 1)) + 0.5);
      idx = idx < 0 ? 0 : (idx > kGainN - 1 ? kGainN - 1 : idx);
      t[ApplyTables::kFacOff + c * 256 + b] = t[ApplyTables::kGainOff + k * kGainN + idx];
    }
  }
  {  // the chroma products of p3YuvToRgb, one entry per chroma byte: the very operations the kernels used to do per quad
    const Yuv2Rgb yk = yuv2rgb_coeffs(UHDR_CG_DISPLAY_P3);
    const float k255 = 1 / 255.0f;
    for (int b = 0; b < 256; b++) {
      const float cf = (float)(b - 128) * k255;
      t[ApplyTables::kChromaVOff + 2 * b] = yk.cr * cf;
      t[ApplyTables::kChromaVOff + 2 * b + 1] = yk.gcr * cf;
      t[ApplyTables::kChromaUOff + 2 * b] = yk.gcb * cf;
      t[ApplyTables::kChromaUOff + 2 * b + 1] = yk.cb * cf;
    }
  }
  float* w = t + ApplyTables::kIdwOff;
  const size_t n = (size_t)s * s * 4;
  fill_idw(w, s, 1, 1);          // mWeights
  fill_idw(w + n, s, 0, 1);      // mWeightsNR
  fill_idw(w + 2 * n, s, 1, 0);  // mWeightsNB
  fill_idw(w + 3 * n, s, 0, 0);  // mWeightsC
  // ---- LDS images of the quad kernels (uhdr_types.h: ApplyTables) --------------------------------------------------------
  for (int i = 0; i < 2048; i++) t[ApplyTables::kSrgbPadOff + i] = srgb[(size_t)(i < kSrgbN ? i : kSrgbN - 1)];
  for (int b = 0; b < 256; b++) t[ApplyTables::kTapOff + 2 * b] = t[ApplyTables::kTapOff + 2 * b + 1] = t[ApplyTables::kU8fOff + b];
  if (s % 2 == 0) {
    // entry (table, oy, ox / 2) = {w0(ox), w0(ox+1), w1(ox), w1(ox+1), w2(ox), w2(ox+1), w3(ox), w3(ox+1)}
    float* pw = t + ApplyTables::idw_pair_off(s);
    const int nidw = ApplyTables::idw_floats(s);
    for (int i = 0; i < nidw; i++) {  // source index i = ((tbl * s + oy) * s + ox) * 4 + k
      const int k = i & 3, pos = i >> 2, ox = pos % s, row = pos / s;  // row = tbl * s + oy
      pw[(row * (s >> 1) + (ox >> 1)) * 8 + k * 2 + (ox & 1)] = w[i];
    }
  }
}

void jpeg_quant_table(int quality, int is_chroma, uint16_t qt[64]) {
  static const uint8_t kLuma[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
                                    14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                                    18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                                    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
  static const uint8_t kChroma[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                                      24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                      99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  const int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  const uint8_t* base = is_chroma ? kChroma : kLuma;
  for (int i = 0; i < 64; i++) {
    long v = ((long)base[i] * scale + 50L) / 100L;
    if (v <= 0L) v = 1L;
    if (v > 255L) v = 255L;
    qt[i] = (uint16_t)v;
  }
}

// ---- baseline Huffman tables (ITU-T T.81 Annex K.3; what jpeg_set_defaults installs, jpegencoderhelper.cpp:187) -------
namespace {
const uint8_t kZigzagToNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kBitsDcLuma[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kBitsDcChroma[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kValDc[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kBitsAcLuma[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kValAcLuma[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kBitsAcChroma[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kValAcChroma[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
}  // namespace

const uint8_t* jpeg_zigzag_to_natural() { return kZigzagToNatural; }

int jpeg_std_huff_table(int is_ac, int is_chroma, uint8_t bits[17], uint8_t vals[256]) {
  const uint8_t* b = is_ac ? (is_chroma ? kBitsAcChroma : kBitsAcLuma) : (is_chroma ? kBitsDcChroma : kBitsDcLuma);
  const uint8_t* v = is_ac ? (is_chroma ? kValAcChroma : kValAcLuma) : kValDc;
  int n = 0;
  for (int i = 0; i <= 16; i++) {
    bits[i] = b[i];
    if (i) n += b[i];
  }
  memset(vals, 0, 256);
  memcpy(vals, v, (size_t)n);
  return n;
}

// T.81 Annex C code assignment, packed for the device: entry = code length << 16 | code.
// Layout: [table 0 (luma) | table 1 (chroma)] x [16 DC entries (size category) | 256 AC entries (run << 4 | size)]
const std::vector<uint32_t>& jpeg_huff_code_tables() {
  static const std::vector<uint32_t> t = [] {
    std::vector<uint32_t> out((size_t)kHuffTabWords, 0u);
    for (int tbl = 0; tbl < 2; tbl++) {
      for (int ac = 0; ac < 2; ac++) {
        uint8_t bits[17], vals[256];
        jpeg_std_huff_table(ac, tbl, bits, vals);
        uint32_t code = 0;
        int k = 0;
        uint32_t* dst = out.data() + tbl * (16 + 256) + (ac ? 16 : 0);
        for (int l = 1; l <= 16; l++) {
          for (int i = 0; i < bits[l]; i++, k++) dst[vals[k]] = ((uint32_t)l << 16) | code++;
          code <<= 1;
        }
      }
    }
    return out;
  }();
  return t;
}

}  // namespace host
}  // namespace uhdr
