// Per-pixel building blocks shared by the encode kernels (tonemap.hip, generate_gainmap.hip,
// encode_fused.hip): the arithmetic of UltraHdr::toneMap's pixel loop (jpegr.cpp:1945-1983,
// 2147-2203) and of encodeGain / computeGain (gainmapmath.cpp:753-782), written once so that the
// fused API-0 kernel is the same sequence of IEEE operations as the three separate ones.
//
// Round 4: the same results from cheaper instruction sequences.  Measured issue cost on gfx950 (tools/ubench9.hip,
// relative to v_fma_f32 = 1): f32 add / mul / fma / and / lshr 1.0; med3, max, cvt, cmp, cndmask, bfe, lshl 1.7;
// f64 add / mul / fma and v_pk_*_f32 1.85 (packing buys nothing); rcp / log / exp 3.2; the compiler's IEEE division
// ~17, its float64 table pow ~30.  Hence:
//   * divisions are v_rcp_f32 + one Newton step (== RN(1/b) for EVERY float, swept on the device) + one Markstein
//     correction per quotient: 8 units for one quotient, 4 for each further one of the same divisor;
//   * LUT indices are v_cvt_rpi_i32_f32 (floor(x + 0.5) computed exactly: the reference's double-form index in one
//     instruction, swept on the device over every float);
//   * srgbOetf's pow reads a table indexed by the argument's own top bits (exact_math.h: pow_direct_f32);
//   * two-pass generation keeps the gain RATIO (hdr + eps) / (sdr + eps) between the passes instead of its log2: min and
//     max commute with the monotone log2, and pass 2's byte is a step function of the ratio (generate_gainmap.hip).
#pragma once
#include "exact_math.h"
#include "lds_copy.h"
#include "pixel_io.h"
#include "uhdr_types.h"

namespace uhdr {

__device__ __forceinline__ uint32_t put8(float v) {  // put*Pixel: *255, +0.5, clip, truncate (gainmapmath.cpp:538-596)
  v *= 255.0f;
  v += 0.5f;
  return (uint32_t)__builtin_amdgcn_fmed3f(v, 0.0f, 255.0f);  // finite arguments: the median IS the two-sided clip
}

// ---- exact float division without the compiler's range scaling ---------------------------------------------------------
// rcp_rn(b) == RN(1 / b) for every normal float b with a normal reciprocal: v_rcp_f32 is within 1 ulp, one Newton step
// r0 + r0 * (1 - b * r0) with the residual in an FMA lands on the correctly rounded value -- checked for all 2^32 bit
// patterns on the device (uhdr_hip_selftest, tests/test_gpu_selftest.py).  div_rn(a, b, r): Markstein's theorem -- with
// r == RN(1 / b) and q0 = RN(a * r) (within one ulp of a / b) the residual a - b * q0 is exact in an FMA and
// RN(q0 + residual * r) is the correctly rounded quotient.  Valid while nothing over- or underflows: |exponents| <= 60
// here (3.4e10 random pairs per range against the compiler's division in the same self-test), and the kernels' operands
// are nit values, pixel sums and table outputs between 2^-40 and 2^15.
__device__ __forceinline__ float rcp_rn(float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  return __builtin_fmaf(__builtin_fmaf(-b, r0, 1.0f), r0, r0);
}
__device__ __forceinline__ float div_rn(float a, float b, float r) {
  const float q0 = a * r;
  return __builtin_fmaf(__builtin_fmaf(-b, q0, a), r, q0);
}
__device__ __forceinline__ float div_rn(float a, float b) { return div_rn(a, b, rcp_rn(b)); }

// ---- LUT index -----------------------------------------------------------------------------------------------------------
// idx = int32(double(x * (N - 1)) + 0.5) (gainmapmath.cpp:127-129, 249-251, 321-323) for x in [0, 1]:
// v_cvt_rpi_i32_f32 computes floor(v + 0.5) without rounding the sum (swept over every float of [0, 2^23] on the
// device), i.e. exactly the double-form expression -- for every N, including the 4096-entry tables whose float-form
// index has an exception (device_math.h).
__device__ __forceinline__ int rpi(float v) {
  int i;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(i) : "v"(v));
  return i;
}
template <int N>
__device__ __forceinline__ int lut_index_rpi(float x) {  // x already clamped to [0, 1]
  return rpi(x * (float)(N - 1));
}

// HDR inverse OETF through the LDS copy of the linearisation table (for HLG the host folded
// hlgOotfApprox into it); has_lut == false: linear input, identityConversion.  The arguments are clamped to [0, 1]
// by every caller (yuv_to_rgb / the 10-bit unpack), so the index needs no clip.
__device__ __forceinline__ Color3 linearise_hdr(Color3 g, const float* lut, bool has_lut, bool lut_4096) {
  if (!has_lut) return g;
  Color3 l;
  if (lut_4096) {
    l.r = lut[min(max(lut_index_rpi<kInvOetfN>(g.r), 0), kInvOetfN - 1)];
    l.g = lut[min(max(lut_index_rpi<kInvOetfN>(g.g), 0), kInvOetfN - 1)];
    l.b = lut[min(max(lut_index_rpi<kInvOetfN>(g.b), 0), kInvOetfN - 1)];
  } else {
    l.r = lut[min(max(lut_index_rpi<kSrgbN>(g.r), 0), kSrgbN - 1)];
    l.g = lut[min(max(lut_index_rpi<kSrgbN>(g.g), 0), kSrgbN - 1)];
    l.b = lut[min(max(lut_index_rpi<kSrgbN>(g.b), 0), kSrgbN - 1)];
  }
  return l;
}
// the same for arguments known to lie in [0, 1] (the quad kernels: yuv_to_rgb's outputs)
template <int N>
__device__ __forceinline__ Color3 lut3_unit(Color3 g, const float* lut) {
  return Color3{lut[lut_index_rpi<N>(g.r)], lut[lut_index_rpi<N>(g.g)], lut[lut_index_rpi<N>(g.b)]};
}

// A monotone step function float -> small code as a bucket table in LDS (host_tables.cpp: build_step_table; the decode
// kernel's HLG / PQ tail is the same construction): clamp the bit pattern into the table's domain, bucket = bits >> shift,
// at most one threshold per bucket.  One 8-byte LDS read, one compare, one select.
__device__ __forceinline__ uint32_t step_code(float v, const uint2* tab, const StepTab& t) {
  uint32_t bits, addr;  // the clamp into the table's domain on the bit pattern: one three-operand median
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(bits) : "v"(__float_as_uint(v)), "v"(t.lo_bits), "v"(t.hi_bits));
  // tab is an LDS array: entry address = tab + (bucket - first bucket) * 8; t.lo_bits >= the first bucket's start (host_tables.cpp:
  // build_step_table), so the difference is never negative and the wave-uniform part folds into the add of a v_lshl_add
  const uint32_t rel = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)tab - t.base8;
  asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(addr) : "v"(bits >> (t.shm3 + 3u)), "s"(rel));
  typedef uint32_t lds_u2 __attribute__((ext_vector_type(2)));
  const lds_u2 e = *(const __attribute__((address_space(3))) lds_u2*)(uintptr_t)addr;  // (uintptr_t: the host pass of the compiler parses this too)
  uint32_t code;  // bits >= threshold ? upper : lower half of the entry's second word, picked by the select itself (SDWA)
  // s_nop 1: a VALU write of vcc needs two wait states before a VALU reads it as a mask (the compiler inserts the same)
  asm("v_cmp_ge_u32_e32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
      : "=v"(code)
      : "v"(bits), "v"(e.x), "v"(e.y)
      : "vcc");
  return code;
}
__device__ __forceinline__ void stage_step_tab(uint2* dst, const StepTab& t, uint32_t tid, uint32_t nthreads) {
  if (t.tab) copy_to_lds(dst, t.tab, t.n * 2u, tid, nthreads);  // (lds_copy.h: several 16-byte loads in flight per thread)
}
// the float64 pow table of srgbOetf (exact_math.h: pow_direct_f32), 16-byte copies
__device__ __forceinline__ void stage_pow_tab(double* dst, const double* math_tab, uint32_t tid, uint32_t nthreads) {
  copy_to_lds(dst, math_tab + kPowDirOff, (uint32_t)kPowDirN * 4u, tid, nthreads);
}

// linear HDR rgb -> linear Display-P3 SDR rgb in [0, 1]: globalTonemap (jpegr.cpp:1951-1977), gamut conversion to P3, clamp.
// The inputs are never negative (table outputs, sanitised half floats), so the reference's `x > 0 ? x * max_sdr / max_hdr
// : 0` is the plain quotient with the divisor kept away from zero: max_hdr == 0 means every channel is 0 and 0 / tiny == 0.
// GAMUT: 1 / 0 fix the gamut conversion at compile time (the quad kernels: no branch inside the pixel code), -1 reads p.gamut_on.
template <int GAMUT = -1>
__device__ __forceinline__ Color3 tone_curve_linear(Color3 l, const ToneMapParams& p) {
  const float hs = p.is_normalized ? p.headroom : 1.0f;  // wave-uniform; x * 1.0f == x
  const float c0 = l.r * hs, c1 = l.g * hs, c2 = l.b * hs;
  const float mx = __builtin_fmaxf(__builtin_fmaxf(c0, c1), c2);
  float ms = 1.0f + div_const(mx, p.headroom_sq, p.headroom_sq_rcp);  // ReinhardMap: mx / (hr * hr), divisor is a per-transfer constant
  const float den = 1.0f + mx;
  ms = div_rn(ms, den);
  ms = ms * mx;
  const float mxs = __builtin_fmaxf(mx, 0x1p-60f);
  const float rmx = rcp_rn(mxs);  // c * max_sdr / max_hdr for the three channels (jpegr.cpp:1968-1972): one shared reciprocal
  Color3 o;
  o.r = div_rn(c0 * ms, mxs, rmx);
  o.g = div_rn(c1 * ms, mxs, rmx);
  o.b = div_rn(c2 * ms, mxs, rmx);
  if (GAMUT > 0 || (GAMUT < 0 && p.gamut_on)) o = mat3_apply(o, p.gamut);
  o.r = clamp01(o.r); o.g = clamp01(o.g); o.b = clamp01(o.b);
  return o;
}
// srgbOetf of a value in [0, 1] with the direct pow table in LDS (exact_math.h: srgb_oetf_direct, same operations): the
// entry address comes straight from the argument's bits -- ((bits >> 16) - first) * 16 as (bits >> 12 & 0xffff0) plus a
// wave-uniform base.  Arguments of the linear segment (below 2^-9 in particular) wrap to an address beyond the workgroup's
// allocation, where LDS reads return 0 without faulting; whatever comes back is dropped by the segment select.
__device__ __forceinline__ float srgb_oetf_lds(float e, const double* powt_lds) {
  const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)powt_lds - (kPowDirFirst << 4);
  const uint32_t addr = ((__float_as_uint(e) >> 12) & 0xffff0u) + base;
  typedef double lds_d2 __attribute__((ext_vector_type(2)));
  const lds_d2 t = *(const __attribute__((address_space(3))) lds_d2*)(uintptr_t)addr;
  const double r = fma((double)e, t.x, -1.0);
  double q = fma(r, kPowDirC4, kPowDirC3);
  q = fma(r, q, kPowDirC2);
  q = fma(r, q, kPowDirC1);
  q = fma(r, q, 1.0);
  const float pw = (1.0f + 0.055f) * (float)(t.y * q) - 0.055f;
  return (e <= 0.0031308f) ? 12.92f * e : pw;
}
// ... -> the three sRGB BYTES of putRgba8888Pixel (srgbOetf, * 255 + 0.5, clip, truncate): through the step table when
// the call has one (identical bytes: the table is built from the same srgb_oetf_direct), else evaluated per channel
__device__ __forceinline__ void tone_curve_bytes(Color3 l, const ToneMapParams& p, const double* powt, const uint2* srgb8,
                                                 uint32_t& r8, uint32_t& g8, uint32_t& b8) {
  const Color3 o = tone_curve_linear(l, p);
  if (p.srgb8.tab) {
    r8 = step_code(o.r, srgb8, p.srgb8);
    g8 = step_code(o.g, srgb8, p.srgb8);
    b8 = step_code(o.b, srgb8, p.srgb8);
  } else {
    r8 = put8(srgb_oetf_direct(o.r, powt));
    g8 = put8(srgb_oetf_direct(o.g, powt));
    b8 = put8(srgb_oetf_direct(o.b, powt));
  }
}

// linear HDR rgb -> gamma-encoded Display-P3 SDR rgb: globalTonemap (jpegr.cpp:1951-1977), gamut
// conversion to P3, clamp, srgbOetf (gainmapmath.cpp:139-148, pow through exact_math.h)
template <int GAMUT = -1>
__device__ __forceinline__ Color3 tone_curve(Color3 l, const ToneMapParams& p, const double* powt_lds) {
  const Color3 o = tone_curve_linear<GAMUT>(l, p);
  return Color3{srgb_oetf_lds(o.r, powt_lds), srgb_oetf_lds(o.g, powt_lds), srgb_oetf_lds(o.b, powt_lds)};
}

// encodeGain (gainmapmath.cpp:758-771): log2 is the DOUBLE libm one in the reference build, the
// normalisation is double arithmetic narrowed to float, then powf, then truncation.  T: the float64 tables in GLOBAL
// memory -- only calls without a step table (gamma != 1, a boost range too dense for one) evaluate per pixel.
__device__ __forceinline__ uint8_t encode_gain(float y_sdr, float y_hdr, const GenParams& p, const double* T, const uint2* gain8 = nullptr) {
  float gain = 1.0f;
  if (y_sdr > 0.0f) gain = div_rn(y_hdr, y_sdr);
  // the clamp, log2, normalisation and truncation as one table lookup (gamma == 1; same bytes: the table is built from
  // log2_table_f64 / div_by_const_f64 on the host); its domain clamp IS the clamp to [min_boost, max_boost]
  if (gain8 && p.gain8.tab) return (uint8_t)step_code(gain, gain8, p.gain8);
  if (gain < p.min_boost) gain = p.min_boost;
  if (gain > p.max_boost) gain = p.max_boost;
  const double lg = log2_table_f64(gain, T);
  const float n = (float)div_by_const_f64(lg - (double)p.log2min, p.log2_range, p.log2_range_rcp);
  const float ng = (p.gamma == 1.0f) ? n : powf(n, p.gamma);  // powf(x, 1) == x exactly
  return (uint8_t)(ng * 255.0f);
}
// computeGain (gainmapmath.cpp:773-782) without its logarithm: the RATIO (hdr + 1e-7) / (sdr + 1e-7), with the dark-pixel
// cap `if (sdr < 2 / 255) gain = min(gain, 2.3f)` applied in the ratio domain: kGainCapRatio (p.gain_cap) is the largest
// float whose log2, narrowed to float, is <= 2.3f -- and it IS 2.3f there (the host checks) -- so that
// (float)log2(min(q, cap)) == min((float)log2(q), 2.3f) for every q.
__device__ __forceinline__ float gain_ratio(float sdr, float hdr, float cap) {
  const float q = div_rn(hdr + 1e-7f, sdr + 1e-7f);
  return sdr < 2.f / 255.0f ? __builtin_fminf(q, cap) : q;
}
// ... and the reference's float: (float)log2((double)ratio).  Used on the six extrema and by the table builder only.
__device__ __forceinline__ float gain_log2_of_ratio(float q, const double* T) { return (float)log2_table_f64(q, T); }

struct F3 {
  float a, b, c;
};

// The gain of one map pixel from the two LINEAR renditions (after gamut conversion and clipNegatives):
// jpegr.cpp:787-815 (one pass, writes the map bytes) / 900-928 (two pass: writes the float gain RATIOS and folds them
// into the running per-channel min / max -- see gain_ratio).
// MC: 1 / 0 fix multi- / single-channel at compile time, -1 reads p.multichannel.
template <bool TWO_PASS, int MC = -1>
__device__ __forceinline__ void gain_of_pixel(Color3 sl, Color3 hl, const GenParams& p, const double* math, uint32_t x, uint32_t y,
                                              float mn[3], float mx[3], const uint2* gain8 = nullptr) {
  if (MC > 0 || (MC < 0 && p.multichannel)) {
    const float sn[3] = {sl.r * 203.0f, sl.g * 203.0f, sl.b * 203.0f};
    const float hn[3] = {hl.r * p.hdr_nits, hl.g * p.hdr_nits, hl.b * p.hdr_nits};
    if constexpr (!TWO_PASS) {
      uint8_t* o = p.out + (size_t)y * p.out_stride * 3 + x * 3;
      o[0] = encode_gain(sn[0], hn[0], p, math, gain8);
      o[1] = encode_gain(sn[1], hn[1], p, math, gain8);
      o[2] = encode_gain(sn[2], hn[2], p, math, gain8);
    } else {
      float v[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        v[c] = gain_ratio(sn[c], hn[c], p.gain_cap);
        mn[c] = __builtin_fminf(mn[c], v[c]);
        mx[c] = __builtin_fmaxf(mx[c], v[c]);
      }
      *(F3*)(p.gain_log2 + ((size_t)y * p.map_w + x) * 3) = F3{v[0], v[1], v[2]};
    }
  } else {
    float sy, hy;
    if (p.use_luminance) {  // SDR-gamut luminance coefficients for BOTH images (jpegr.cpp:803-805)
      sy = (p.lum[0] * sl.r + p.lum[1] * sl.g + p.lum[2] * sl.b) * 203.0f;
      hy = (p.lum[0] * hl.r + p.lum[1] * hl.g + p.lum[2] * hl.b) * p.hdr_nits;
    } else {
      sy = fmaxf(sl.r, fmaxf(sl.g, sl.b)) * 203.0f;
      hy = fmaxf(hl.r, fmaxf(hl.g, hl.b)) * p.hdr_nits;
    }
    if constexpr (!TWO_PASS) {
      p.out[(size_t)y * p.out_stride + x] = encode_gain(sy, hy, p, math, gain8);
    } else {
      const float v = gain_ratio(sy, hy, p.gain_cap);
      p.gain_log2[(size_t)y * p.map_w + x] = v;
      mn[0] = __builtin_fminf(mn[0], v);
      mx[0] = __builtin_fmaxf(mx[0], v);
    }
  }
}

// The running extrema of the gain RATIOS start at +inf / 0, which no ratio attains; the conversion to the reference's log2
// extrema maps them to ITS initial values 127 / -128 (jpegr.cpp:846-847), so a stripe without samples still contributes
// the identity of the merge.
#define UHDR_RATIO_MIN_INIT __builtin_inff()
#define UHDR_RATIO_MAX_INIT 0.0f
__device__ __forceinline__ float log2_extremum(float q, bool is_min, const double* T) {
  if (is_min ? q == UHDR_RATIO_MIN_INIT : q == UHDR_RATIO_MAX_INIT) return is_min ? 127.0f : -128.0f;
  return gain_log2_of_ratio(q, T);
}

// per-workgroup min / max of the two-pass gains -> partials[blockIdx.x * 6 + {min r,g,b, max r,g,b}]
template <int BLOCK>
__device__ __forceinline__ void reduce_block_minmax(const float mn[3], const float mx[3], float* partials) {
  __shared__ float s_red[BLOCK / 64][6];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float a = wave_min(mn[c]), b = wave_max(mx[c]);
    if (lane == 0) { s_red[wv][c] = a; s_red[wv][3 + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s_red[0][threadIdx.x];
    for (int k = 1; k < BLOCK / 64; k++)
      v = threadIdx.x < 3 ? fminf(v, s_red[k][threadIdx.x]) : fmaxf(v, s_red[k][threadIdx.x]);
    partials[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
  }
}

}  // namespace uhdr
