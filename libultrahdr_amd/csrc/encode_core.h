// Per-pixel building blocks shared by the encode kernels (tonemap.hip, generate_gainmap.hip,
// encode_fused.hip): the arithmetic of UltraHdr::toneMap's pixel loop (jpegr.cpp:1945-1983,
// 2147-2203) and of encodeGain / computeGain (gainmapmath.cpp:753-782), written once so that the
// fused API-0 kernel is the same sequence of IEEE operations as the three separate ones.
#pragma once
#include "exact_math.h"
#include "pixel_io.h"
#include "uhdr_types.h"

namespace uhdr {

__device__ __forceinline__ uint32_t put8(float v) {  // put*Pixel: *255, +0.5, clip, truncate (gainmapmath.cpp:538-596)
  v *= 255.0f;
  v += 0.5f;
  v = (v < 0.0f) ? 0.0f : ((v > 255.0f) ? 255.0f : v);
  return (uint32_t)v;
}

// HDR inverse OETF through the LDS copy of the linearisation table (for HLG the host folded
// hlgOotfApprox into it); has_lut == false: linear input, identityConversion
__device__ __forceinline__ Color3 linearise_hdr(Color3 g, const float* lut, bool has_lut, bool lut_4096) {
  if (!has_lut) return g;
  Color3 l;
  if (lut_4096) {  // HLG / PQ tables: exact double-form index
    l.r = lut[lut_index_f64<kInvOetfN>(g.r)];
    l.g = lut[lut_index_f64<kInvOetfN>(g.g)];
    l.b = lut[lut_index_f64<kInvOetfN>(g.b)];
  } else {
    l.r = lut[lut_index_f32<kSrgbN>(g.r)];
    l.g = lut[lut_index_f32<kSrgbN>(g.g)];
    l.b = lut[lut_index_f32<kSrgbN>(g.b)];
  }
  return l;
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
  if (t.tab)
    for (uint32_t i = tid; i < t.n; i += nthreads) dst[i] = t.tab[i];
}

// linear HDR rgb -> linear Display-P3 SDR rgb in [0, 1]: globalTonemap (jpegr.cpp:1951-1977), gamut conversion to P3, clamp
__device__ __forceinline__ Color3 tone_curve_linear(Color3 l, const ToneMapParams& p) {
  float c0 = l.r, c1 = l.g, c2 = l.b;
  const float hr = p.headroom;
  if (p.is_normalized) { c0 *= hr; c1 *= hr; c2 *= hr; }
  float mx = c0;
  if (c1 > mx) mx = c1;
  if (c2 > mx) mx = c2;
  float ms = 1.0f + div_const(mx, p.headroom_sq, p.headroom_sq_rcp);  // ReinhardMap: mx / (hr * hr), divisor is a per-transfer constant
  ms /= 1.0f + mx;
  ms = ms * mx;
  // c * max_sdr / max_hdr for the three channels (jpegr.cpp:1968-1972): one shared float64 reciprocal of
  // max_hdr, each quotient exact (device_math.h: rcp64_of_f32, div_by_rcp64); mx == 0 implies c <= 0
  const double rmx = rcp64_of_f32(mx, __builtin_amdgcn_rcpf(mx));
  Color3 o;
  o.r = c0 > 0.0f ? div_by_rcp64(c0 * ms, rmx) : 0.0f;
  o.g = c1 > 0.0f ? div_by_rcp64(c1 * ms, rmx) : 0.0f;
  o.b = c2 > 0.0f ? div_by_rcp64(c2 * ms, rmx) : 0.0f;
  if (p.gamut_on) o = mat3_apply(o, p.gamut);
  o.r = clamp01(o.r); o.g = clamp01(o.g); o.b = clamp01(o.b);
  return o;
}
// ... -> the three sRGB BYTES of putRgba8888Pixel (srgbOetf, * 255 + 0.5, clip, truncate): through the step table when
// the call has one (identical bytes: the table is built from the same srgb_oetf_table), else evaluated per channel
__device__ __forceinline__ void tone_curve_bytes(Color3 l, const ToneMapParams& p, const double* math, const uint2* srgb8,
                                                 uint32_t& r8, uint32_t& g8, uint32_t& b8) {
  const Color3 o = tone_curve_linear(l, p);
  if (p.srgb8.tab) {
    r8 = step_code(o.r, srgb8, p.srgb8);
    g8 = step_code(o.g, srgb8, p.srgb8);
    b8 = step_code(o.b, srgb8, p.srgb8);
  } else {
    r8 = put8(srgb_oetf_table(o.r, math));
    g8 = put8(srgb_oetf_table(o.g, math));
    b8 = put8(srgb_oetf_table(o.b, math));
  }
}

// linear HDR rgb -> gamma-encoded Display-P3 SDR rgb: globalTonemap (jpegr.cpp:1951-1977), gamut
// conversion to P3, clamp, srgbOetf (gainmapmath.cpp:139-148, pow through exact_math.h)
__device__ __forceinline__ Color3 tone_curve(Color3 l, const ToneMapParams& p, const double* math) {
  float c0 = l.r, c1 = l.g, c2 = l.b;
  const float hr = p.headroom;
  if (p.is_normalized) { c0 *= hr; c1 *= hr; c2 *= hr; }
  float mx = c0;
  if (c1 > mx) mx = c1;
  if (c2 > mx) mx = c2;
  float ms = 1.0f + div_const(mx, p.headroom_sq, p.headroom_sq_rcp);  // ReinhardMap: mx / (hr * hr), divisor is a per-transfer constant
  ms /= 1.0f + mx;
  ms = ms * mx;
  // c * max_sdr / max_hdr for the three channels (jpegr.cpp:1968-1972): one shared float64 reciprocal of
  // max_hdr, each quotient exact (device_math.h: rcp64_of_f32, div_by_rcp64); mx == 0 implies c <= 0
  const double rmx = rcp64_of_f32(mx, __builtin_amdgcn_rcpf(mx));
  Color3 o;
  o.r = c0 > 0.0f ? div_by_rcp64(c0 * ms, rmx) : 0.0f;
  o.g = c1 > 0.0f ? div_by_rcp64(c1 * ms, rmx) : 0.0f;
  o.b = c2 > 0.0f ? div_by_rcp64(c2 * ms, rmx) : 0.0f;
  if (p.gamut_on) o = mat3_apply(o, p.gamut);
  o.r = clamp01(o.r); o.g = clamp01(o.g); o.b = clamp01(o.b);
  Color3 og = {srgb_oetf_table(o.r, math), srgb_oetf_table(o.g, math), srgb_oetf_table(o.b, math)};
  return og;
}

// encodeGain (gainmapmath.cpp:758-771): log2 is the DOUBLE libm one in the reference build, the
// normalisation is double arithmetic narrowed to float, then powf, then truncation.
__device__ __forceinline__ uint8_t encode_gain(float y_sdr, float y_hdr, const GenParams& p, const double* T, const uint2* gain8 = nullptr) {
  float gain = 1.0f;
  if (y_sdr > 0.0f) gain = y_hdr / y_sdr;
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
// computeGain (gainmapmath.cpp:773-782)
__device__ __forceinline__ float compute_gain(float sdr, float hdr, const double* T) {
  float gain = (float)log2_table_f64((hdr + 1e-7f) / (sdr + 1e-7f), T);
  if (sdr < 2.f / 255.0f) gain = fminf(gain, 2.3f);
  return gain;
}

struct F3 {
  float a, b, c;
};

// The gain of one map pixel from the two LINEAR renditions (after gamut conversion and clipNegatives):
// jpegr.cpp:787-815 (one pass, writes the map bytes) / 900-928 (two pass, writes float log2 gains and
// folds them into the running per-channel min / max).
template <bool TWO_PASS>
__device__ __forceinline__ void gain_of_pixel(Color3 sl, Color3 hl, const GenParams& p, const double* math, uint32_t x, uint32_t y,
                                              float mn[3], float mx[3], const uint2* gain8 = nullptr) {
  if (p.multichannel) {
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
        v[c] = compute_gain(sn[c], hn[c], math);
        mn[c] = fminf(mn[c], v[c]);
        mx[c] = fmaxf(mx[c], v[c]);
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
      const float v = compute_gain(sy, hy, math);
      p.gain_log2[(size_t)y * p.map_w + x] = v;
      mn[0] = fminf(mn[0], v);
      mx[0] = fmaxf(mx[0], v);
    }
  }
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
