// applyGainMap on gfx950: SDR base image + gain map + metadata -> HDR pixels.
// Reference loop: /root/reference/lib/src/jpegr.cpp:1714-1812 (UltraHdr::applyGainMap).
//
// Two kernels:
//   apply_quad_kernel   -- the hot one.  YCbCr 4:2:0 base (what a JPEG base image decodes to), or 4:2:2 / 4:4:4 /
//                          RGBA8888 (camera / API-0 streams); one work item of a lane = one 2x2 luma quad, a lane
//                          owns two quads 128 pixels apart, a wave a 256 x 2 pixel strip, so every global
//                          store instruction of the wave writes one contiguous 1 KiB (F16) / 512 B
//                          (1010102) run and the two runs of a row follow each other.  All per-call
//                          tables (sRGB-EOTF, gain LUT / byte->factor, byte->float, IDW weights, the HLG
//                          output-code thresholds) are staged in LDS once per workgroup; exactly the
//                          resident workgroups are launched and each wave walks its column strip.
//   apply_generic_kernel-- one thread per pixel, every other format / scale combination the reference
//                          accepts (RGB888 base, odd sizes, odd or non-integer scale, gamma != 1 at scale > 1).
// HBM-bound by design: 1.5 B (4:2:0) + map + 8 B (F16) per pixel, no intermediate buffers.
#include <cstdlib>
#include "lds_copy.h"
#include <type_traits>

#include "idct_core.h"
#include "uhdr_types.h"

namespace uhdr {

namespace {

constexpr int kBlock = 256;

// ---------------------------------------------------------------------------------------------
// shared tail: linear SDR rgb (after the optional SDR-side gamut conversion) x gain factors ->
// packed output pixel.  jpegr.cpp:1765-1805.
// ---------------------------------------------------------------------------------------------
template <int OUT>  // 0 linear F16, 1 HLG 1010102, 2 PQ 1010102
struct OutPix {
  using type = uint32_t;
};
template <>
struct OutPix<0> {
  using type = uint2;
};

// 10-bit output code of the HLG tail for a clamped v in [0,1]:
//   code(v) = max{ c : T[c] <= v }   with the host-built thresholds T (host_tables.cpp), T[0] = 0.
// T layout: [0, kOetfThrN) float thresholds (entries above the last reachable code are 2.0f), then
// kOetfEstN packed bucket entries E[k] = c_lo | needs_search << 15 | c_hi << 16 for bucket
// k = bits(v) >> 18 (4065 buckets of <= 3 % relative width cover [0,1]); c_lo / c_hi are the codes at
// the bucket's ends.  Linear interpolation on the low 18 bits estimates the code; the two
// thresholds around the estimate settle it.  The host has PROVEN, bucket by bucket, that this gives
// the exact code (the estimate never strays more than one step; see make_threshold_block) and flags
// the buckets where it does not -- there a 10-step binary search over T runs instead.  Either way
// the result is what the reference computes with the host libm: no per-pixel powf, no 256 KiB gather.
__device__ __forceinline__ uint32_t oetf_code_search(float v, const float* T) {
  uint32_t lo = 0;
#pragma unroll
  for (uint32_t step = 512; step; step >>= 1)
    if (v >= T[lo + step]) lo += step;  // lo + step <= 1023
  return lo;
}
template <int OUT>
__device__ __forceinline__ uint32_t oetf_code(float v, const float* T) {
  const uint32_t bits = __float_as_uint(v);
  const uint32_t e = ((const uint32_t*)(T + kOetfThrN))[bits >> 18];
  const uint32_t c_lo = e & 0x7fffu, c_hi = e >> 16;
  const uint32_t est = c_lo + ((__umul24(c_hi - c_lo, bits & 0x3ffffu) + 0x20000u) >> 18);  // rounded; 10-bit x 18-bit is exact in the 24-bit multiplier
  const float t0 = T[est], t1 = T[est + 1];
  uint32_t code = est + (v >= t1 ? 1u : 0u) - (v < t0 ? 1u : 0u);
  if (__builtin_amdgcn_ballot_w64((e & 0x8000u) != 0) != 0) {  // some lane sits in a flagged bucket
    if (e & 0x8000u) code = oetf_code_search(v, T);
  }
  return code;
}
__device__ __forceinline__ uint32_t pack_codes_1010102(uint32_t r, uint32_t g, uint32_t b) {
  return r | (g << 10) | (b << 20) | (0x3u << 30);
}

// Quad kernel: the same step function (HLG: clamp, powf(v, 1/1.2), OETF LUT, 10-bit quantisation; PQ: clamp, OETF LUT,
// quantisation -- jpegr.cpp:1775-1805) as a bucket table in LDS, built and verified on the host
// (host_tables.cpp: make_bucket_table).  Non-negative floats order like their bit patterns, so the clamp to [0, 1]
// is a three-operand integer median (negative values and -0.0 have the sign bit set: negative as integers -> 0),
// the bucket is bits >> shift, and inside a bucket the code changes at most once:
//     code = bits >= thr ? hi : lo          entry = {thr, lo | hi << 16}
// One 8-byte LDS read, one compare and one select per channel (the select picks the upper or lower 16 bits of the entry's
// second word directly: v_cndmask_b32_sdwa); no powf, no 256 KiB gather, no search.
template <int OUT>
__device__ __forceinline__ uint32_t oetf_code_bucket(float v, uint32_t tab_rel, uint32_t lo_bits, uint32_t hi_bits) {
  constexpr int SH = (OUT == 1) ? kOetfBucketShiftHlg : kOetfBucketShiftPq;
  // clampPixelFloat on the bit pattern (v_med3_i32; negative values and -0.0 are negative integers): lo_bits = the start of
  // the table's first bucket (everything below the first threshold compares below it there too -- host_tables.cpp:
  // build_step_table), hi_bits = 1.0f, or the saturation point of a prescaled table
  uint32_t bits, addr;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(bits) : "v"(__float_as_uint(v)), "v"(lo_bits), "v"(hi_bits));
  // LDS address of the entry = table + (bucket - first bucket) * 8: the wave-uniform part (tab_rel) is the add of a v_lshl_add
  asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(addr) : "v"(bits >> SH), "s"(tab_rel));
  typedef uint32_t lds_u2 __attribute__((ext_vector_type(2)));
  const lds_u2 e = *(const __attribute__((address_space(3))) lds_u2*)(uintptr_t)addr;  // (uintptr_t: the host pass of the compiler parses this too)
  uint32_t code;
  // s_nop 1: a VALU write of vcc needs two wait states before a VALU reads it as a mask (the compiler inserts the same)
  asm("v_cmp_ge_u32_e32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32_sdwa %0, %3, %3, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
      : "=v"(code)
      : "v"(bits), "v"(e.x), "v"(e.y)
      : "vcc");
  return code;
}

template <int OUT>
__device__ __forceinline__ typename OutPix<OUT>::type finish_pixel(Color3 lin, float f0, float f1,
                                                                   float f2, const ApplyParams& p,
                                                                   int nch) {
  // applyGainLUT: ((e + offset_sdr) * factor) - offset_hdr   (gainmapmath.cpp:807-810, 848-855)
  // single-channel maps use channel-0 metadata for all three colours.
  Color3 h;
  if (nch == 1) {
    h.r = ((lin.r + p.offset_sdr[0]) * f0) - p.offset_hdr[0];
    h.g = ((lin.g + p.offset_sdr[0]) * f0) - p.offset_hdr[0];
    h.b = ((lin.b + p.offset_sdr[0]) * f0) - p.offset_hdr[0];
  } else {
    h.r = ((lin.r + p.offset_sdr[0]) * f0) - p.offset_hdr[0];
    h.g = ((lin.g + p.offset_sdr[1]) * f1) - p.offset_hdr[1];
    h.b = ((lin.b + p.offset_sdr[2]) * f2) - p.offset_hdr[2];
  }
  if constexpr (OUT == 0) {
    if (p.hdr_gamut_on) h = mat3_apply(h, p.gamut);
    return pack_rgba_f16(clamp_linear(h.r), clamp_linear(h.g), clamp_linear(h.b));
  } else {
    const float peak = (OUT == 1) ? 1000.0f : 10000.0f;  // kHlgMaxNits / kPqMaxNits
    h.r = div_const(h.r * 203.0f, peak, 1.0f / peak);     // two roundings, as written in the reference
    h.g = div_const(h.g * 203.0f, peak, 1.0f / peak);
    h.b = div_const(h.b * 203.0f, peak, 1.0f / peak);
    if (p.hdr_gamut_on) h = mat3_apply(h, p.gamut);
    if constexpr (OUT == 1) {
      // clampPixelFloat, hlgInverseOotfApprox (powf), OETF LUT, colorToRgba1010102: one threshold lookup each
      return pack_codes_1010102(oetf_code<OUT>(clamp01(h.r), p.oetf_thr), oetf_code<OUT>(clamp01(h.g), p.oetf_thr),
                                oetf_code<OUT>(clamp01(h.b), p.oetf_thr));
    } else {  // PQ has no per-pixel transcendental: gather the 10-bit code of the reference's own 65536-node table
      const uint16_t* lut = (const uint16_t*)p.oetf_thr;  // pqOetfLUT + colorToRgba1010102 per node (host_tables.cpp)
      return pack_codes_1010102(lut[lut_index_f32<kOetfN>(clamp01(h.r))], lut[lut_index_f32<kOetfN>(clamp01(h.g))],
                                lut[lut_index_f32<kOetfN>(clamp01(h.b))]);
    }
  }
}

// gain value (0..1) -> factor through the GainLUT (gainmapmath.h:483-489)
__device__ __forceinline__ float gain_factor(float gain, const float* gain_tab, int ch,
                                             const ApplyParams& p) {
  if (!p.gamma_is_one[ch]) gain = (float)pow((double)gain, (double)p.gamma_inv[ch]);
  return gain_tab[ch * kGainN + lut_index_f32<kGainN>(gain)];
}

__device__ __forceinline__ uint32_t div_scale(uint32_t x, const ApplyParams& p) {
  return p.scale == 1 ? x : __umulhi(x, p.scale_magic);
}

// ---------------------------------------------------------------------------------------------
// integer-scale sampler (sampleMap / sampleMap3Channel with ShepardsIDW tables,
// gainmapmath.cpp:920-956, 1026-1080).  u8f = byte/255.0f table, idw = 4 weight tables.
// ---------------------------------------------------------------------------------------------
template <int NCH>
__device__ __forceinline__ void sample_map_table(const ApplyParams& p, const float* u8f,
                                                 const float* idw, uint32_t x, uint32_t yg,
                                                 float out[3]) {
  const uint32_t s = p.scale;
  uint32_t xl = div_scale(x, p), yl = div_scale(yg, p);
  const uint32_t ox = x - xl * s, oy = yg - yl * s;
  uint32_t xu = xl + 1, yu = yl + 1;
  xl = min(xl, p.gm.w - 1);
  xu = min(xu, p.gm.w - 1);
  yl = min(yl, p.gm.h - 1);
  yu = min(yu, p.gm.h - 1);
  int tbl = 0;
  if (xl == xu && yl == yu) tbl = 3;
  else if (xl == xu) tbl = 1;
  else if (yl == yu) tbl = 2;
  const float* w = idw + (size_t)tbl * s * s * 4 + (oy * s + ox) * 4;
  const float w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
  const uint8_t* d = (const uint8_t*)p.gm.p[0];
  const size_t st = p.gm.stride[0];
  const int bpp = p.map_bpp;
  const uint8_t* a1 = d + (xl + yl * st) * bpp;
  const uint8_t* a2 = d + (xl + yu * st) * bpp;
  const uint8_t* a3 = d + (xu + yl * st) * bpp;
  const uint8_t* a4 = d + (xu + yu * st) * bpp;
#pragma unroll
  for (int c = 0; c < NCH; c++)
    out[c] = u8f[a1[c]] * w0 + u8f[a2[c]] * w1 + u8f[a3[c]] * w2 + u8f[a4[c]] * w3;
}

// non-integer scale sampler (gainmapmath.cpp:871-918, 958-1024): per-pixel distances; the
// reference's pow(d,2)/sqrt are the double libm ones.
__device__ __forceinline__ float pyth(float dx, float dy) {
  return (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}
template <int NCH>
__device__ void sample_map_float(const ApplyParams& p, const float* u8f, uint32_t x, uint32_t yg,
                                 float out[3]) {
  const float xm = (float)x / p.scale_f, ym = (float)yg / p.scale_f;
  uint32_t xl = (uint32_t)floorf(xm), yl = (uint32_t)floorf(ym);
  uint32_t xu = xl + 1, yu = yl + 1;
  xl = min(xl, p.gm.w - 1);
  xu = min(xu, p.gm.w - 1);
  yl = min(yl, p.gm.h - 1);
  yu = min(yu, p.gm.h - 1);
  const uint8_t* d = (const uint8_t*)p.gm.p[0];
  const size_t st = p.gm.stride[0];
  const int bpp = p.map_bpp;
  float e1[3], e2[3], e3[3], e4[3];
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    e1[c] = u8f[d[(xl + yl * st) * bpp + c]];
    e2[c] = u8f[d[(xl + yu * st) * bpp + c]];
    e3[c] = u8f[d[(xu + yl * st) * bpp + c]];
    e4[c] = u8f[d[(xu + yu * st) * bpp + c]];
  }
  const float d1 = pyth(xm - (float)xl, ym - (float)yl);
  const float d2 = pyth(xm - (float)xl, ym - (float)yu);
  const float d3 = pyth(xm - (float)xu, ym - (float)yl);
  const float d4 = pyth(xm - (float)xu, ym - (float)yu);
  const float w1 = 1.0f / d1, w2 = 1.0f / d2, w3 = 1.0f / d3, w4 = 1.0f / d4;
  const float tot = w1 + w2 + w3 + w4;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    float v = e1[c] * (w1 / tot) + e2[c] * (w2 / tot) + e3[c] * (w3 / tot) + e4[c] * (w4 / tot);
    // early-outs in source order; the 1-channel sampler returns e2 (not e4) when the 4th
    // distance is zero (gainmapmath.cpp:908)
    if (d4 == 0.0f) v = (NCH == 1) ? e2[c] : e4[c];
    if (d3 == 0.0f) v = e3[c];
    if (d2 == 0.0f) v = e2[c];
    if (d1 == 0.0f) v = e1[c];
    out[c] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// generic kernel: one thread per pixel, every base format / scale; the sRGB, gain and byte -> float tables
// live in LDS (the IDW weights, whose size depends on the scale, stay in global memory), a workgroup
// walks tiles of 256 consecutive pixels of one row
// ---------------------------------------------------------------------------------------------
template <int OUT>
__global__ __launch_bounds__(kBlock) void apply_generic_kernel(const ApplyParams p) {
  __shared__ float s_srgb[kSrgbN];
  __shared__ float s_gain[3 * kGainN];
  __shared__ float s_u8f[256];
  copy_to_lds(s_srgb, p.tables + ApplyTables::kSrgbOff, kSrgbN, threadIdx.x, kBlock);
  copy_to_lds(s_gain, p.tables + ApplyTables::kGainOff, 3 * kGainN, threadIdx.x, kBlock);
  copy_to_lds(s_u8f, p.tables + ApplyTables::kU8fOff, 256u, threadIdx.x, kBlock);
  __syncthreads();
  const float* srgb = s_srgb;
  const float* gain_tab = s_gain;
  const float* u8f = s_u8f;  // b / 255.0f for every byte b: also the reference's RGBA8888 / RGB888 sample normalisation
  const float* idw = p.tables + ApplyTables::kIdwOff;
  const uint32_t w = p.sdr.w, h = p.sdr.h;
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    // get_pixel_fn (gainmapmath.cpp:354-470)
    Color3 g;
    const int fmt = p.sdr.fmt;
    if (fmt == UHDR_IMG_FMT_32bppRGBA8888) {
      uint32_t v = ((const uint32_t*)p.sdr.p[0])[x + (size_t)y * p.sdr.stride[0]];
      g.r = u8f[v & 0xff];
      g.g = u8f[(v >> 8) & 0xff];
      g.b = u8f[(v >> 16) & 0xff];
    } else if (fmt == UHDR_IMG_FMT_24bppRGB888) {
      const uint8_t* q = (const uint8_t*)p.sdr.p[0] + (size_t)x * 3 + (size_t)y * p.sdr.stride[0] * 3;
      g.r = u8f[q[0]];
      g.g = u8f[q[1]];
      g.b = u8f[q[2]];
    } else {
      const uint32_t hf = fmt == UHDR_IMG_FMT_24bppYCbCr444 ? 1 : 2;
      const uint32_t vf = fmt == UHDR_IMG_FMT_12bppYCbCr420 ? 2 : 1;
      uint8_t yy = ((const uint8_t*)p.sdr.p[0])[x + (size_t)y * p.sdr.stride[0]];
      uint8_t uu = ((const uint8_t*)p.sdr.p[1])[x / hf + (size_t)(y / vf) * p.sdr.stride[1]];
      uint8_t vv = ((const uint8_t*)p.sdr.p[2])[x / hf + (size_t)(y / vf) * p.sdr.stride[2]];
      g.r = (float)yy * (1 / 255.0f);
      g.g = (float)((int)uu - 128) * (1 / 255.0f);
      g.b = (float)((int)vv - 128) * (1 / 255.0f);
    }
    if (!p.sdr_is_rgb) g = yuv_to_rgb(g.r, g.g, g.b, p.yuv);  // p3YuvToRgb, always (jpegr.cpp:1723)
    Color3 lin = {srgb[lut_index_f32<kSrgbN>(g.r)], srgb[lut_index_f32<kSrgbN>(g.g)],
                  srgb[lut_index_f32<kSrgbN>(g.b)]};
    if (p.sdr_gamut_on) lin = mat3_apply(lin, p.gamut);
    float gn[3];
    const uint32_t yg = y + p.y0;
    if (p.map_ch == 1) {
      if (p.scale) sample_map_table<1>(p, u8f, idw, x, yg, gn);
      else sample_map_float<1>(p, u8f, x, yg, gn);
      gn[1] = gn[2] = gn[0];
    } else {
      if (p.scale) sample_map_table<3>(p, u8f, idw, x, yg, gn);
      else sample_map_float<3>(p, u8f, x, yg, gn);
    }
    float f0 = gain_factor(gn[0], gain_tab, 0, p), f1 = f0, f2 = f0;
    if (p.map_ch != 1) {
      f1 = gain_factor(gn[1], gain_tab, 1, p);
      f2 = gain_factor(gn[2], gain_tab, 2, p);
    }
    auto px = finish_pixel<OUT>(lin, f0, f1, f2, p, p.map_ch);
    using T = typename OutPix<OUT>::type;
    ((T*)p.dst.p[0])[x + (size_t)y * p.dst.stride[0]] = px;
  }
}

// ---------------------------------------------------------------------------------------------
// quad kernel (4:2:0 / 4:2:2 / 4:4:4 / RGBA8888 base, even geometry, gamma == 1 or scale == 1).
//   BASE  : 0 YCbCr 4:2:0 (the quad shares one chroma sample), 1 YCbCr 4:4:4, 2 packed RGBA8888.
//   MAPFMT: 0 Y400, 1 RGB888, 2 RGBA8888.
//   SMODE : 0 scale == 1 (byte -> factor table, no interpolation),
//           1 even integer scale (the four taps are shared by the whole 2x2 quad; only the
//             weights differ per pixel).
// One lane = one chroma sample = a 2x2 luma quad.  The two pixels of a row are carried as a
// 2-vector so the float pipeline maps onto the packed v_pk_{mul,add}_f32 instructions (each packed
// op is still one IEEE mul / add per element: bit-identical to scalar code).  All addressing is
// 32-bit offsets from uniform (SGPR) base pointers.
// ---------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kQuadsPerLane = 2;  // 2x2 quads a lane owns per quad row (128 pixels apart)
constexpr int kOversub = 1;       // launch_quad: workgroups launched per resident workgroup (UHDR_HIP_OVERSUB overrides)

__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }
__device__ __forceinline__ f2 clamp01_2(f2 v) { return (f2){clamp01(v.x), clamp01(v.y)}; }
// LUT index for x in [0,1] (no clip needed: clamp01 guarantees the range), as a BYTE offset
// int(fma(x, 1023, 0.5)) == int(double(float(x * 1023)) + 0.5) for every float x in [0, 1]
// (exhaustive check: tests/test_host_logic.py::test_lut_index_float_equivalence), so one packed
// FMA replaces the multiply/add pair.
__device__ __forceinline__ uint2 lut_off_1024(f2 x) {
  const f2 t = __builtin_elementwise_fma(x, splat(1023.0f), splat(0.5f));
  return (uint2){(uint32_t)(int)t.x << 2, (uint32_t)(int)t.y << 2};
}
// Same index for an UNCLAMPED argument a (the reference clamps to [0,1] first):
//   a < 0  -> the reference's index is 0; v_cvt_u32_f32 saturates negatives (and NaN) to 0
//   a > 1  -> the reference's index is 1023; here it is floor(a*1023 + 0.5) >= 1023 and the LDS
//             copy of the table is padded with entry 1023 up to kSrgbPad entries
// so clamp01 + index costs add/fma/cvt instead of add/clamp/fma/cvt.  |a| < 2 for every
// Y'CbCr input (y <= 1, |c*chroma| <= 1.772 * 0.5), i.e. index < 2048.
constexpr int kSrgbPad = 2048;
__device__ __forceinline__ uint32_t cvt_u32_sat(float v) {
  uint32_t r;
  asm("v_cvt_u32_f32 %0, %1" : "=v"(r) : "v"(v));
  return r;
}
__device__ __forceinline__ uint2 lut_off_unclamped(f2 a) {
  const f2 t = __builtin_elementwise_fma(a, splat(1023.0f), splat(0.5f));
  return (uint2){cvt_u32_sat(t.x) << 2, cvt_u32_sat(t.y) << 2};
}
// v_cvt_pkrtz_f16_f32 on raw bit patterns: two floats -> two halves, round toward zero
__device__ __forceinline__ uint32_t pkrtz_bits(uint32_t a, uint32_t b) {
  typedef __fp16 h2 __attribute__((ext_vector_type(2)));
  const h2 v = __builtin_amdgcn_cvt_pkrtz(__uint_as_float(a), __uint_as_float(b));
  return __builtin_bit_cast(uint32_t, v);
}
// floatToHalf (gainmapmath.h:160-173) for 0 <= v <= kMaxPixelFloatHdrLinear, including results in the
// sub-normal half range.  With b = bits + 0x1000 (the routine's round-half-up step): when the exponent
// of b is in the normal-half range the result is the float of bits b truncated (v_cvt_pkrtz); below it
// the routine computes (((0x7FF000 + mant(b)) >> (125 - exp(b))) + 1) >> 1, i.e. it takes the +0x1000 back
// out and rounds the ORIGINAL value: (floor(v * 2^25) + 1) >> 1, which is also 0 exactly where the routine
// flushes to zero.  Checked against the bit routine for every float of the sub-normal range and across
// both boundaries: tests/test_host_logic.py::test_half_subnormal_formula.
__device__ __forceinline__ uint32_t half_small_pair(uint32_t bits0, uint32_t bits1) {
  const uint32_t b0 = bits0 + 0x1000u, b1 = bits1 + 0x1000u;
  const uint32_t nrm = pkrtz_bits(b0, b1);
  const uint32_t s0 = (cvt_u32_sat(__uint_as_float(bits0) * 33554432.0f) + 1u) >> 1;
  const uint32_t s1 = (cvt_u32_sat(__uint_as_float(bits1) * 33554432.0f) + 1u) >> 1;
  const uint32_t lo = b0 < (113u << 23) ? s0 : (nrm & 0xffffu);
  const uint32_t hi = b1 < (113u << 23) ? s1 : (nrm >> 16);
  return lo | (hi << 16);
}
__device__ __forceinline__ f2 lds_gather(const float* base, uint2 byte_off) {
  const char* b = (const char*)base;
  return (f2){*(const float*)(b + byte_off.x), *(const float*)(b + byte_off.y)};
}
// the smallest float whose floatToHalf lands in the normal-half range: bits + 0x1000 >= 113 << 23
#define UHDR_HALF_FAST_MIN_BITS ((113u << 23) - 0x1000u)

// Streaming stores: every output byte is written once and never read back, so the quad kernels
// store with the nontemporal hint (global_store ... nt).  Measured on MI355X (tools/kbench, 8K):
// map A 80 -> 61 us, map B 87 -> 82 us, map C 92 -> 85 us.  The same hint on the LOADS is a loss
// (the 2-byte luma / 1-byte chroma loads of neighbouring instructions share cache lines), so the
// loads stay ordinary cached loads.  Ceiling for this 5.5 : 8 read : write mix with 16-byte accesses
// and no arithmetic (tools/ubench3): 80-82 us.
// Plane access through buffer resources: base (SGPR x 4, loop invariant) + a wave-uniform 32-bit row offset in an SGPR
// (`soffset`, computed on the scalar unit) + the lane's loop-invariant 32-bit column offset (`voffset`).  The address
// arithmetic of a load or store is then done by the memory pipeline: no per-row vector instruction at all (the flat
// `global_load v, v_off, s[base]` form needs base + row as a 64-bit scalar and LLVM reassociates that into vector adds).
// Raw buffers, stride 0, num_records 0xffffffff: offsets are plain byte offsets below 4 GiB (apply_quad_mode checks).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t plane_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)0xffffffffu, 0x00020000);
}
__device__ __forceinline__ uint32_t ld_u8(rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b8(r, voff, soff, 0); }
__device__ __forceinline__ uint32_t ld_u16(rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0); }
__device__ __forceinline__ uint2 ld_u64(rsrc_t r, uint32_t voff, uint32_t soff) {
  typedef uint32_t v2 __attribute__((ext_vector_type(2)));
  const v2 a = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return make_uint2(a.x, a.y);
}
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ void stream_store(void* a, T v) {
  __builtin_nontemporal_store(v, (T*)a);
}
// The OUTPUT stays on flat global stores (row offset added to the lane's column offset in the vector unit, one v_add per
// store).  A 16-byte buffer store with the row offset in an SGPR (`buffer_store_dwordx4 v[8:11], v33, s[40:43], s3 offen`)
// was measured to read its data registers for about sixteen cycles with NO interlock on gfx950: the compiler -- which inserts
// the documented wait state only when soffset is not a register (GCNHazardRecognizer::createsVALUHazard) -- scheduled the
// next pixel's `v_cvt_f32_ubyte` / `v_pk_mul_f32` into v[8:9] right behind the store, and lanes 12..63 stored the next
// pixel's luma instead of their blue / alpha halves (tools/dbg_apply.py, round 3).  Loads are unaffected: their VGPRs are
// written on return, under vmcnt.

// Raw bytes of one lane's quad, loaded one tile AHEAD of their use: on gfx9 stores and loads retire
// through the same in-order vmcnt counter, so a tile whose loads are issued after the previous
// tile's stores would wait for those stores to be acknowledged by memory.  Issuing the next
// tile's loads before the current tile's stores takes the store latency off the critical path.
template <int MAPFMT, int SMODE, int BASE>
struct QuadRaw {
  static constexpr int NCH = (MAPFMT == 0) ? 1 : 3;
  uint32_t y0, y1;  // BASE 0/1: two luma bytes of row 0 / row 1; BASE 2: unused
  uint32_t u, v;    // BASE 0: the quad's chroma bytes; BASE 1: row-0 chroma pairs (two bytes each); BASE 3: row-0 chroma bytes
  uint32_t c[(BASE == 0) ? 1 : 4];  // BASE 1 / 3: {.., .., u row 1, v row 1}; BASE 2: the four RGBA8888 pixels {r0p0, r0p1, r1p0, r1p1}
  uint32_t m[(SMODE == 0) ? 4 : 4 * NCH];  // SMODE 0: map bytes {row0 lo, row0 hi, row1 lo, row1 hi}; SMODE 1: tap bytes [tap][ch]
  uint32_t wrow;    // SMODE 1: row part of the weight-table index (wave-uniform)
  uint32_t y;       // first row of the quad (wave-uniform)
};

// SGPR budget: a wave may use at most 80 SGPRs if 8 workgroups of 256 threads are to be resident
// per CU (MI355X_MICROARCH.md, "Residency"); the launcher sizes the grid to exactly that residency
// so the whole image is processed in a single, balanced round.
//
// Work assignment: a wave owns one 128-pixel-wide column strip and walks down the image in steps
// of p.row_groups quad rows.  Everything that depends on the column (pixel offsets, gain-map tap
// columns, weight-table column, edge flags) is therefore loop invariant and lives in VGPRs;
// everything that depends on the row is wave-uniform and is computed on the scalar unit.
//
// SRC 1 (BASE 0 only): the base image arrives as JPEG coefficient blocks (p.coef_src, what jpeg_read_coefficients()
// yields) and never exists as planes in HBM.  A wave then owns 128 x 16 pixel tiles (one 4:2:0 MCU row of sixteen
// luma blocks): it dequantizes and inverse-transforms the tile's 32 + 8 + 8 blocks into its private LDS tile
// (idct_core.h, six eight-block passes) and feeds the same per-quad arithmetic from there; the gain map is still
// read from memory.  3 B/px of coefficients in instead of 3 in + 1.5 out + 1.5 in over four launches.
// Workgroup size: 256 for the linear F16 output (8 workgroups per CU share nothing but 14 KB of tables); 1024 for the
// HLG / PQ outputs, whose output-code bucket table is 41 / 17 KB -- sixteen waves share one copy and two such
// workgroups (32 waves) still fit a CU's 160 KB.
// F16 output: 256-thread workgroups.  HLG / PQ output: 768 threads share one copy of the large LDS tables (code buckets,
// 31 - 72 KB per workgroup); two workgroups per CU = 6 waves per SIMD, which leaves the wave 80 VGPRs and the full SGPR file
// (at 8 waves per SIMD the compiler may hand out 72 SGPRs -- 800 / 8 minus the trap handler's 16 -- and the HLG variants
// spilled their kernel-argument pointer and buffer resources into VGPR lanes inside the row loop)
template <int OUT> constexpr int quad_block() { return OUT == 0 ? kBlock : 768; }
template <int OUT, int SRC> constexpr int quad_waves_per_eu() { return (OUT != 0 && SRC == 0) ? 6 : 1; }
// SGPR budget of a variant: 80 keeps eight 256-thread workgroups resident per CU (MI355X_MICROARCH.md "Residency"); the
// interpolating variants (SMODE 1) and the coefficient input need more scalar state (four buffer resources, the row
// arithmetic of the taps) and take 96 = seven workgroups per CU rather than spilling SGPRs into VGPR lanes
#if defined(UHDR_EXP_SGPR96_ALL)  // tools/kbench experiment
template <int SMODE, int SRC> constexpr int quad_sgprs() { return 96; }
#else
template <int SMODE, int SRC> constexpr int quad_sgprs() { return (SMODE == 0 && SRC == 0) ? 80 : 96; }
#endif
template <int OUT, int MAPFMT, int SMODE, int BASE, int SRC = 0>
__device__ __forceinline__ void apply_quad_body(const ApplyParams& p) {
  static_assert(SRC == 0 || BASE == 0, "coefficient input is a 4:2:0 base image");
  constexpr int BLK = quad_block<OUT>();
  constexpr int NCH = (MAPFMT == 0) ? 1 : 3;
  constexpr int BPP = (MAPFMT == 0) ? 1 : (MAPFMT == 1 ? 3 : 4);
  using Raw = QuadRaw<MAPFMT, SMODE, BASE>;
  __shared__ __attribute__((aligned(16))) float s_srgb[kSrgbPad];
  __shared__ __attribute__((aligned(16))) float s_gain[(SMODE == 0) ? 4 : NCH * kGainN];
  __shared__ __attribute__((aligned(16))) float s_u8f[(BASE == 2) ? 256 : 4];  // byte / 255.0f: RGBA8888 base samples (BASE 2)
  __shared__ __attribute__((aligned(16))) float s_tap[(SMODE == 1) ? 512 : 4];  // map taps (SMODE 1): byte / 255.0f, pre-splatted {x, x}
  __shared__ __attribute__((aligned(16))) float s_fac[(SMODE == 0) ? NCH * 256 : 4];
  // chroma byte -> p3YuvToRgb's products {cr * vf, gcr * vf} / {gcb * uf, cb * uf} (host_tables.cpp): two 8-byte LDS reads per
  // chroma sample replace two integer subtractions, two conversions and six multiplications.  (Entries pre-splatted to
  // {a, a, b, b} save four register moves per quad but cost 4 KB more of staging per workgroup: measured slower, 8K map C
  // 80.8 vs 77.9 us.)
  __shared__ __attribute__((aligned(16))) float s_cv[(BASE == 0 || BASE == 3) ? 512 : 4];
  __shared__ __attribute__((aligned(16))) float s_cu[(BASE == 0 || BASE == 3) ? 512 : 4];
  auto chroma = [&](uint32_t ub, uint32_t vb, f2& crv, f2& gcbu, f2& gcrv, f2& cbu) {
    const f2 tv = *(const f2*)(s_cv + 2 * vb), tu = *(const f2*)(s_cu + 2 * ub);
    crv = splat(tv.x); gcrv = splat(tv.y); gcbu = splat(tu.x); cbu = splat(tu.y);
  };
  __shared__ __attribute__((aligned(8))) uint2 s_code[(OUT == 1) ? kOetfBucketsHlg : (OUT == 2 ? kOetfBucketsPq : 1)];  // HLG / PQ output-code buckets
  // IDW weights re-laid out for pixel PAIRS: entry (table, oy, ox/2) holds
  // {w0(ox), w0(ox+1), w1(ox), w1(ox+1), w2(ox), w2(ox+1), w3(ox), w3(ox+1)}
  __shared__ __attribute__((aligned(16))) float s_idw[(SMODE == 0) ? 4 : 4 * kMaxIdwScaleLds * kMaxIdwScaleLds * 4];

  const uint32_t tid = threadIdx.x;
  // Per-workgroup tables -> LDS.  Called AFTER the wave has issued the global loads of its first work item, so that the
  // staging (a round trip to L2 per table) overlaps the first HBM accesses instead of preceding them.
  // 16-byte copies: every region of the table block the host laid out as an LDS image (uhdr_types.h: ApplyTables) is a
  // multiple of four floats and starts on a 16-byte boundary
  auto copy16 = [&](float* dst, const float* src, uint32_t nfloats) {
    copy_words_to_lds<4>((uint32_t*)dst, (const uint32_t*)src, nfloats, tid, BLK);  // (lds_copy.h: four 16-byte loads in flight per thread)
  };
  auto stage_tables = [&]() {
    copy16(s_srgb, p.tables + ApplyTables::kSrgbPadOff, kSrgbPad);
    if constexpr (OUT != 0) {
      copy_to_lds(s_code, p.oetf_buckets, p.oetf_n * (uint32_t)(sizeof(s_code[0]) / 4), tid, BLK);
    }
    if constexpr (BASE == 0 || BASE == 3) {
      copy16(s_cv, p.tables + ApplyTables::kChromaVOff, 512);
      copy16(s_cu, p.tables + ApplyTables::kChromaUOff, 512);
    }
    if constexpr (BASE == 2) copy16(s_u8f, p.tables + ApplyTables::kU8fOff, 256);
    if constexpr (SMODE == 0) {
      copy16(s_fac, p.tables + ApplyTables::kFacOff, NCH * 256);
    } else {
      copy16(s_gain, p.tables + ApplyTables::kGainOff, NCH * kGainN);
      copy16(s_tap, p.tables + ApplyTables::kTapOff, 512);
      copy16(s_idw, p.tables + ApplyTables::idw_pair_off((int)p.scale), 4 * p.scale * p.scale * 4);
    }
    __syncthreads();
  };

  const uint32_t qw = p.sdr.w >> 1, qh = p.sdr.h >> 1;
  const uint32_t strips_x = (qw + 64 * kQuadsPerLane - 1) / (64 * kQuadsPerLane);  // a wave owns 128 * kQuadsPerLane pixel columns
  const uint32_t lane = tid & 63;
  // ---- prefetcher workgroups (round 4) ------------------------------------------------------------------------------------
  // A write-dominated launch (8K, Y400 map at scale 4: 51 MB in, 265 MB out) whose inputs are cold pays 14 us for them: its
  // loads are latency critical -- every one feeds arithmetic that feeds a store -- and they queue behind a saturated write
  // stream.  A read sweep ahead of the consumers fixes that (the access pattern alone: 64 -> 55 us, tools/ubench8), but as a
  // prologue of every wave (touch_ahead, round 3) it holds all stores back for as long as it lasts.  Here the first
  // p.prefetch_wgs workgroups do nothing but sweep: cooperatively, front to back in the order the row bands will be
  // consumed (luma : Cb : Cr = 4 : 1 : 1 per step, the small map first), twelve line-touching loads (8 KiB each) in flight per wave and no
  // dependent work, so their latency costs nothing; everybody else finds the lines in L2 / the infinity cache.  They finish
  // in the first third of the launch and leave.
  const uint32_t pf_wgs = SRC == 0 ? p.prefetch_wgs : 0u;
  if (SRC == 0 && blockIdx.x < pf_wgs) {
    const uint32_t lane_ = tid & 63, nw = pf_wgs * (BLK / 64), pw = blockIdx.x * (BLK / 64) + (tid >> 6);
    const uint8_t* py = (const uint8_t*)p.sdr.p[0];
    const uint8_t* pu = (const uint8_t*)p.sdr.p[1];
    const uint8_t* pv = (const uint8_t*)p.sdr.p[2];
    const uint8_t* pm = (const uint8_t*)p.gm.p[0];
    const uint32_t qh_ = p.sdr.h >> 1;
    const uint32_t by = (qh_ * 2 - 1) * p.sdr.stride[0] + p.sdr.w, bc = (qh_ - 1) * p.sdr.stride[1] + p.sdr.w / 2;
    const uint32_t bm = ((p.gm.h - 1) * p.gm.stride[0] + p.gm.w) * BPP;
    uint32_t acc = 0;
    // one TOUCH per 128-byte line: lane l of a load reads the first word of line `chunk * 64 + l`, so a single wave instruction
    // brings 8 KiB of the plane into L2 / the infinity cache (and keeps 64 line fetches in flight instead of the 8 of a
    // contiguous 16-byte-per-lane read); the word itself is irrelevant
    auto touch = [&](const uint8_t* base, uint32_t bytes, uint32_t chunk) -> uint32_t {
      const uint32_t mis = (4u - (uint32_t)((uintptr_t)base & 3u)) & 3u;
      const uint32_t last = bytes > mis + 4u ? (bytes - mis - 4u) & ~3u : 0u;  // byte offset of the last whole word
      const uint32_t off = min((chunk * 64u + lane_) * 128u, last);
      return *(const uint32_t*)(base + mis + off);
    };
    for (uint32_t c = pw; c * 8192u < bm; c += nw) acc ^= touch(pm, bm, c);  // the map: small at scale > 1, needed from the first row on
    const uint32_t steps = (bc / 8192u + nw) / nw;  // every wave runs the same number of steps (its last chunks clamp)
    for (uint32_t it = 0; it < steps; it += 2) {
      uint32_t x[12];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint32_t c = (it + h) * nw + pw;
#pragma unroll
        for (int k = 0; k < 4; k++) x[6 * h + k] = touch(py, by, 4 * c + k);
        x[6 * h + 4] = touch(pu, bc, c);
        x[6 * h + 5] = touch(pv, bc, c);
      }
#pragma unroll
      for (int k = 0; k < 12; k++) acc ^= x[k];
    }
    if (acc == 0x9e3779b9u && p.n_frames == 0xffffffffu) ((uint8_t*)p.dst.p[0])[0] = 0;  // never true: keeps the loads alive
    return;
  }
  const uint32_t wave = (blockIdx.x - pf_wgs) * (BLK / 64) + __builtin_amdgcn_readfirstlane(tid >> 6);  // SGPR
  const uint32_t groups = p.row_groups;
  const uint32_t per_frame = groups * strips_x;
  // a few surplus waves of the last workgroup have no work: they help staging the tables and leave
  const bool live = SRC != 0 || wave < per_frame * p.n_frames;
  // batch: a wave stays inside ONE frame, so its plane pointers are loop invariant (five scalar
  // loads from the frame table, before the loop)
  const uint32_t frame = live ? wave / per_frame : 0u, wf = live ? wave - frame * per_frame : 0u;
  const uint32_t qy0 = wf / strips_x, sx = wf - qy0 * strips_x;
  const uint8_t *yp, *up, *vp, *mp;
  uint8_t* dp;
  if (p.n_frames > 1) {  // the frame's five plane pointers: scalar loads from the kernel-argument segment
    const FramePtrs& fp = p.frame_tab[frame];
    yp = fp.y; up = fp.u; vp = fp.v; mp = fp.map; dp = fp.dst;
  } else {
    yp = (const uint8_t*)p.sdr.p[0]; up = (const uint8_t*)p.sdr.p[1]; vp = (const uint8_t*)p.sdr.p[2];
    mp = (const uint8_t*)p.gm.p[0]; dp = (uint8_t*)p.dst.p[0];
  }

  const rsrc_t ry = plane_rsrc(yp), ru = plane_rsrc(up), rv = plane_rsrc(vp), rm = plane_rsrc(mp);
  (void)ru; (void)rv;
  const uint32_t sy = p.sdr.stride[0], su = p.sdr.stride[1], sv = p.sdr.stride[2];
  const uint32_t sm = p.gm.stride[0];
  const bool map_even = (sm & 1u) == 0;  // RGB888 map: every row starts at an even byte offset
  (void)map_even;
  constexpr uint32_t OPX = (OUT == 0) ? 8 : 4;  // output bytes per pixel
  const uint32_t sd = p.dst.stride[0] * OPX;    // destination row pitch in bytes
  const float k255 = 1 / 255.0f;
  const Yuv2Rgb yk = p.yuv;
  const f2 off_s0 = splat(p.offset_sdr[0]), off_h0 = splat(p.offset_hdr[0]);
  const f2 off_s1 = splat(p.offset_sdr[NCH == 1 ? 0 : 1]), off_h1 = splat(p.offset_hdr[NCH == 1 ? 0 : 1]);
  const f2 off_s2 = splat(p.offset_sdr[NCH == 1 ? 0 : 2]), off_h2 = splat(p.offset_hdr[NCH == 1 ? 0 : 2]);
  const uint32_t scale = p.scale, half_scale = p.scale >> 1, magic = p.scale_magic;
  const uint32_t gmh1 = p.gm.h - 1, y0g = p.y0;
  // LDS byte address of the code table minus its first bucket's offset (wraps modulo 2^32, as does the add that uses it)
  const uint32_t code_rel = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)s_code - p.oetf_base8;
  const uint32_t code_lo = p.oetf_lo_bits, code_hi = p.oetf_hi_bits;
  (void)code_rel; (void)code_lo; (void)code_hi;

  // ---- loop-invariant, per-lane column state --------------------------------------------------
  // A lane owns kQuadsPerLane quads of every quad row, 128 pixels apart: each load / store
  // instruction of the wave still covers one contiguous run (128 B of luma, 1 KiB of F16 output), and
  // the runs of one row follow each other, so a wave writes 2 KiB contiguous per row.  Measured on the
  // bare access pattern (tools/ubench4): 79 us -> 75 us for the 8K map-C frame.
  // A ragged last strip is shifted left so that it ends at the image edge (it overlaps its
  // neighbour and rewrites identical pixels): every lane is always live.
  struct Col {
    uint32_t xc;                   // pixel column of the quad
    uint32_t col_l, col_u, wcol;   // SMODE 1: tap column byte offsets, column part of the weight index
  };
  Col col[kQuadsPerLane];
  auto make_col = [&](uint32_t xc) -> Col {
    Col c;
    c.xc = xc;
    c.col_l = c.col_u = c.wcol = 0;
    if constexpr (SMODE == 1) {
      const uint32_t gmw1 = p.gm.w - 1;
      uint32_t xl = __umulhi(xc, magic);
      const uint32_t ox = xc - xl * scale;
      const uint32_t xu = min(xl + 1, gmw1);
      xl = min(xl, gmw1);
      c.col_l = xl * BPP;
      c.col_u = xu * BPP;
      // table select: 0 default, 1 no-right, 2 no-bottom, 3 corner (gainmapmath.cpp:946-953)
      c.wcol = ((xl == xu ? 1u : 0u) * scale * half_scale + (ox >> 1)) * 8;
    }
    return c;
  };
#pragma unroll
  for (int hq = 0; hq < kQuadsPerLane; hq++)
    col[hq] = make_col(SRC == 0 ? (min((sx * kQuadsPerLane + hq) * 64, qw - 64) + lane) * 2 : lane * 2);  // SRC 1: set per tile
  bool store_ok = true;  // SRC 1: lanes / rows of a tile that lie outside the image compute but do not store
  (void)store_ok;

  // ---- issue the loads of quad row qy_ (wave-uniform) -------------------------------------------
  auto fetch = [&](uint32_t qy_, auto HQ) -> Raw {
    constexpr int hq = decltype(HQ)::value;
    const uint32_t xc = col[hq].xc, xq = xc >> 1, xmap = xc * BPP, col_l = col[hq].col_l, col_u = col[hq].col_u;
    (void)xq; (void)xmap; (void)col_l; (void)col_u;
    Raw r;
    qy_ = min(qy_, qh - 1);  // past the end: recompute the last row (identical bytes)
    const uint32_t y = qy_ * 2;
    r.y = y;
    // scalar row offset (SGPR) + per-lane column offset (VGPR, loop invariant), added by the memory pipeline
    if constexpr (SRC == 1) {  // luma / chroma come from the wave's LDS tile (filled in by the tile loop)
      r.y0 = r.y1 = r.u = r.v = 0;
    } else if constexpr (BASE == 2) {  // packed RGBA8888: two pixels (8 bytes) per row
      const uint32_t prow = y * sy * 4;
      const uint2 a = ld_u64(ry, xc * 4, prow), b = ld_u64(ry, xc * 4, prow + sy * 4);
      r.c[0] = a.x; r.c[1] = a.y; r.c[2] = b.x; r.c[3] = b.y;
      r.y0 = r.y1 = r.u = r.v = 0;
    } else {
      const uint32_t yrow = y * sy;
      r.y0 = ld_u16(ry, xc, yrow);
      r.y1 = ld_u16(ry, xc, yrow + sy);
      if constexpr (BASE == 0) {
        r.u = ld_u8(ru, xq, qy_ * su);
        r.v = ld_u8(rv, xq, qy_ * sv);
      } else if constexpr (BASE == 3) {  // 4:2:2: one chroma sample per row of the quad
        r.u = ld_u8(ru, xq, y * su);
        r.v = ld_u8(rv, xq, y * sv);
        r.c[2] = ld_u8(ru, xq, y * su + su);
        r.c[3] = ld_u8(rv, xq, y * sv + sv);
        r.c[0] = r.c[1] = 0;
      } else {  // 4:4:4: a chroma pair per row
        r.u = ld_u16(ru, xc, y * su);
        r.v = ld_u16(rv, xc, y * sv);
        r.c[2] = ld_u16(ru, xc, y * su + su);
        r.c[3] = ld_u16(rv, xc, y * sv + sv);
        r.c[0] = r.c[1] = 0;
      }
    }
    const uint32_t yg = y + y0g;
    if constexpr (SMODE == 0) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const uint32_t mrow = (yg + k) * sm * BPP;
        if constexpr (MAPFMT == 0) {
          r.m[2 * k] = ld_u16(rm, xmap, mrow);
          r.m[2 * k + 1] = 0;
        } else if constexpr (MAPFMT == 1) {
          // 6 bytes at a 2-byte aligned offset: ONE 8-byte load (the part runs with unaligned access enabled; the two bytes
          // read too many belong to the next lane's pixels or the row's padding and are never looked at) -- three 16-bit
          // loads per row made this layout issue ten load instructions per quad where the RGBA8888 map needs six, and it ran no
          // faster for 7 % fewer bytes (round 3: 0.646 against 0.719).  The LAST map row keeps the exact loads: two bytes
          // beyond it may be beyond the allocation (a 3840 x 2160 x 3 byte plane ends on a page boundary).
          if ((yg + k) < gmh1) {  // wave-uniform
            if (map_even) {
              // ... and that load naturally aligned (4 bytes) when the row pitch is even: the dword pair that holds the six bytes,
              // shifted down by 0 or 16 bits
              const uint32_t t = xmap + (mrow & 2u);
              const uint2 a = ld_u64(rm, t & ~3u, mrow & ~3u);
              const uint32_t sh = (t & 2u) * 8u;
              r.m[2 * k] = __builtin_amdgcn_alignbit(a.y, a.x, sh);
              r.m[2 * k + 1] = a.y >> sh;
            } else {
              const uint2 a = ld_u64(rm, xmap, mrow);
              r.m[2 * k] = a.x;
              r.m[2 * k + 1] = a.y;
            }
          } else {
            r.m[2 * k] = ld_u16(rm, xmap, mrow) | (ld_u16(rm, xmap + 2, mrow) << 16);
            r.m[2 * k + 1] = ld_u16(rm, xmap + 4, mrow);
          }
        } else {
          const uint2 a = ld_u64(rm, xmap, mrow);
          r.m[2 * k] = a.x;
          r.m[2 * k + 1] = a.y;
        }
      }
      r.wrow = 0;
    } else {
      uint32_t yl = __umulhi(yg, magic);
      const uint32_t oy = yg - yl * scale;
      const uint32_t yu = min(yl + 1, gmh1);
      yl = min(yl, gmh1);
      r.wrow = ((yl == yu ? 2u : 0u) * scale + oy) * half_scale * 8;
      const uint32_t rl = yl * sm * BPP, rup = yu * sm * BPP;
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        r.m[0 * NCH + c] = ld_u8(rm, col_l + c, rl);
        r.m[1 * NCH + c] = ld_u8(rm, col_l + c, rup);
        r.m[2 * NCH + c] = ld_u8(rm, col_u + c, rl);
        r.m[3 * NCH + c] = ld_u8(rm, col_u + c, rup);
      }
    }
    return r;
  };

  // ---- compute + store one quad row -----------------------------------------------------------
  // GM: the gamut conversion of this call -- 0 none, 1 on the SDR side (before the gain is applied), 2 on the HDR side (after).
  // It is wave-uniform and loop-invariant, so the pixel loop exists once per case (see the dispatch below) instead of
  // merging two register assignments behind a branch in every row.
  auto process = [&](const Raw& q, auto HQ, auto GM) {
    constexpr int hq = decltype(HQ)::value;
    constexpr int gmode = decltype(GM)::value;
    const uint32_t xdst = col[hq].xc * OPX, wcol = col[hq].wcol;
    (void)wcol;
    f2 tap[(SMODE == 0) ? 1 : 4][NCH];
    if constexpr (SMODE == 1) {
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int c = 0; c < NCH; c++) tap[k][c] = *(const f2*)(s_tap + 2 * q.m[k * NCH + c]);
    }
    // getYuv4abPixel chroma (gainmapmath.cpp:370-374) and the p3YuvToRgb chroma products shared
    // by the four pixels (gainmapmath.cpp:177-181)
    // BASE 0: one chroma sample for the quad; BASE 1: per pixel (computed per row below)
    f2 crv = splat(0.0f), gcbu = crv, gcrv = crv, cbu = crv;
    if constexpr (BASE == 0) {
      chroma(q.u, q.v, crv, gcbu, gcrv, cbu);
    }
    const uint32_t drow = q.y * sd;  // wave-uniform row offset
#pragma unroll
    for (int r = 0; r < 2; r++) {
      f2 lr, lg, lb;
      if constexpr (BASE == 2) {
        // getRgba8888Pixel: byte / 255.0f (the host-built table holds exactly those quotients), then
        // srgbInvOetfLUT; the samples are in [0, 1], no clamp involved
        const uint32_t a = q.c[2 * r], b = q.c[2 * r + 1];
        const f2 er = {s_u8f[a & 0xff], s_u8f[b & 0xff]};
        const f2 eg = {s_u8f[(a >> 8) & 0xff], s_u8f[(b >> 8) & 0xff]};
        const f2 eb = {s_u8f[(a >> 16) & 0xff], s_u8f[(b >> 16) & 0xff]};
        lr = lds_gather(s_srgb, lut_off_1024(er));
        lg = lds_gather(s_srgb, lut_off_1024(eg));
        lb = lds_gather(s_srgb, lut_off_1024(eb));
      } else {
        const uint32_t yb = r == 0 ? q.y0 : q.y1;
        const f2 yf = (f2){(float)(yb & 0xff), (float)(yb >> 8)} * k255;
        if constexpr (BASE == 3) {  // 4:2:2: this row's chroma sample, shared by its two pixels
          chroma(r == 0 ? q.u : q.c[2], r == 0 ? q.v : q.c[3], crv, gcbu, gcrv, cbu);
        }
        if constexpr (BASE == 1) {  // 4:4:4: this row's own chroma pair
          const uint32_t ub = r == 0 ? q.u : q.c[2], vb = r == 0 ? q.v : q.c[3];
          const f2 uf = (f2){(float)((int)(ub & 0xff) - 128), (float)((int)(ub >> 8) - 128)} * k255;
          const f2 vf = (f2){(float)((int)(vb & 0xff) - 128), (float)((int)(vb >> 8) - 128)} * k255;
          crv = yk.cr * vf; gcbu = yk.gcb * uf; gcrv = yk.gcr * vf; cbu = yk.cb * uf;
        }
        // p3YuvToRgb + clampPixelFloat + srgbInvOetfLUT; the clamp is absorbed by the index
        // conversion and the padded table (see lut_off_unclamped)
        lr = lds_gather(s_srgb, lut_off_unclamped(yf + crv));
        lg = lds_gather(s_srgb, lut_off_unclamped(yf - gcbu - gcrv));
        lb = lds_gather(s_srgb, lut_off_unclamped(yf + cbu));
      }
      if constexpr (gmode == 1) {
        const Mat3& m = p.gamut;
        const f2 nr = m.m[0] * lr + m.m[1] * lg + m.m[2] * lb;
        const f2 ng = m.m[3] * lr + m.m[4] * lg + m.m[5] * lb;
        const f2 nb = m.m[6] * lr + m.m[7] * lg + m.m[8] * lb;
        lr = nr; lg = ng; lb = nb;
      }
      f2 f0, f1, f2_;
      if constexpr (SMODE == 0) {
        const uint32_t a = q.m[2 * r], b = q.m[2 * r + 1];
        if constexpr (MAPFMT == 0) {
          f0 = (f2){s_fac[a & 0xff], s_fac[(a >> 8) & 0xff]};
          f1 = f0; f2_ = f0;
        } else if constexpr (MAPFMT == 1) {  // R0 G0 B0 R1 | G1 B1
          f0 = (f2){s_fac[a & 0xff], s_fac[a >> 24]};
          f1 = (f2){s_fac[256 + ((a >> 8) & 0xff)], s_fac[256 + (b & 0xff)]};
          f2_ = (f2){s_fac[512 + ((a >> 16) & 0xff)], s_fac[512 + ((b >> 8) & 0xff)]};
        } else {
          f0 = (f2){s_fac[a & 0xff], s_fac[b & 0xff]};
          f1 = (f2){s_fac[256 + ((a >> 8) & 0xff)], s_fac[256 + ((b >> 8) & 0xff)]};
          f2_ = (f2){s_fac[512 + ((a >> 16) & 0xff)], s_fac[512 + ((b >> 16) & 0xff)]};
        }
      } else {
        const float* wp = s_idw + (q.wrow + r * half_scale * 8) + wcol;  // scalar row part + lane column part
        const float4 wa = *(const float4*)wp;
        const float4 wb = *(const float4*)(wp + 4);
        const f2 w0 = {wa.x, wa.y}, w1 = {wa.z, wa.w}, w2 = {wb.x, wb.y}, w3 = {wb.z, wb.w};
        // sampleMap: e1*w0 + e2*w1 + e3*w2 + e4*w3, left to right (gainmapmath.cpp:955, 1079)
        const f2 g0 = tap[0][0] * w0 + tap[1][0] * w1 + tap[2][0] * w2 + tap[3][0] * w3;
        f0 = lds_gather(s_gain, lut_off_1024(g0));  // weights sum to 1 +- 1ulp: the index stays <= 1023
        if constexpr (NCH == 3) {
          const f2 g1 = tap[0][1] * w0 + tap[1][1] * w1 + tap[2][1] * w2 + tap[3][1] * w3;
          const f2 g2 = tap[0][2] * w0 + tap[1][2] * w1 + tap[2][2] * w2 + tap[3][2] * w3;
          f1 = lds_gather(s_gain + kGainN, lut_off_1024(g1));
          f2_ = lds_gather(s_gain + 2 * kGainN, lut_off_1024(g2));
        } else {
          f1 = f0; f2_ = f0;
        }
      }
      // applyGainLUT: ((e + offset_sdr) * factor) - offset_hdr  (gainmapmath.cpp:807-810, 848-855)
      f2 hr = ((lr + off_s0) * f0) - off_h0;
      f2 hg = ((lg + off_s1) * f1) - off_h1;
      f2 hb = ((lb + off_s2) * f2_) - off_h2;
      if constexpr (OUT == 0) {
        if constexpr (gmode == 2) {
          const Mat3& m = p.gamut;
          const f2 nr = m.m[0] * hr + m.m[1] * hg + m.m[2] * hb;
          const f2 ng = m.m[3] * hr + m.m[4] * hg + m.m[5] * hb;
          const f2 nb = m.m[6] * hr + m.m[7] * hg + m.m[8] * hb;
          hr = nr; hg = ng; hb = nb;
        }
        const float c0r = clamp_linear(hr.x), c0g = clamp_linear(hg.x), c0b = clamp_linear(hb.x);
        const float c1r = clamp_linear(hr.y), c1g = clamp_linear(hg.y), c1b = clamp_linear(hb.y);
        // all six values are >= 0, so "every one is in the normal-half range" is one min + compare
        const float mn = fminf(__builtin_fminf(__builtin_fminf(c0r, c0g), c0b), __builtin_fminf(__builtin_fminf(c1r, c1g), c1b));  // two v_min3 + one v_min
        uint4 o;
        // floatToHalf for normal halves = add 0x1000 to the bits, then truncate: v_cvt_pkrtz
        o.x = pkrtz_bits(__float_as_uint(c0r) + 0x1000u, __float_as_uint(c0g) + 0x1000u);
        o.y = pkrtz_bits(__float_as_uint(c0b) + 0x1000u, 0x3F800000u);
        o.z = pkrtz_bits(__float_as_uint(c1r) + 0x1000u, __float_as_uint(c1g) + 0x1000u);
        o.w = pkrtz_bits(__float_as_uint(c1b) + 0x1000u, 0x3F800000u);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(__float_as_uint(mn) < UHDR_HALF_FAST_MIN_BITS) != 0, 0)) {
          // some value of the wave lands in the sub-normal half range (black or near-black pixels)
          o.x = half_small_pair(__float_as_uint(c0r), __float_as_uint(c0g));
          o.y = half_small_pair(__float_as_uint(c0b), 0x3F800000u - 0x1000u);
          o.z = half_small_pair(__float_as_uint(c1r), __float_as_uint(c1g));
          o.w = half_small_pair(__float_as_uint(c1b), 0x3F800000u - 0x1000u);
        }
        if (SRC == 0 || store_ok) stream_store<u4v>(dp + (drow + r * sd + xdst), (u4v){o.x, o.y, o.z, o.w});
      } else {
        const float peak = (OUT == 1) ? 1000.0f : 10000.0f;  // kHlgMaxNits / kPqMaxNits
        const f2 p203 = splat(203.0f), pk = splat(peak), rpk = splat(1.0f / peak);
        // (x * 203) / peak, two roundings as written in the reference; the division is div_const's
        // exact three-instruction form (device_math.h), packed
        auto div_peak = [&](f2 x) {
          const f2 a = x * p203, q0 = a * rpk;
          const f2 r = __builtin_elementwise_fma(-pk, q0, a);
          return __builtin_elementwise_fma(r, rpk, q0);
        };
        if (!p.oetf_prescaled) {  // else the code table was built on the unscaled value (host_tables.cpp: make_bucket_table)
          hr = div_peak(hr);
          hg = div_peak(hg);
          hb = div_peak(hb);
        }
        if constexpr (gmode == 2) {
          const Mat3& m = p.gamut;
          const f2 nr = m.m[0] * hr + m.m[1] * hg + m.m[2] * hb;
          const f2 ng = m.m[3] * hr + m.m[4] * hg + m.m[5] * hb;
          const f2 nb = m.m[6] * hr + m.m[7] * hg + m.m[8] * hb;
          hr = nr; hg = ng; hb = nb;
        }
        uint2 o;
        o.x = pack_codes_1010102(oetf_code_bucket<OUT>(hr.x, code_rel, code_lo, code_hi), oetf_code_bucket<OUT>(hg.x, code_rel, code_lo, code_hi),
                                 oetf_code_bucket<OUT>(hb.x, code_rel, code_lo, code_hi));
        o.y = pack_codes_1010102(oetf_code_bucket<OUT>(hr.y, code_rel, code_lo, code_hi), oetf_code_bucket<OUT>(hg.y, code_rel, code_lo, code_hi),
                                 oetf_code_bucket<OUT>(hb.y, code_rel, code_lo, code_hi));
        if (SRC == 0 || store_ok) stream_store<u2v>(dp + (drow + r * sd + xdst), (u2v){o.x, o.y});
      }
    }
  };

  // software pipeline, ping-pong registers: the loads of the next work item (the lane's other quad of
  // this row, then the first quad of the next row) are in flight while the current one is computed
  // and stored.  Rows past the end are clamped (see fetch) and recompute the last row.
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, kQuadsPerLane - 1>;
  static_assert(kQuadsPerLane == 2, "the pipeline below alternates between the lane's two quads");
  using G0 = std::integral_constant<int, 0>;
  using G1 = std::integral_constant<int, 1>;
  using G2 = std::integral_constant<int, 2>;
  if constexpr (SRC == 0) {
    Raw a = fetch(qy0, H0{});  // in flight while the tables are staged
    // ---- touch-ahead (launcher: small inputs, cold in HBM) -----------------------------------------------------------------
    // Narrow reads that trickle into a saturated write stream cost DRAM far more than their bytes: at 8K with the Y400
    // map (51 MB in, 265 MB out) the access pattern alone takes 64-70 us with cold inputs against 47 us when they sit in
    // the infinity cache (tools/ubench8).  So the launch opens with ONE read burst: wave k reads the k-th slice of every
    // input plane with 16-byte loads (1 KiB contiguous per instruction) -- whichever wave needs the lines later finds them
    // in L2 / the 256 MiB infinity cache -- and the rest of the launch is a pure store stream to DRAM (pattern: 55 us).
    // The values are folded into a word that is only inspected after the row loop, so the loads just stay in flight.
    uint32_t touched = 0;
    if (p.touch_ahead && live) {
      const uint32_t nw = per_frame * p.n_frames;
      // one plane: this wave's slice, four 1 KiB read instructions (16 bytes per lane) in flight at a time; branch free --
      // a slot past the end of the slice re-reads its last unit
      auto plane = [&](const uint8_t* base, uint32_t bytes) {
        const uint32_t mis = (16u - (uint32_t)((uintptr_t)base & 15u)) & 15u;  // to the first 16-byte boundary
        if (bytes < mis + 16u) return;
        const uint32_t n16 = (bytes - mis) >> 4;                                // whole 16-byte units
        const uint32_t per = ((n16 + nw - 1) / nw + 63u) & ~63u;               // units per wave: whole instructions
        const uint32_t lo = wave * per, hi = min(lo + per, n16);
        if (lo >= hi) return;  // wave-uniform
        const u4v* src = (const u4v*)(base + mis);
        for (uint32_t i0 = lo; i0 < hi; i0 += 256) {
          u4v x[4];
#pragma unroll
          for (int k = 0; k < 4; k++) x[k] = src[min(i0 + k * 64u + lane, hi - 1u)];
#pragma unroll
          for (int k = 0; k < 4; k++) touched ^= x[k].x ^ x[k].y ^ x[k].z ^ x[k].w;
        }
      };
      if constexpr (BASE == 2) {
        plane(yp, ((qh * 2 - 1) * sy + p.sdr.w) * 4);
      } else {
        plane(yp, (qh * 2 - 1) * sy + p.sdr.w);
        const uint32_t crows = BASE == 0 ? qh : qh * 2, cw = BASE == 1 ? p.sdr.w : p.sdr.w / 2;
        plane(up, (crows - 1) * su + cw);
        plane(vp, (crows - 1) * sv + cw);
      }
      plane(mp, ((p.gm.h - 1) * sm + p.gm.w) * BPP);
    }
    stage_tables();
    if (!live) return;
    // this wave's quad rows: qy0, qy0 + groups, ... below qh (the look-ahead fetch past the last one re-reads the last row:
    // loads only, nothing is stored twice)
    const uint32_t my_iter = (qh - qy0 + groups - 1) / groups;
    auto rows = [&](auto GM) {
      for (uint32_t i = 0; i < my_iter; i++) {
        const Raw b = fetch(qy0 + i * groups, H1{});
        process(a, H0{}, GM);
        a = fetch(qy0 + (i + 1) * groups, H0{});
        process(b, H1{}, GM);
      }
    };
    if (p.sdr_gamut_on) rows(G1{});
    else if (p.hdr_gamut_on) rows(G2{});
    else rows(G0{});
    if (touched == 0x9e3779b9u && p.n_frames == 0xffffffffu) dp[0] = 0;  // never true: keeps the sweep's loads alive
  } else {
    // ---- coefficient input: 128 x 16 pixel tiles, IDCT into the wave's LDS tile, then eight quad rows -------
    stage_tables();
    __shared__ int s_ws[BLK / 64][8 * 8 * 9];
    __shared__ int s_q[3][64];
    __shared__ __attribute__((aligned(8))) uint8_t s_yt[BLK / 64][16 * 128];
    __shared__ __attribute__((aligned(8))) uint8_t s_ct[BLK / 64][2][8 * 64];
    const CoefSrc* __restrict__ cs = p.coef_src;
    if (tid < 192) s_q[tid >> 6][tid & 63] = cs->q[tid >> 6][tid & 63];
    __syncthreads();
    const int wv = (int)__builtin_amdgcn_readfirstlane(tid >> 6);
    int* ws = s_ws[wv];
    uint8_t* yt = s_yt[wv];
    uint8_t* cbt = s_ct[wv][0];
    uint8_t* crt = s_ct[wv][1];
    const int rr = (int)lane >> 3, rb = (int)lane & 7;
    const int16_t* cy = cs->coef[0];
    const int16_t* ccb = cs->coef[1];
    const int16_t* ccr = cs->coef[2];
    const int bw0 = cs->bw[0], bh0 = cs->bh[0], bw1 = cs->bw[1], bh1 = cs->bh[1], bw2 = cs->bw[2], bh2 = cs->bh[2];
    const uint32_t tiles_x = (p.sdr.w + 127) >> 7, tiles_y = (p.sdr.h + 15) >> 4, ntiles = tiles_x * tiles_y;
    const uint32_t nwaves = gridDim.x * (BLK / 64);
    auto tiles = [&](auto GM) {
    for (uint32_t t = wave; t < ntiles; t += nwaves) {
      const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
      // the tile's six eight-block units: Y rows 2ty, 2ty+1 x two halves, Cb, Cr (all loads issued up front)
      int v[6][8];
      int big[6] = {0, 0, 0, 0, 0, 0};
      {
        int q[8];
#pragma unroll
        for (int c = 0; c < 8; c++) q[c] = s_q[0][rr * 8 + c];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int by = (int)(2 * ty) + (u >> 1);
          idct::load_dequant_row(cy, bw0, by, (int)(tx * 16) + (u & 1) * 8 + rb, rr, q, v[u], big[u], by < bh0);
        }
#pragma unroll
        for (int c = 0; c < 8; c++) q[c] = s_q[1][rr * 8 + c];
        idct::load_dequant_row(ccb, bw1, (int)ty, (int)(tx * 8) + rb, rr, q, v[4], big[4], (int)ty < bh1);
#pragma unroll
        for (int c = 0; c < 8; c++) q[c] = s_q[2][rr * 8 + c];
        idct::load_dequant_row(ccr, bw2, (int)ty, (int)(tx * 8) + rb, rr, q, v[5], big[5], (int)ty < bh2);
      }
      const uint32_t xraw = tx * 128 + lane * 2;
      const bool lane_ok = xraw < p.sdr.w;
      col[0] = make_col(min(xraw, p.sdr.w - 2));
      Raw a = fetch(ty * 8, H0{});  // the first quad row's gain-map bytes are in flight during the transforms
#pragma unroll
      for (int u = 0; u < 6; u++) {
        uint32_t sm8[8];
        idct::idct_wave(ws, v[u], big[u], rr, rb, sm8);
        const uint32_t lo = sm8[0] | (sm8[1] << 8) | (sm8[2] << 16) | (sm8[3] << 24);
        const uint32_t hi = sm8[4] | (sm8[5] << 8) | (sm8[6] << 16) | (sm8[7] << 24);
        uint8_t* d = (u < 4) ? yt + ((u >> 1) * 8 + rr) * 128 + ((u & 1) * 8 + rb) * 8
                             : (u == 4 ? cbt : crt) + rr * 64 + rb * 8;
        *(uint2*)d = make_uint2(lo, hi);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): the tile is complete in LDS
#pragma unroll 1
      for (uint32_t qr = 0; qr < 8; qr++) {
        const uint32_t qy = ty * 8 + qr;
        Raw nxt = a;
        if (qr < 7) nxt = fetch(qy + 1, H0{});
        a.y0 = *(const uint16_t*)(yt + (2 * qr) * 128 + lane * 2);
        a.y1 = *(const uint16_t*)(yt + (2 * qr + 1) * 128 + lane * 2);
        a.u = cbt[qr * 64 + lane];
        a.v = crt[qr * 64 + lane];
        store_ok = lane_ok && (qy < qh);
        process(a, H0{}, GM);
        a = nxt;
      }
      __builtin_amdgcn_wave_barrier();  // the next tile overwrites the LDS tile
    }
    };
    if (p.sdr_gamut_on) tiles(G1{});
    else if (p.hdr_gamut_on) tiles(G2{});
    else tiles(G0{});
  }
}

// The two entry points differ in their SGPR budget only (the attribute takes a literal): see quad_sgprs
template <int OUT, int MAPFMT, int SMODE, int BASE, int SRC = 0>
__global__ __launch_bounds__(quad_block<OUT>()) __attribute__((amdgpu_num_sgpr(80), amdgpu_waves_per_eu(quad_waves_per_eu<OUT, SRC>(), 8))) void apply_quad_kernel(const ApplyParams p) {
  static_assert(quad_sgprs<SMODE, SRC>() == 80, "this entry point is the 80-SGPR one");
  apply_quad_body<OUT, MAPFMT, SMODE, BASE, SRC>(p);
}
template <int OUT, int MAPFMT, int SMODE, int BASE, int SRC = 0>
__global__ __launch_bounds__(quad_block<OUT>()) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(quad_waves_per_eu<OUT, SRC>(), 8))) void apply_quad_kernel_s96(const ApplyParams p) {
  static_assert(quad_sgprs<SMODE, SRC>() == 96, "this entry point is the 96-SGPR one");
  apply_quad_body<OUT, MAPFMT, SMODE, BASE, SRC>(p);
}
// The HLG / PQ variants that interpolate a THREE-channel map (OUT != 0, MAPFMT != 0, SMODE 1) do not fit the 80 VGPRs that six
// waves per SIMD (two 768-thread workgroups per CU) leave a wave: 60 - 164 bytes per lane went to scratch (round-4 review).
// This entry point asks for three waves per SIMD (one workgroup per CU, up to 168 VGPRs): no private segment.  It is the only
// form of those variants in the library: the six-wave spilling build lost by 3 - 25 % (profiles/r05_spill_wpe{3,6}.txt; the A/B
// arm is kept as tools/spill_exp.patch, not in the product).
template <int OUT, int MAPFMT, int SMODE, int BASE, int SRC = 0>
__global__ __launch_bounds__(quad_block<OUT>()) __attribute__((amdgpu_num_sgpr(96), amdgpu_waves_per_eu(3, 8))) void apply_quad_kernel_s96w3(const ApplyParams p) {
  static_assert(quad_sgprs<SMODE, SRC>() == 96 && OUT != 0 && MAPFMT != 0 && SMODE == 1 && SRC == 0, "the spilling HLG / PQ variants only");
  apply_quad_body<OUT, MAPFMT, SMODE, BASE, SRC>(p);
}

// Resident workgroups of a kernel on the current device = CUs x blocks per CU (occupancy API; the
// quad kernels cap their SGPRs so the API's answer is exact -- MI355X_MICROARCH.md "Residency").
template <typename K>
int resident_blocks(K kernel, int block = kBlock, int sgprs = 80) {
  int dev = 0, per_cu = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 2048;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, 0) != hipSuccess || per_cu <= 0) per_cu = block == kBlock ? 4 : 1;
  // 256-thread workgroups of a kernel that uses 81 - 96 SGPRs: the hardware admits 7 per CU where the API says 8
  // (MI355X_MICROARCH.md, "Residency": min(API, 8, 800 / (ceil(sgpr / 16) * 16 + 16))); the quad kernels declare 96
  if (block == kBlock && sgprs > 80 && per_cu > 7) per_cu = 7;
  if (per_cu > 8) per_cu = 8;
  if (const char* e = getenv("UHDR_HIP_BLOCKS_PER_CU")) {  // tuning knob (tools/kbench)
    const int v = atoi(e);
    if (v >= 1 && v <= 8) per_cu = v;
  }
  return per_cu * cus;
}

template <int OUT, int MAPFMT, int SMODE, int BASE>
hipError_t launch_quad(const ApplyParams& p, hipStream_t s) {
  constexpr int BLK = quad_block<OUT>();
  constexpr bool k80 = quad_sgprs<SMODE, 0>() == 80;
  constexpr bool kSpills = !k80 && OUT != 0 && MAPFMT != 0 && SMODE == 1;  // see apply_quad_kernel_s96w3
  static const int resident = [] {
    if constexpr (k80) return resident_blocks(apply_quad_kernel<OUT, MAPFMT, SMODE, BASE>, BLK, 80);
    else if constexpr (kSpills) return resident_blocks(apply_quad_kernel_s96w3<OUT, MAPFMT, SMODE, BASE>, BLK, 96);
    else return resident_blocks(apply_quad_kernel_s96<OUT, MAPFMT, SMODE, BASE>, BLK, 96);
  }();
  const uint32_t n_frames = p.n_frames ? p.n_frames : 1;
  const uint32_t strips_x = (p.sdr.w / 2 + 64 * kQuadsPerLane - 1) / (64 * kQuadsPerLane), qh = p.sdr.h / 2;
  // one balanced round: all workgroups resident; a wave owns a column strip of one frame and every
  // `groups`-th quad row of it
  // Row groups: `over` times the resident waves.  With exactly the resident waves (over = 1) every wave walks qh / groups
  // rows in lock step and the launch ends when the slowest one does; a grid a few times larger lets the dispatcher refill
  // a CU as soon as one of its workgroups retires (tools/ubench6 on the bare access pattern, 8K: map C 77 -> 75 -> 73 us
  // at over = 1 / 2 / 4).  The price is the table staging of the extra workgroups (L2 -> LDS, overlapped with the
  // streaming of the resident ones).
  static const uint32_t over = [] {
    const char* e = getenv("UHDR_HIP_OVERSUB");
    const int v = e ? atoi(e) : kOversub;
    return (uint32_t)(v >= 1 && v <= 16 ? v : kOversub);
  }();
  // prefetcher workgroups (apply_quad_body): single 4:2:0 frames with a subsampled map -- the write-dominated launches whose cold
  // inputs cost most -- whose inputs the host layer does not expect in the infinity cache.  UHDR_HIP_PREFETCH_WGS=<n> overrides
  // (0: off).
  static const int pf_env = [] { const char* e = getenv("UHDR_HIP_PREFETCH_WGS"); return e ? atoi(e) : -1; }();
  uint32_t pf = 0;
  if (BASE == 0 && SMODE == 1 && n_frames == 1 && (size_t)p.sdr.w * p.sdr.h >= (size_t)3840 * 2160 && p.sdr.stride[1] == p.sdr.stride[2] &&
      (!p.inputs_hot || pf_env > 0)) {
    // a prefetcher's rate is its CU's miss rate (about one 128-byte line per 7 cycles), so the sweep has to be spread over
    // many CUs to stay ahead of the consumers: 8K, 6 rotating buffer sets: 76 us without, 65 / 62 us with 160 / 320 of 1792
    // workgroups; inputs that are still cached lose 6 us to the duplicate reads (54.6 -> 60.6), hence `inputs_hot`
    pf = pf_env >= 0 ? (uint32_t)pf_env : (uint32_t)resident * 5u / 28u;
    if (pf > (uint32_t)resident / 2u) pf = (uint32_t)resident / 2u;
  }
  const uint32_t max_waves = ((uint32_t)resident - pf) * (BLK / 64) * over;
  uint32_t groups = max_waves / (strips_x * n_frames);
  if (groups > qh) groups = qh;
  if (groups < 1) groups = 1;
  const uint32_t nwaves = groups * strips_x * n_frames;
  ApplyParams q = p;
  q.n_frames = n_frames;
  q.row_groups = groups;
  q.tiles_per_wave = (qh + groups - 1) / groups;  // quad rows per wave
  q.prefetch_wgs = pf;
  {
    // touch-ahead is opt-in (UHDR_HIP_TOUCH_AHEAD=1).  Measured at 8K with the Y400 map (51 MB in, 265 MB out): the bare
    // access pattern gains what the read burst promises (cold inputs 64 -> 55 us, tools/ubench8), the kernel only 76.6 ->
    // 73.6 us -- its launch lasts long enough for part of the touched lines to leave the caches again before their rows
    // come up -- and it costs 5 us when the inputs are already cached (59.9 -> 64.7 us); with a full-resolution map the
    // burst is 40 % of the traffic and the launch gets slower (80 -> 96 us).
    static const int force = [] { const char* e = getenv("UHDR_HIP_TOUCH_AHEAD"); return e ? atoi(e) : 0; }();
    q.touch_ahead = force > 0 ? 1u : 0u;
  }
  const int grid = (int)((nwaves + BLK / 64 - 1) / (BLK / 64)) + (int)pf;
  if constexpr (k80) {
    hipLaunchKernelGGL((apply_quad_kernel<OUT, MAPFMT, SMODE, BASE>), dim3(grid), dim3(BLK), 0, s, q);
  } else if constexpr (kSpills) {
    hipLaunchKernelGGL((apply_quad_kernel_s96w3<OUT, MAPFMT, SMODE, BASE>), dim3(grid), dim3(BLK), 0, s, q);
  } else {
    hipLaunchKernelGGL((apply_quad_kernel_s96<OUT, MAPFMT, SMODE, BASE>), dim3(grid), dim3(BLK), 0, s, q);
  }
  return hipGetLastError();
}
// coefficient input (SRC 1): resident workgroups, waves stride over the 128 x 16 pixel tiles
template <int OUT, int MAPFMT, int SMODE>
hipError_t launch_quad_coef(const ApplyParams& p, hipStream_t s) {
  constexpr int BLK = quad_block<OUT>();
  static const int resident = resident_blocks(apply_quad_kernel_s96<OUT, MAPFMT, SMODE, 0, 1>, BLK, 96);
  const uint32_t ntiles = ((p.sdr.w + 127) / 128) * ((p.sdr.h + 15) / 16);
  uint32_t grid = (ntiles + BLK / 64 - 1) / (BLK / 64);
  if (grid > (uint32_t)resident) grid = (uint32_t)resident;
  ApplyParams q = p;
  q.n_frames = 1;
  q.row_groups = 1;
  q.tiles_per_wave = 0;
  hipLaunchKernelGGL((apply_quad_kernel_s96<OUT, MAPFMT, SMODE, 0, 1>), dim3(grid), dim3(BLK), 0, s, q);
  return hipGetLastError();
}
template <int OUT, int MAPFMT>
hipError_t launch_quad_coef_s(const ApplyParams& p, int smode, hipStream_t s) {
  return smode == 0 ? launch_quad_coef<OUT, MAPFMT, 0>(p, s) : launch_quad_coef<OUT, MAPFMT, 1>(p, s);
}
template <int OUT>
hipError_t launch_quad_coef_m(const ApplyParams& p, int mapfmt, int smode, hipStream_t s) {
  switch (mapfmt) {
    case 0: return launch_quad_coef_s<OUT, 0>(p, smode, s);
    case 1: return launch_quad_coef_s<OUT, 1>(p, smode, s);
    default: return launch_quad_coef_s<OUT, 2>(p, smode, s);
  }
}

template <int OUT, int MAPFMT, int BASE>
hipError_t launch_quad_s(const ApplyParams& p, int smode, hipStream_t s) {
  return smode == 0 ? launch_quad<OUT, MAPFMT, 0, BASE>(p, s) : launch_quad<OUT, MAPFMT, 1, BASE>(p, s);
}
template <int OUT, int BASE>
hipError_t launch_quad_m(const ApplyParams& p, int mapfmt, int smode, hipStream_t s) {
  switch (mapfmt) {
    case 0: return launch_quad_s<OUT, 0, BASE>(p, smode, s);
    case 1: return launch_quad_s<OUT, 1, BASE>(p, smode, s);
    default: return launch_quad_s<OUT, 2, BASE>(p, smode, s);
  }
}
template <int OUT>
hipError_t launch_quad_b(const ApplyParams& p, int base, int mapfmt, int smode, hipStream_t s) {
  switch (base) {
    case 0: return launch_quad_m<OUT, 0>(p, mapfmt, smode, s);
    case 1: return launch_quad_m<OUT, 1>(p, mapfmt, smode, s);
    case 3: return launch_quad_m<OUT, 3>(p, mapfmt, smode, s);
    default: return launch_quad_m<OUT, 2>(p, mapfmt, smode, s);
  }
}

inline bool aligned_to(const void* ptr, size_t a) { return ((uintptr_t)ptr % a) == 0; }

}  // namespace

// Does the quad kernel's layout contract hold?  Returns its SMODE (0 / 1) or -1.
int apply_quad_mode(const ApplyParams& p) {
  const int out = p.out_ct == UHDR_CT_LINEAR ? 0 : (p.out_ct == UHDR_CT_HLG ? 1 : 2);
  const int mapfmt = p.gm.fmt == UHDR_IMG_FMT_8bppYCbCr400 ? 0 : (p.gm.fmt == UHDR_IMG_FMT_24bppRGB888 ? 1 : 2);
  const size_t out_bytes = out == 0 ? 8 : 4;
  // base layouts the quad kernel reads: 4:2:0, 4:2:2, 4:4:4 (chroma pairs as 16-bit loads), packed RGBA8888 (8-byte loads)
  bool base_ok = false;
  // the kernel addresses every base plane with 32-bit byte offsets (row * stride + column): rows + 1 covers the
  // clamped look-ahead row of the software pipeline
  auto fits32 = [&](int pl) { return (uint64_t)p.sdr.stride[pl] * ((uint64_t)p.sdr.h + 1) < 0xFFFFFFFFull; };
  if (p.sdr.fmt == UHDR_IMG_FMT_12bppYCbCr420 || p.sdr.fmt == UHDR_IMG_FMT_16bppYCbCr422) {
    base_ok = (p.sdr.stride[0] % 2 == 0) && aligned_to(p.sdr.p[0], 2) && fits32(0) && fits32(1) && fits32(2);
  } else if (p.sdr.fmt == UHDR_IMG_FMT_24bppYCbCr444) {
    base_ok = (p.sdr.stride[0] % 2 == 0) && (p.sdr.stride[1] % 2 == 0) && (p.sdr.stride[2] % 2 == 0) &&
              aligned_to(p.sdr.p[0], 2) && aligned_to(p.sdr.p[1], 2) && aligned_to(p.sdr.p[2], 2) && fits32(0) && fits32(1) && fits32(2);
  } else if (p.sdr.fmt == UHDR_IMG_FMT_32bppRGBA8888) {
    base_ok = (p.sdr.stride[0] % 2 == 0) && aligned_to(p.sdr.p[0], 8) && (uint64_t)p.sdr.stride[0] * 4 * p.sdr.h < 0xFFFFFFFFull;
  }
  const bool quad = base_ok && (p.sdr.w % 2 == 0) && (p.sdr.h % 2 == 0) &&
                    (p.y0 % 2 == 0) &&
                    aligned_to(p.dst.p[0], 16) && ((p.dst.stride[0] * out_bytes) % 16 == 0) &&
                    p.sdr.w < 65536 && (p.sdr.h + p.y0) < 65536 && p.sdr.w >= 128 &&
                    // 32-bit byte offsets inside the kernel
                    (uint64_t)p.dst.stride[0] * out_bytes * p.sdr.h < 0xFFFFFFFFull &&
                    (uint64_t)p.gm.stride[0] * p.gm.h * 4 < 0xFFFFFFFFull;
  if (!quad) return -1;
  if (out != 0 && (!p.oetf_buckets || p.oetf_n == 0)) return -1;  // HLG / PQ: the verified output-code bucket table
  if (p.scale == 1) {
    // the gain map must cover every base pixel and allow the vector loads used per format
    if (p.gm.w < p.sdr.w || p.gm.h < p.sdr.h + p.y0) return -1;
    if (mapfmt == 0 && !(p.gm.stride[0] % 2 == 0 && aligned_to(p.gm.p[0], 2))) return -1;
    if (mapfmt == 1 && !((p.gm.stride[0] * 3) % 2 == 0 && aligned_to(p.gm.p[0], 2))) return -1;
    if (mapfmt == 2 && !(p.gm.stride[0] % 2 == 0 && aligned_to(p.gm.p[0], 8))) return -1;
    return 0;
  }
  if (p.scale >= 2 && p.scale % 2 == 0 && p.scale <= (uint32_t)kMaxIdwScaleLds && p.gamma_is_one[0] &&
      p.gamma_is_one[1] && p.gamma_is_one[2])
    return 1;  // gamma != 1 needs pow() per sample: generic kernel
  return -1;
}

// Base image in coefficient form (p.coef_src, a device CoefSrc; p.sdr carries the geometry of the 4:2:0 image the
// coefficients decode to).  Only the quad kernel has this input: hipErrorInvalidValue when its contract does not hold.
hipError_t launch_apply_gainmap_coef(const ApplyParams& p, hipStream_t s) {
  const int out = p.out_ct == UHDR_CT_LINEAR ? 0 : (p.out_ct == UHDR_CT_HLG ? 1 : 2);
  const int mapfmt = p.gm.fmt == UHDR_IMG_FMT_8bppYCbCr400 ? 0 : (p.gm.fmt == UHDR_IMG_FMT_24bppRGB888 ? 1 : 2);
  const int smode = apply_quad_mode(p);
  if (smode < 0 || p.sdr.fmt != UHDR_IMG_FMT_12bppYCbCr420 || !p.coef_src || p.n_frames > 1) return hipErrorInvalidValue;
  switch (out) {
    case 0: return launch_quad_coef_m<0>(p, mapfmt, smode, s);
    case 1: return launch_quad_coef_m<1>(p, mapfmt, smode, s);
    default: return launch_quad_coef_m<2>(p, mapfmt, smode, s);
  }
}

// Picks the quad kernel when its layout assumptions hold, otherwise the generic one.
// Batch mode (p.n_frames > 1) exists on the quad path only; the host layer falls back to one
// launch per frame when apply_quad_mode() says no.
hipError_t launch_apply_gainmap(const ApplyParams& p, hipStream_t s) {
  const int out = p.out_ct == UHDR_CT_LINEAR ? 0 : (p.out_ct == UHDR_CT_HLG ? 1 : 2);
  const int mapfmt = p.gm.fmt == UHDR_IMG_FMT_8bppYCbCr400 ? 0 : (p.gm.fmt == UHDR_IMG_FMT_24bppRGB888 ? 1 : 2);
  const int smode = apply_quad_mode(p);
  if (smode >= 0) {
    const int base = p.sdr.fmt == UHDR_IMG_FMT_12bppYCbCr420 ? 0 : (p.sdr.fmt == UHDR_IMG_FMT_24bppYCbCr444 ? 1 :
                     (p.sdr.fmt == UHDR_IMG_FMT_16bppYCbCr422 ? 3 : 2));
    switch (out) {
      case 0: return launch_quad_b<0>(p, base, mapfmt, smode, s);
      case 1: return launch_quad_b<1>(p, base, mapfmt, smode, s);
      default: return launch_quad_b<2>(p, base, mapfmt, smode, s);
    }
  }
  if (p.n_frames > 1) return hipErrorInvalidValue;
  const size_t total = (size_t)((p.sdr.w + kBlock - 1) / kBlock) * p.sdr.h;  // tiles
  int grid = (int)min(total, (size_t)2048);
  if (grid < 1) grid = 1;
  switch (out) {
    case 0: hipLaunchKernelGGL((apply_generic_kernel<0>), dim3(grid), dim3(kBlock), 0, s, p); break;
    case 1: hipLaunchKernelGGL((apply_generic_kernel<1>), dim3(grid), dim3(kBlock), 0, s, p); break;
    default: hipLaunchKernelGGL((apply_generic_kernel<2>), dim3(grid), dim3(kBlock), 0, s, p); break;
  }
  return hipGetLastError();
}

}  // namespace uhdr
