// generateGainMap on gfx950: SDR + HDR renditions -> 8-bit log-ratio gain map.
// Reference loops: /root/reference/lib/src/jpegr.cpp:753-818 (one pass), 866-931 + 992-1013 (two
// pass) with encodeGain / computeGain / affineMapGain from lib/src/gainmapmath.cpp:753-789.
//
// One thread per map pixel (each reads its s x s box of both images once: the minimum traffic); a
// workgroup walks tiles of 256 consecutive map pixels of one row.  The kernel is instantiated per
// (SDR format, HDR format) so the pixel unpack has no per-pixel format dispatch.  All look-up
// tables live in LDS:
//   sRGB inverse OETF (1024 floats), HDR inverse OETF (4096 floats; for HLG the host has already
//   folded hlgOotfApprox's powf(x, 1.2f) into it -- an exact fusion, the composite is evaluated with
//   the host libm at the 4096 table nodes), and the float64 tables of exact_math.h that replace the
//   per-pixel double log2.
// Two-pass mode keeps the float log2-gain plane in HBM between the passes -- the reference does the
// same (jpegr.cpp:842-844) -- and reduces per-channel min/max with wavefront shuffles -> LDS ->
// one partial per workgroup -> a single-workgroup final reduction (deterministic, no float
// atomics).  Across GPUs the 6 floats are all-reduced by the host layer (RCCL MIN / MAX).
#include "encode_core.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;
constexpr int kGenBlock = 512;  // generate_kernel: 8 waves share one 29 KB table set -> 3 workgroups = 24 waves per CU (VGPR limit)
constexpr int kMaxGrid = 2048;  // the host layer sizes the partials buffer for this many workgroups

struct GenLds {
  float srgb[kSrgbN];
  float hdr[kInvOetfN];
  double math[kMathTabDoubles];
  UnormTables unorm;  // x / 255.0f, x / 1023.0f
};

// QUAD: 4:2:0 SDR + P010 HDR at scale 1 with even geometry -- one thread per 2x2 quad, both images read
// with the coalesced quad fetches of pixel_io.h (luma as one vector load per row, chroma once per quad)
template <int SDRF, int HDRF, bool TWO_PASS, bool QUAD>
__global__ __launch_bounds__(kGenBlock) void generate_kernel(const GenParams p, float* partials) {
  __shared__ GenLds L;
  __shared__ uint2 s_gain8[TWO_PASS ? 1 : kStepTabMax];  // one pass: clamped gain -> map byte (host_tables.cpp)
  const uint32_t tid = threadIdx.x;
  if constexpr (!TWO_PASS) stage_step_tab(s_gain8, p.gain8, tid, kGenBlock);
  for (uint32_t i = tid; i < kSrgbN; i += kGenBlock) L.srgb[i] = p.srgb_lut[i];
  if (p.hdr_inv_lut)
    for (uint32_t i = tid; i < (uint32_t)p.hdr_inv_n; i += kGenBlock) L.hdr[i] = p.hdr_inv_lut[i];
  for (uint32_t i = tid; i < kMathTabDoubles; i += kGenBlock) L.math[i] = p.math_tab[i];
  fill_unorm_tables(L.unorm, tid, kGenBlock);
  __syncthreads();

  const bool hdr_lut = p.hdr_inv_lut != nullptr, hdr_lut_4096 = p.hdr_inv_n == kInvOetfN;
  float mn[3] = {127.0f, 127.0f, 127.0f}, mx[3] = {-128.0f, -128.0f, -128.0f};
  // one map pixel from the two samples as fetch_pixel / sample_box deliver them
  auto do_pixel = [&](Color3 s, Color3 h, uint32_t x, uint32_t y) {
    if (!p.sdr_is_rgb) s = yuv_to_rgb(s.r, s.g, s.b, p.sdr_yuv);
    Color3 sl = {L.srgb[lut_index_f32<kSrgbN>(s.r)], L.srgb[lut_index_f32<kSrgbN>(s.g)], L.srgb[lut_index_f32<kSrgbN>(s.b)]};
    if (p.sdr_gamut_on) sl = mat3_apply(sl, p.sdr_gamut);
    sl.r = clip_neg(sl.r); sl.g = clip_neg(sl.g); sl.b = clip_neg(sl.b);
    if (!p.hdr_is_rgb) h = yuv_to_rgb(h.r, h.g, h.b, p.hdr_yuv);
    Color3 hl = linearise_hdr(h, L.hdr, hdr_lut, hdr_lut_4096);
    if (p.hdr_gamut_on) hl = mat3_apply(hl, p.hdr_gamut);
    hl.r = clip_neg(hl.r); hl.g = clip_neg(hl.g); hl.b = clip_neg(hl.b);
    gain_of_pixel<TWO_PASS>(sl, hl, p, L.math, x, y, mn, mx, TWO_PASS ? nullptr : s_gain8);
  };
  if constexpr (QUAD) {
    const uint32_t qw = p.map_w / 2, qh = p.map_h / 2;
    const uint32_t tiles_x = (qw + kGenBlock - 1) / kGenBlock, tiles = tiles_x * qh;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
      const uint32_t qy = t / tiles_x, qx = (t - qy * tiles_x) * kGenBlock + tid;
      if (qx >= qw) continue;
      const QuadYuv sq = fetch_quad_420(p.sdr, qx, qy);
      const QuadYuv hq = fetch_quad_p010(p.hdr, qx, qy, &L.unorm);
#pragma unroll
      for (int k = 0; k < 4; k++) do_pixel(sq.px[k], hq.px[k], 2 * qx + (k & 1), 2 * qy + (k >> 1));
    }
  } else {
    const uint32_t mw = p.map_w, mh = p.map_h;
    const uint32_t tiles_x = (mw + kGenBlock - 1) / kGenBlock, tiles = tiles_x * mh;
    for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
      const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kGenBlock + tid;
      if (x >= mw) continue;
      do_pixel(sample_box<SDRF>(p.sdr, p.scale, x, y, &L.unorm), sample_box<HDRF>(p.hdr, p.scale, x, y, &L.unorm), x, y);
    }
  }
  if constexpr (TWO_PASS) reduce_block_minmax<kGenBlock>(mn, mx, partials);
}

__global__ void reduce_minmax_kernel(const float* partials, int n, float* out6) {
  __shared__ float s_red[4][6];
  float mn[3] = {127.0f, 127.0f, 127.0f}, mx[3] = {-128.0f, -128.0f, -128.0f};
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    for (int c = 0; c < 3; c++) {
      mn[c] = fminf(mn[c], partials[(size_t)i * 6 + c]);
      mx[c] = fmaxf(mx[c], partials[(size_t)i * 6 + 3 + c]);
    }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int c = 0; c < 3; c++) {
    const float a = wave_min(mn[c]), b = wave_max(mx[c]);
    if (lane == 0) { s_red[wv][c] = a; s_red[wv][3 + c] = b; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    float v = s_red[0][threadIdx.x];
    for (int k = 1; k < 4; k++)
      v = threadIdx.x < 3 ? fminf(v, s_red[k][threadIdx.x]) : fmaxf(v, s_red[k][threadIdx.x]);
    out6[threadIdx.x] = v;
  }
}

// affineMapGain (gainmapmath.cpp:784-789) over the float plane (jpegr.cpp:992-1013): one thread per
// four consecutive samples of a row (16-byte load, 4-byte store) when the geometry allows
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void affine_kernel(const AffineParams p) {
  const uint32_t row_elems = p.map_w * p.nch;
  const uint32_t per_row = VEC4 ? row_elems / 4 : row_elems;
  const uint32_t tiles_x = (per_row + kBlock - 1) / kBlock, tiles = tiles_x * p.map_h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, j = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (j >= per_row) continue;
    const float* src = p.gain_log2 + (size_t)y * row_elems;
    uint8_t* dst = p.out + (size_t)y * p.out_stride * p.nch;
    auto map1 = [&](float g, uint32_t e) -> uint32_t {
      const uint32_t c = p.nch == 3 ? e % 3 : 0;
      const float mn_c = p.dev ? p.dev->mn[c] : p.mn[c];
      const double rr_c = p.dev ? p.dev->range_rcp[c] : p.range_rcp[c];
      float m = div_by_rcp64(g - mn_c, rr_c);  // (g - min) / (max - min), exact (device_math.h)
      if (p.gamma != 1.0f) m = (float)pow((double)m, (double)p.gamma);
      m *= 255.0f;
      float t2 = m + 0.5f;
      t2 = (t2 < 0.0f) ? 0.0f : ((t2 > 255.0f) ? 255.0f : t2);
      return (uint32_t)t2;
    };
    if constexpr (VEC4) {
      const float4 g = *(const float4*)(src + j * 4);
      const uint32_t e = j * 4;
      *(uint32_t*)(dst + e) = map1(g.x, e) | (map1(g.y, e + 1) << 8) | (map1(g.z, e + 2) << 16) | (map1(g.w, e + 3) << 24);
    } else {
      dst[j] = (uint8_t)map1(src[j], j);
    }
  }
}

// Wide variant: one thread maps 16 consecutive map pixels of a row (48 samples for a 3-channel map: twelve
// 16-byte loads in flight per lane, three 16-byte nontemporal stores; the channel of every sample is a
// compile-time constant).  The narrow kernel above has one load and one store per thread and is bound by
// memory latency.
template <int NCH>
__global__ __launch_bounds__(kBlock) void affine_wide_kernel(const AffineParams p) {
  constexpr int NS = 16 * NCH;  // samples per thread
  const uint32_t row_elems = p.map_w * NCH, per_row = row_elems / NS;
  const uint32_t total = per_row * p.map_h, tiles = (total + kBlock - 1) / kBlock;  // flat: narrow maps still fill the lanes
  const float* pmn = p.dev ? p.dev->mn : p.mn;  // wave-uniform: scalar loads either way
  const double* prr = p.dev ? p.dev->range_rcp : p.range_rcp;
  const float mn[3] = {pmn[0], pmn[NCH == 3 ? 1 : 0], pmn[NCH == 3 ? 2 : 0]};
  const double rr[3] = {prr[0], prr[NCH == 3 ? 1 : 0], prr[NCH == 3 ? 2 : 0]};
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t idx = t * kBlock + threadIdx.x;
    if (idx >= total) continue;
    const uint32_t y = idx / per_row, j = idx - y * per_row;
    const float4* src = (const float4*)(p.gain_log2 + (size_t)y * row_elems + (size_t)j * NS);
    float4 g[NS / 4];
#pragma unroll
    for (int k = 0; k < NS / 4; k++) g[k] = src[k];
    uint32_t o[NS / 4];
#pragma unroll
    for (int k = 0; k < NS / 4; k++) {
      const float v[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int c = (4 * k + i) % NCH;
        float m = div_by_rcp64(v[i] - mn[c], rr[c]);  // (g - min) / (max - min), exact (device_math.h)
        if (p.gamma != 1.0f) m = (float)pow((double)m, (double)p.gamma);
        m *= 255.0f;
        float t2 = m + 0.5f;
        t2 = (t2 < 0.0f) ? 0.0f : ((t2 > 255.0f) ? 255.0f : t2);
        w |= (uint32_t)t2 << (8 * i);
      }
      o[k] = w;
    }
    uint8_t* dst = p.out + (size_t)y * p.out_stride * NCH + (size_t)j * NS;
#pragma unroll
    for (int k = 0; k < NS / 16; k++)
      __builtin_nontemporal_store((u4v){o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]}, (u4v*)(dst + 16 * k));
  }
}

int gen_grid(uint32_t tiles) {
  static const int resident = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 1024;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus * 3;  // 512-thread workgroups, 29 KB of LDS tables each: three are resident per CU (24 waves, the VGPR limit)
  }();
  uint32_t g = tiles < (uint32_t)resident ? tiles : (uint32_t)resident;
  if (g > kMaxGrid) g = kMaxGrid;
  if (g < 1) g = 1;
  return (int)g;
}

template <int SDRF, int HDRF>
void launch_gen(const GenParams& p, bool two_pass, int grid, float* partials, hipStream_t s) {
  if (two_pass) hipLaunchKernelGGL((generate_kernel<SDRF, HDRF, true, false>), dim3(grid), dim3(kGenBlock), 0, s, p, partials);
  else hipLaunchKernelGGL((generate_kernel<SDRF, HDRF, false, false>), dim3(grid), dim3(kGenBlock), 0, s, p, partials);
}
template <int SDRF>
void launch_gen_h(const GenParams& p, bool two_pass, int grid, float* partials, hipStream_t s) {
  switch (p.hdr.fmt) {
    case UHDR_IMG_FMT_24bppYCbCrP010: return launch_gen<SDRF, UHDR_IMG_FMT_24bppYCbCrP010>(p, two_pass, grid, partials, s);
    case UHDR_IMG_FMT_32bppRGBA1010102: return launch_gen<SDRF, UHDR_IMG_FMT_32bppRGBA1010102>(p, two_pass, grid, partials, s);
    default: return launch_gen<SDRF, -1>(p, two_pass, grid, partials, s);
  }
}

}  // namespace

// Two-pass: p.minmax must have room for 6 floats followed by kMaxGrid*6 floats of partials
// (the host layer allocates 6 + 2048*6).
hipError_t launch_generate_gainmap(const GenParams& p, bool two_pass, hipStream_t s) {
  float* partials = two_pass ? p.minmax + 6 : nullptr;
  if (p.scale == 1 && p.sdr.fmt == UHDR_IMG_FMT_12bppYCbCr420 && p.hdr.fmt == UHDR_IMG_FMT_24bppYCbCrP010 &&
      quad_layout_ok(p.sdr) && quad_layout_ok(p.hdr)) {  // the API-1 default: quad lanes, coalesced plane loads
    const uint32_t qtiles = ((p.map_w / 2 + kGenBlock - 1) / kGenBlock) * (p.map_h / 2);
    const int qgrid = gen_grid(qtiles);
    constexpr int S = UHDR_IMG_FMT_12bppYCbCr420, H = UHDR_IMG_FMT_24bppYCbCrP010;
    if (two_pass) {
      hipLaunchKernelGGL((generate_kernel<S, H, true, true>), dim3(qgrid), dim3(kGenBlock), 0, s, p, partials);
      hipLaunchKernelGGL(reduce_minmax_kernel, dim3(1), dim3(256), 0, s, (const float*)partials, qgrid, p.minmax);
    } else {
      hipLaunchKernelGGL((generate_kernel<S, H, false, true>), dim3(qgrid), dim3(kGenBlock), 0, s, p, partials);
    }
    return hipGetLastError();
  }
  const uint32_t tiles = ((p.map_w + kGenBlock - 1) / kGenBlock) * p.map_h;
  const int grid = gen_grid(tiles);
  switch (p.sdr.fmt) {
    case UHDR_IMG_FMT_12bppYCbCr420: launch_gen_h<UHDR_IMG_FMT_12bppYCbCr420>(p, two_pass, grid, partials, s); break;
    case UHDR_IMG_FMT_32bppRGBA8888: launch_gen_h<UHDR_IMG_FMT_32bppRGBA8888>(p, two_pass, grid, partials, s); break;
    default: launch_gen_h<-1>(p, two_pass, grid, partials, s); break;
  }
  if (two_pass) hipLaunchKernelGGL(reduce_minmax_kernel, dim3(1), dim3(256), 0, s, (const float*)partials, grid, p.minmax);
  return hipGetLastError();
}

// ---- striped two-pass generation: the exchange step on the device ------------------------------------------------------
// {min0..2, max0..2} -> {min0..2, -max0..2}: negating the maxima turns the per-channel min AND max merge of
// jpegr.cpp:932-938 into ONE elementwise minimum, i.e. a single all-reduce(min) over 6 floats.  `empty`: this rank's
// stripe holds no map sample; it contributes the identity of the merge (the reference's initial values 127 / -128).
__global__ void minmax_pack_kernel(const float* mm6, float* merged6, int empty) {
  const int i = threadIdx.x;
  if (i < 3) merged6[i] = empty ? 127.0f : mm6[i];
  else if (i < 6) merged6[i] = empty ? 128.0f : -mm6[i];
}
// jpegr.cpp:969-986: clamp to [-14.3, 15.6], the user's min / max content-boost hints, the epsilon guard; then the
// affine map's per-channel constants exactly as the host computes them (float subtraction, float64 reciprocal).
__global__ void minmax_finalize_kernel(const FinalizeParams p) {
  const int i = threadIdx.x;
  if (i >= 3) return;
  float gmin = p.merged[i], gmax = -p.merged[3 + i];
  if (i < p.nch) {
    gmin = gmin < -14.3f ? -14.3f : (gmin > 15.6f ? 15.6f : gmin);
    gmax = gmax < -14.3f ? -14.3f : (gmax > 15.6f ? 15.6f : gmax);
    if (p.has_max_hint) gmax = gmax < p.log2_max_hint ? gmax : p.log2_max_hint;
    if (p.has_min_hint) gmin = gmin < p.log2_min_hint ? p.log2_min_hint : gmin;
    if (fabsf(gmax - gmin) < 1.1920928955078125e-07f) gmax += 0.1f;  // FLT_EPSILON
  }
  p.out->mn[i] = gmin;
  p.out->mx[i] = gmax;
  p.out->range_rcp[i] = 1.0 / (double)(gmax - gmin);
  p.out_mm[i] = gmin;
  p.out_mm[3 + i] = gmax;
}
hipError_t launch_minmax_pack(const float* mm6, float* merged6, int empty, hipStream_t s) {
  hipLaunchKernelGGL(minmax_pack_kernel, dim3(1), dim3(64), 0, s, mm6, merged6, empty);
  return hipGetLastError();
}
hipError_t launch_minmax_finalize(const FinalizeParams& p, hipStream_t s) {
  hipLaunchKernelGGL(minmax_finalize_kernel, dim3(1), dim3(64), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_reduce_minmax(const float* partials, int n, float* out6, hipStream_t s) {
  hipLaunchKernelGGL(reduce_minmax_kernel, dim3(1), dim3(256), 0, s, partials, n, out6);
  return hipGetLastError();
}

hipError_t launch_affine_map(const AffineParams& p, hipStream_t s) {
  const uint32_t row_elems = p.map_w * p.nch;
  if ((p.nch == 1 || p.nch == 3) && p.map_w % 16 == 0 && ((size_t)p.out_stride * p.nch) % 16 == 0 && (((uintptr_t)p.out & 15) == 0) &&
      (((uintptr_t)p.gain_log2 & 15) == 0)) {
    const uint32_t total = (p.map_w / 16) * p.map_h;
    const uint32_t tiles = (total + kBlock - 1) / kBlock;
    const int grid = (int)(tiles < 8192u ? (tiles ? tiles : 1u) : 8192u);
    if (p.nch == 3) hipLaunchKernelGGL((affine_wide_kernel<3>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((affine_wide_kernel<1>), dim3(grid), dim3(kBlock), 0, s, p);
    return hipGetLastError();
  }
  const bool vec4 = (row_elems % 4 == 0) && ((p.out_stride * p.nch) % 4 == 0) && (((uintptr_t)p.out & 3) == 0) &&
                    (((uintptr_t)p.gain_log2 & 15) == 0);
  const uint32_t per_row = vec4 ? row_elems / 4 : row_elems;
  const uint32_t tiles = ((per_row + kBlock - 1) / kBlock) * p.map_h;
  const int grid = (int)(tiles < 8192u ? (tiles ? tiles : 1u) : 8192u);
  if (vec4) hipLaunchKernelGGL((affine_kernel<true>), dim3(grid), dim3(kBlock), 0, s, p);
  else hipLaunchKernelGGL((affine_kernel<false>), dim3(grid), dim3(kBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace uhdr
