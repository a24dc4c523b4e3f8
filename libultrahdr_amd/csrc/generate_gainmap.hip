// generateGainMap on gfx950: SDR + HDR renditions -> 8-bit log-ratio gain map.
// Reference loops: /root/reference/lib/src/jpegr.cpp:753-818 (one pass), 866-931 + 992-1013 (two
// pass) with encodeGain / computeGain / affineMapGain from lib/src/gainmapmath.cpp:753-789.
//
// One thread per map pixel (each reads its s x s box of both images once: the minimum traffic).
// Two-pass mode keeps the float log2-gain plane in HBM between the passes -- the reference does the
// same (jpegr.cpp:842-844) -- and reduces per-channel min/max with wavefront shuffles -> LDS ->
// one partial per workgroup -> a single-workgroup final reduction (deterministic, no float
// atomics).  Across GPUs the 6 floats are all-reduced by the host layer (RCCL MIN / MAX).
#include "pixel_io.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ Color3 inv_oetf(Color3 e, const GenParams& p) {
  if (!p.hdr_inv_lut) return e;  // linear input: identityConversion
  Color3 o;
  if (p.hdr_inv_n == kInvOetfN) {  // 4096-entry HLG / PQ tables: exact double-form index
    o.r = p.hdr_inv_lut[lut_index_f64<kInvOetfN>(e.r)];
    o.g = p.hdr_inv_lut[lut_index_f64<kInvOetfN>(e.g)];
    o.b = p.hdr_inv_lut[lut_index_f64<kInvOetfN>(e.b)];
  } else {
    o.r = p.hdr_inv_lut[lut_index_f32<kSrgbN>(e.r)];
    o.g = p.hdr_inv_lut[lut_index_f32<kSrgbN>(e.g)];
    o.b = p.hdr_inv_lut[lut_index_f32<kSrgbN>(e.b)];
  }
  return o;
}

// encodeGain (gainmapmath.cpp:758-771): log2 is the DOUBLE libm one in the reference build, the
// normalisation is double arithmetic narrowed to float, then powf, then truncation.
__device__ __forceinline__ uint8_t encode_gain(float y_sdr, float y_hdr, const GenParams& p) {
  float gain = 1.0f;
  if (y_sdr > 0.0f) gain = y_hdr / y_sdr;
  if (gain < p.min_boost) gain = p.min_boost;
  if (gain > p.max_boost) gain = p.max_boost;
  const float n = (float)((log2((double)gain) - (double)p.log2min) / (double)(p.log2max - p.log2min));
  const float ng = (p.gamma == 1.0f) ? n : powf(n, p.gamma);  // powf(x, 1) == x exactly
  return (uint8_t)(ng * 255.0f);
}
// computeGain (gainmapmath.cpp:773-782)
__device__ __forceinline__ float compute_gain(float sdr, float hdr) {
  float gain = (float)log2((double)((hdr + 1e-7f) / (sdr + 1e-7f)));
  if (sdr < 2.f / 255.0f) gain = fminf(gain, 2.3f);
  return gain;
}

template <bool TWO_PASS>
__global__ __launch_bounds__(kBlock) void generate_kernel(const GenParams p, float* partials) {
  const uint32_t mw = p.map_w, mh = p.map_h;
  const size_t total = (size_t)mw * mh;
  float mn[3] = {127.0f, 127.0f, 127.0f}, mx[3] = {-128.0f, -128.0f, -128.0f};
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const uint32_t y = (uint32_t)(i / mw), x = (uint32_t)(i - (size_t)y * mw);
    Color3 s = sample_box(p.sdr, p.scale, x, y);
    if (!p.sdr_is_rgb) s = yuv_to_rgb(s.r, s.g, s.b, p.sdr_yuv);
    Color3 sl = {p.srgb_lut[lut_index_f32<kSrgbN>(s.r)], p.srgb_lut[lut_index_f32<kSrgbN>(s.g)],
                 p.srgb_lut[lut_index_f32<kSrgbN>(s.b)]};
    if (p.sdr_gamut_on) sl = mat3_apply(sl, p.sdr_gamut);
    sl.r = clip_neg(sl.r); sl.g = clip_neg(sl.g); sl.b = clip_neg(sl.b);

    Color3 h = sample_box(p.hdr, p.scale, x, y);
    if (!p.hdr_is_rgb) h = yuv_to_rgb(h.r, h.g, h.b, p.hdr_yuv);
    Color3 hl = inv_oetf(h, p);
    if (p.hdr_is_hlg) {  // hlgOotfApprox: powf(x, 1.2f) per channel (gainmapmath.cpp:293-295)
      hl.r = powf(hl.r, 1.2f); hl.g = powf(hl.g, 1.2f); hl.b = powf(hl.b, 1.2f);
    }
    if (p.hdr_gamut_on) hl = mat3_apply(hl, p.hdr_gamut);
    hl.r = clip_neg(hl.r); hl.g = clip_neg(hl.g); hl.b = clip_neg(hl.b);

    if (p.multichannel) {
      const float sn[3] = {sl.r * 203.0f, sl.g * 203.0f, sl.b * 203.0f};
      const float hn[3] = {hl.r * p.hdr_nits, hl.g * p.hdr_nits, hl.b * p.hdr_nits};
      if constexpr (!TWO_PASS) {
        uint8_t* o = p.out + ((size_t)x + (size_t)y * p.out_stride) * 3;
        o[0] = encode_gain(sn[0], hn[0], p);
        o[1] = encode_gain(sn[1], hn[1], p);
        o[2] = encode_gain(sn[2], hn[2], p);
      } else {
        float* o = p.gain_log2 + i * 3;
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float v = compute_gain(sn[c], hn[c]);
          o[c] = v;
          mn[c] = fminf(mn[c], v);
          mx[c] = fmaxf(mx[c], v);
        }
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
        p.out[(size_t)x + (size_t)y * p.out_stride] = encode_gain(sy, hy, p);
      } else {
        const float v = compute_gain(sy, hy);
        p.gain_log2[i] = v;
        mn[0] = fminf(mn[0], v);
        mx[0] = fmaxf(mx[0], v);
      }
    }
  }
  if constexpr (TWO_PASS) {
    __shared__ float s_red[kBlock / 64][6];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float a = wave_min(mn[c]), b = wave_max(mx[c]);
      if (lane == 0) { s_red[wv][c] = a; s_red[wv][3 + c] = b; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
      float v = s_red[0][threadIdx.x];
      for (int k = 1; k < kBlock / 64; k++)
        v = threadIdx.x < 3 ? fminf(v, s_red[k][threadIdx.x]) : fmaxf(v, s_red[k][threadIdx.x]);
      partials[(size_t)blockIdx.x * 6 + threadIdx.x] = v;
    }
  }
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

// affineMapGain (gainmapmath.cpp:784-789) over the float plane (jpegr.cpp:992-1013)
__global__ __launch_bounds__(kBlock) void affine_kernel(const AffineParams p) {
  const size_t row_elems = (size_t)p.map_w * p.nch;
  const size_t total = row_elems * p.map_h;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const size_t y = i / row_elems, e = i - y * row_elems;
    const int c = (int)(e % p.nch);
    float m = (p.gain_log2[i] - p.mn[c]) / (p.mx[c] - p.mn[c]);
    if (p.gamma != 1.0f) m = (float)pow((double)m, (double)p.gamma);
    m *= 255.0f;
    float t = m + 0.5f;
    t = (t < 0.0f) ? 0.0f : ((t > 255.0f) ? 255.0f : t);
    p.out[y * (size_t)p.out_stride * p.nch + e] = (uint8_t)t;
  }
}

}  // namespace

static int gen_grid(size_t total) {
  size_t g = (total + kBlock - 1) / kBlock;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return (int)g;
}

// Two-pass: p.minmax must have room for 6 floats followed by gen_grid*6 floats of partials
// (the host layer allocates 6 + 2048*6).
hipError_t launch_generate_gainmap(const GenParams& p, bool two_pass, hipStream_t s) {
  const int grid = gen_grid((size_t)p.map_w * p.map_h);
  if (!two_pass) {
    hipLaunchKernelGGL((generate_kernel<false>), dim3(grid), dim3(kBlock), 0, s, p, (float*)nullptr);
    return hipGetLastError();
  }
  float* partials = p.minmax + 6;
  hipLaunchKernelGGL((generate_kernel<true>), dim3(grid), dim3(kBlock), 0, s, p, partials);
  hipLaunchKernelGGL(reduce_minmax_kernel, dim3(1), dim3(256), 0, s, (const float*)partials, grid, p.minmax);
  return hipGetLastError();
}

hipError_t launch_affine_map(const AffineParams& p, hipStream_t s) {
  const int grid = gen_grid((size_t)p.map_w * p.map_h * p.nch);
  hipLaunchKernelGGL(affine_kernel, dim3(grid), dim3(kBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace uhdr
