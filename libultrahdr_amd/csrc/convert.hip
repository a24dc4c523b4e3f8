// Pixel-format / YUV-encoding conversions of the hot path on gfx950.
//   transformYuv420 / transformYuv444  (/root/reference/lib/src/gainmapmath.cpp:686-748), driven by
//   UltraHdr::convertYuv (lib/src/jpegr.cpp:436-518) -- in place, 8-bit planar.
//   convert_raw_input_to_ycbcr         (lib/src/gainmapmath.cpp:1291-1482) -- RGBA1010102 -> P010 /
//   10-bit 4:4:4, RGBA8888 / RGB888 -> 4:2:0 / 4:4:4 (full range).
// Integer outputs: bit-exact against the reference is required and tested.
#include "pixel_io.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint8_t st8(float v) {  // static_cast<uint8_t>(CLIP3(v, 0, 255))
  v = (v < 0.0f) ? 0.0f : ((v > 255.0f) ? 255.0f : v);
  return (uint8_t)v;
}

// One thread per 2x2 quad.  In place is safe: a quad reads only its own 4 lumas + 1 chroma pair.
__global__ __launch_bounds__(kBlock) void transform_yuv420_kernel(const YuvXformParams p) {
  const uint32_t qw = p.img.w / 2, qh = p.img.h / 2;
  const uint32_t tiles_x = (qw + kBlock - 1) / kBlock, tiles = tiles_x * qh;
  uint8_t* yp = (uint8_t*)p.img.p[0];
  uint8_t* up = (uint8_t*)p.img.p[1];
  uint8_t* vp = (uint8_t*)p.img.p[2];
  const size_t sy = p.img.stride[0];
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t qy = t / tiles_x, qx = (t - qy * tiles_x) * kBlock + threadIdx.x;
    if (qx >= qw) continue;
    uint8_t* y0 = yp + (size_t)(qy * 2) * sy + qx * 2;
    uint8_t* y1 = y0 + sy;
    uint8_t* uq = up + (size_t)qy * p.img.stride[1] + qx;
    uint8_t* vq = vp + (size_t)qy * p.img.stride[2] + qx;
    const float u = (float)((int)*uq - 128) * (1 / 255.0f);
    const float v = (float)((int)*vq - 128) * (1 / 255.0f);
    const Color3 a = mat3_apply({(float)y0[0] * (1 / 255.0f), u, v}, p.c);
    const Color3 b = mat3_apply({(float)y0[1] * (1 / 255.0f), u, v}, p.c);
    const Color3 c = mat3_apply({(float)y1[0] * (1 / 255.0f), u, v}, p.c);
    const Color3 d = mat3_apply({(float)y1[1] * (1 / 255.0f), u, v}, p.c);
    const float nu = (((a.g + b.g) + c.g) + d.g) / 4.0f;
    const float nv = (((a.b + b.b) + c.b) + d.b) / 4.0f;
    y0[0] = st8(a.r * 255.0f + 0.5f);
    y0[1] = st8(b.r * 255.0f + 0.5f);
    y1[0] = st8(c.r * 255.0f + 0.5f);
    y1[1] = st8(d.r * 255.0f + 0.5f);
    *uq = st8(nu * 255.0f + 128.0f + 0.5f);
    *vq = st8(nv * 255.0f + 128.0f + 0.5f);
  }
}

__global__ __launch_bounds__(kBlock) void transform_yuv444_kernel(const YuvXformParams p) {
  const uint32_t w = p.img.w, h = p.img.h;
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    uint8_t* yq = (uint8_t*)p.img.p[0] + (size_t)y * p.img.stride[0] + x;
    uint8_t* uq = (uint8_t*)p.img.p[1] + (size_t)y * p.img.stride[1] + x;
    uint8_t* vq = (uint8_t*)p.img.p[2] + (size_t)y * p.img.stride[2] + x;
    const Color3 o = mat3_apply({(float)*yq * (1 / 255.0f), (float)((int)*uq - 128) * (1 / 255.0f),
                                 (float)((int)*vq - 128) * (1 / 255.0f)}, p.c);
    *yq = st8(o.r * 255.0f + 0.5f);
    *uq = st8(o.g * 255.0f + 128.0f + 0.5f);
    *vq = st8(o.b * 255.0f + 128.0f + 0.5f);
  }
}

__device__ __forceinline__ float clipf(float v, float hi) { return (v < 0.0f) ? 0.0f : ((v > hi) ? hi : v); }

// chroma-subsampled variants: one thread per 2x2 quad; 4:4:4 variants: one thread per pixel
template <bool TEN_BIT>
__global__ __launch_bounds__(kBlock) void rgb_to_ycbcr420_kernel(const RgbToYcbcrParams p) {
  __shared__ UnormTables ut;  // sample / 255.0f, sample / 1023.0f (the reference divides)
  fill_unorm_tables(ut, threadIdx.x, kBlock);
  __syncthreads();
  const uint32_t qw = p.src.w / 2, qh = p.src.h / 2;
  const uint32_t tiles_x = (qw + kBlock - 1) / kBlock, tiles = tiles_x * qh;
  const float scale = TEN_BIT ? 1023.0f : 255.0f;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t qy = t / tiles_x, qx = (t - qy * tiles_x) * kBlock + threadIdx.x;
    if (qx >= qw) continue;
    Color3 q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      q[k] = rgb_to_yuv(fetch_pixel(p.src, qx * 2 + (k & 1), qy * 2 + (k >> 1), &ut), p.k);
      q[k].r = clipf(q[k].r * scale + 0.5f, scale);
    }
    float u = (q[0].g + q[1].g + q[2].g + q[3].g) / 4;
    float v = (q[0].b + q[1].b + q[2].b + q[3].b) / 4;
    if constexpr (TEN_BIT) {  // gainmapmath.cpp:1345-1364 (P010: value << 6, interleaved UV)
      uint16_t* yd = (uint16_t*)p.dst.p[0];
      uint16_t* cd = (uint16_t*)p.dst.p[1];
      const size_t sy = p.dst.stride[0];
      yd[(size_t)(qy * 2) * sy + qx * 2] = (uint16_t)((uint32_t)q[0].r << 6);
      yd[(size_t)(qy * 2) * sy + qx * 2 + 1] = (uint16_t)((uint32_t)q[1].r << 6);
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2] = (uint16_t)((uint32_t)q[2].r << 6);
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2 + 1] = (uint16_t)((uint32_t)q[3].r << 6);
      u = clipf((u * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      v = clipf((v * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      cd[(size_t)qy * p.dst.stride[1] + qx * 2] = (uint16_t)((uint32_t)u << 6);
      cd[(size_t)qy * p.dst.stride[1] + qx * 2 + 1] = (uint16_t)((uint32_t)v << 6);
    } else {  // gainmapmath.cpp:1426-1444
      uint8_t* yd = (uint8_t*)p.dst.p[0];
      const size_t sy = p.dst.stride[0];
      yd[(size_t)(qy * 2) * sy + qx * 2] = (uint8_t)q[0].r;
      yd[(size_t)(qy * 2) * sy + qx * 2 + 1] = (uint8_t)q[1].r;
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2] = (uint8_t)q[2].r;
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2 + 1] = (uint8_t)q[3].r;
      ((uint8_t*)p.dst.p[1])[(size_t)qy * p.dst.stride[1] + qx] = (uint8_t)clipf(u * 255.0f + 0.5f + 128.0f, 255.0f);
      ((uint8_t*)p.dst.p[2])[(size_t)qy * p.dst.stride[2] + qx] = (uint8_t)clipf(v * 255.0f + 0.5f + 128.0f, 255.0f);
    }
  }
}

template <bool TEN_BIT>
__global__ __launch_bounds__(kBlock) void rgb_to_ycbcr444_kernel(const RgbToYcbcrParams p) {
  __shared__ UnormTables ut;
  fill_unorm_tables(ut, threadIdx.x, kBlock);
  __syncthreads();
  const uint32_t w = p.src.w, h = p.src.h;
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    const Color3 q = rgb_to_yuv(fetch_pixel(p.src, x, y, &ut), p.k);
    if constexpr (TEN_BIT) {  // gainmapmath.cpp:1386-1402
      ((uint16_t*)p.dst.p[0])[(size_t)y * p.dst.stride[0] + x] = (uint16_t)clipf((q.r * 1023.0f) + 0.5f, 1023.0f);
      ((uint16_t*)p.dst.p[1])[(size_t)y * p.dst.stride[1] + x] = (uint16_t)clipf((q.g * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      ((uint16_t*)p.dst.p[2])[(size_t)y * p.dst.stride[2] + x] = (uint16_t)clipf((q.b * 1023.0f) + 512.0f + 0.5f, 1023.0f);
    } else {  // gainmapmath.cpp:1458-1472
      ((uint8_t*)p.dst.p[0])[(size_t)y * p.dst.stride[0] + x] = (uint8_t)clipf(q.r * 255.0f + 0.5f, 255.0f);
      ((uint8_t*)p.dst.p[1])[(size_t)y * p.dst.stride[1] + x] = (uint8_t)clipf(q.g * 255.0f + 0.5f + 128.0f, 255.0f);
      ((uint8_t*)p.dst.p[2])[(size_t)y * p.dst.stride[2] + x] = (uint8_t)clipf(q.b * 255.0f + 0.5f + 128.0f, 255.0f);
    }
  }
}

// copy_raw_image's two repacking cases (gainmapmath.cpp:1565-1597): RGB888 -> RGBA8888 (alpha 0xff) and
// RGBA8888 -> Y400 (the R byte).  MODE 0 / 1.  One lane per pixel, tiles of 256 pixels of one row.
template <int MODE>
__global__ __launch_bounds__(kBlock) void repack_kernel(const uint8_t* __restrict__ src, size_t src_pitch,
                                                        uint8_t* __restrict__ dst, size_t dst_pitch, uint32_t w, uint32_t h) {
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    const uint8_t* s = src + (size_t)y * src_pitch;
    uint8_t* d = dst + (size_t)y * dst_pitch;
    if constexpr (MODE == 0) {
      ((uint32_t*)d)[x] = s[3 * x] | ((uint32_t)s[3 * x + 1] << 8) | ((uint32_t)s[3 * x + 2] << 16) | (0xffu << 24);
    } else {
      d[x] = (uint8_t)(((const uint32_t*)s)[x] & 0xff);
    }
  }
}

inline int grid_for(size_t total) {
  size_t g = (total + kBlock - 1) / kBlock;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

hipError_t launch_transform_yuv(const YuvXformParams& p, hipStream_t s) {
  if (p.img.fmt == UHDR_IMG_FMT_12bppYCbCr420) {
    hipLaunchKernelGGL(transform_yuv420_kernel, dim3(grid_for((size_t)(p.img.w / 2) * (p.img.h / 2))),
                       dim3(kBlock), 0, s, p);
  } else {
    hipLaunchKernelGGL(transform_yuv444_kernel, dim3(grid_for((size_t)p.img.w * p.img.h)), dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

hipError_t launch_repack(int mode, const void* src, size_t src_pitch, void* dst, size_t dst_pitch, uint32_t w, uint32_t h,
                         hipStream_t s) {
  const int g = grid_for((size_t)w * h);
  if (mode == 0) hipLaunchKernelGGL((repack_kernel<0>), dim3(g), dim3(kBlock), 0, s, (const uint8_t*)src, src_pitch, (uint8_t*)dst, dst_pitch, w, h);
  else hipLaunchKernelGGL((repack_kernel<1>), dim3(g), dim3(kBlock), 0, s, (const uint8_t*)src, src_pitch, (uint8_t*)dst, dst_pitch, w, h);
  return hipGetLastError();
}

hipError_t launch_rgb_to_ycbcr(const RgbToYcbcrParams& p, hipStream_t s) {
  const bool ten = p.src.fmt == UHDR_IMG_FMT_32bppRGBA1010102;
  const bool sub = p.dst.fmt == UHDR_IMG_FMT_24bppYCbCrP010 || p.dst.fmt == UHDR_IMG_FMT_12bppYCbCr420;
  if (sub) {
    const int g = grid_for((size_t)(p.src.w / 2) * (p.src.h / 2));
    if (ten) hipLaunchKernelGGL((rgb_to_ycbcr420_kernel<true>), dim3(g), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((rgb_to_ycbcr420_kernel<false>), dim3(g), dim3(kBlock), 0, s, p);
  } else {
    const int g = grid_for((size_t)p.src.w * p.src.h);
    if (ten) hipLaunchKernelGGL((rgb_to_ycbcr444_kernel<true>), dim3(g), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((rgb_to_ycbcr444_kernel<false>), dim3(g), dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

}  // namespace uhdr
