// toneMap on gfx950: HDR rendition -> SDR rendition (extended Reinhard on max-RGB).
// Reference: /root/reference/lib/src/jpegr.cpp:1945-1983 (ReinhardMap, globalTonemap, ScaleTo8Bit)
// and the loop at jpegr.cpp:2147-2203.
//
// P010 -> YCbCr 4:2:0: one thread per 2x2 quad (chroma is the mean of the four converted pixels,
// accumulated in (row, col) order like the reference).  Other formats: one thread per pixel.
//
// The kernels are instantiated per HDR format; the HDR linearisation table (inverse OETF with, for
// HLG, hlgOotfApprox folded in by the host -- see generate_gainmap.hip) and the float64 tables
// behind srgbOetf's pow (exact_math.h) are staged in LDS.  A workgroup walks tiles of 256
// consecutive quads / pixels of one row.
#include "encode_core.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint8_t scale_to_8bit(float v) {  // jpegr.cpp:1979-1983
  int i = (int)roundf(v * 255.0f);
  return (uint8_t)min(max(i, 0), 255);
}
struct ToneLds {
  float hdr[kInvOetfN];  // RGBA1010102 input with p.lin10: the first 1024 entries hold code -> linear value
  double math[kMathTabDoubles];
  UnormTables unorm;
};
__device__ __forceinline__ void stage_tables(const ToneMapParams& p, ToneLds& L) {
  if (p.hdr_inv_lut)
    for (uint32_t i = threadIdx.x; i < (uint32_t)p.hdr_inv_n; i += kBlock) L.hdr[i] = p.hdr_inv_lut[i];
  for (uint32_t i = threadIdx.x; i < kMathTabDoubles; i += kBlock) L.math[i] = p.math_tab[i];
  fill_unorm_tables(L.unorm, threadIdx.x, kBlock);
  __syncthreads();
}

// one HDR sample (as fetch_pixel returns it) -> gamma-encoded Display-P3 SDR rgb
__device__ __forceinline__ Color3 tone_map_sample(const ToneMapParams& p, const ToneLds& L, Color3 g) {
  if (!p.hdr_is_rgb) g = yuv_to_rgb(g.r, g.g, g.b, p.hdr_yuv);
  const Color3 l = linearise_hdr(g, L.hdr, p.hdr_inv_lut != nullptr, p.hdr_inv_n == kInvOetfN);
  return tone_curve(l, p, L.math);
}
template <int HDRF>
__device__ __forceinline__ Color3 tone_map_pixel(const ToneMapParams& p, const ToneLds& L, uint32_t x, uint32_t y) {
  return tone_map_sample(p, L, fetch_pixel<HDRF>(p.hdr, x, y, &L.unorm));
}

__global__ __launch_bounds__(kBlock) void tonemap_p010_kernel(const ToneMapParams p) {
  __shared__ ToneLds L;
  stage_tables(p, L);
  const uint32_t qw = p.hdr.w / 2, qh = p.hdr.h / 2;
  const uint32_t tiles_x = (qw + kBlock - 1) / kBlock, tiles = tiles_x * qh;
  uint8_t* yp = (uint8_t*)p.sdr.p[0];
  uint8_t* up = (uint8_t*)p.sdr.p[1];
  uint8_t* vp = (uint8_t*)p.sdr.p[2];
  const bool vec_in = quad_layout_ok(p.hdr);
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t qy = t / tiles_x, qx = (t - qy * tiles_x) * kBlock + threadIdx.x;
    if (qx >= qw) continue;
    float su = 0.0f, sv = 0.0f;
    uint32_t yb[2][2];
    QuadYuv hq;
    if (vec_in) {  // coalesced: one dword of luma per row, the (U, V) pair once
      hq = fetch_quad_p010(p.hdr, qx, qy, &L.unorm);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) hq.px[k] = fetch_pixel<UHDR_IMG_FMT_24bppYCbCrP010>(p.hdr, qx * 2 + (k & 1), qy * 2 + (k >> 1), &L.unorm);
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 2; c++) {
        Color3 og = tone_map_sample(p, L, hq.px[r * 2 + c]);
        Color3 yuv = rgb_to_yuv(og, p.p3);
        yuv.g += 0.5f;
        yuv.b += 0.5f;
        yb[r][c] = scale_to_8bit(yuv.r);
        su += yuv.g;
        sv += yuv.b;
      }
    su /= 4.0f;
    sv /= 4.0f;
    const size_t sy = p.sdr.stride[0];
    uint8_t* y0 = yp + (size_t)(qy * 2) * sy + qx * 2;
    if (((sy | (uintptr_t)yp) & 1) == 0) {  // two luma bytes per row as one 16-bit store
      *(uint16_t*)y0 = (uint16_t)(yb[0][0] | (yb[0][1] << 8));
      *(uint16_t*)(y0 + sy) = (uint16_t)(yb[1][0] | (yb[1][1] << 8));
    } else {
      y0[0] = (uint8_t)yb[0][0]; y0[1] = (uint8_t)yb[0][1];
      y0[sy] = (uint8_t)yb[1][0]; y0[sy + 1] = (uint8_t)yb[1][1];
    }
    up[(size_t)qy * p.sdr.stride[1] + qx] = scale_to_8bit(su);
    vp[(size_t)qy * p.sdr.stride[2] + qx] = scale_to_8bit(sv);
  }
}

template <int HDRF>
__global__ __launch_bounds__(kBlock) void tonemap_pixel_kernel(const ToneMapParams p) {
  __shared__ ToneLds L;
  __shared__ uint2 s_srgb8[kStepTabMax];  // RGBA8888 output: clamped linear value -> sRGB byte
  const bool code_lin = HDRF == UHDR_IMG_FMT_32bppRGBA1010102 && p.lin10 != nullptr;
  if (code_lin)
    for (uint32_t i = threadIdx.x; i < 1024; i += kBlock) L.hdr[i] = p.lin10[i];
  stage_step_tab(s_srgb8, p.srgb8, threadIdx.x, kBlock);
  if (code_lin) {  // everything of stage_tables except the inverse-OETF table
    for (uint32_t i = threadIdx.x; i < kMathTabDoubles; i += kBlock) L.math[i] = p.math_tab[i];
    fill_unorm_tables(L.unorm, threadIdx.x, kBlock);
    __syncthreads();
  } else {
    stage_tables(p, L);
  }
  const uint32_t w = p.hdr.w, h = p.hdr.h;
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    if (p.sdr.fmt == UHDR_IMG_FMT_32bppRGBA8888) {  // putRgba8888Pixel (gainmapmath.cpp:538-552)
      Color3 l;
      if (code_lin) {  // unpack + inverse OETF (+ OOTF) of a 10-bit code is one table entry
        const uint32_t v = ((const uint32_t*)p.hdr.p[0])[x + (size_t)y * p.hdr.stride[0]];
        l = Color3{L.hdr[v & 0x3ffu], L.hdr[(v >> 10) & 0x3ffu], L.hdr[(v >> 20) & 0x3ffu]};
      } else {
        Color3 g = fetch_pixel<HDRF>(p.hdr, x, y, &L.unorm);
        if (!p.hdr_is_rgb) g = yuv_to_rgb(g.r, g.g, g.b, p.hdr_yuv);
        l = linearise_hdr(g, L.hdr, p.hdr_inv_lut != nullptr, p.hdr_inv_n == kInvOetfN);
      }
      uint32_t r8, g8, b8;
      tone_curve_bytes(l, p, L.math, s_srgb8, r8, g8, b8);
      ((uint32_t*)p.sdr.p[0])[x + (size_t)y * p.sdr.stride[0]] = r8 | (g8 << 8) | (b8 << 16) | (255u << 24);
    } else {  // 4:4:4: p3RgbToYuv, +0.5 chroma offset, putYuv444Pixel (gainmapmath.cpp:579-596)
      const Color3 og = tone_map_pixel<HDRF>(p, L, x, y);
      Color3 yuv = rgb_to_yuv(og, p.p3);
      yuv.g += 0.5f;
      yuv.b += 0.5f;
      ((uint8_t*)p.sdr.p[0])[x + (size_t)y * p.sdr.stride[0]] = (uint8_t)put8(yuv.r);
      ((uint8_t*)p.sdr.p[1])[x + (size_t)y * p.sdr.stride[1]] = (uint8_t)put8(yuv.g);
      ((uint8_t*)p.sdr.p[2])[x + (size_t)y * p.sdr.stride[2]] = (uint8_t)put8(yuv.b);
    }
  }
}

// per_cu: workgroups of LDS tables that fit a CU's 160 KB -- 24 KB each (P010 kernel): six; 40 KB (pixel kernel, with the
// sRGB byte table): four
int tone_grid(uint32_t tiles, int per_cu) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int resident = cus * per_cu;
  const uint32_t g = tiles < (uint32_t)resident ? tiles : (uint32_t)resident;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

hipError_t launch_tone_map(const ToneMapParams& p, hipStream_t s) {
  if (p.hdr.fmt == UHDR_IMG_FMT_24bppYCbCrP010) {
    const uint32_t qw = p.hdr.w / 2, qh = p.hdr.h / 2;
    const int grid = tone_grid(((qw + kBlock - 1) / kBlock) * qh, 6);
    hipLaunchKernelGGL(tonemap_p010_kernel, dim3(grid), dim3(kBlock), 0, s, p);
  } else {
    const int grid = tone_grid(((p.hdr.w + kBlock - 1) / kBlock) * p.hdr.h, 4);
    switch (p.hdr.fmt) {
      case UHDR_IMG_FMT_32bppRGBA1010102:
        hipLaunchKernelGGL((tonemap_pixel_kernel<UHDR_IMG_FMT_32bppRGBA1010102>), dim3(grid), dim3(kBlock), 0, s, p);
        break;
      case UHDR_IMG_FMT_64bppRGBAHalfFloat:
        hipLaunchKernelGGL((tonemap_pixel_kernel<UHDR_IMG_FMT_64bppRGBAHalfFloat>), dim3(grid), dim3(kBlock), 0, s, p);
        break;
      default:
        hipLaunchKernelGGL((tonemap_pixel_kernel<-1>), dim3(grid), dim3(kBlock), 0, s, p);
        break;
    }
  }
  return hipGetLastError();
}

}  // namespace uhdr
