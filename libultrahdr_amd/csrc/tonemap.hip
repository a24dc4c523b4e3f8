// toneMap on gfx950: HDR rendition -> SDR rendition (extended Reinhard on max-RGB).
// Reference: /root/reference/lib/src/jpegr.cpp:1945-1983 (ReinhardMap, globalTonemap, ScaleTo8Bit)
// and the loop at jpegr.cpp:2147-2203.
//
// P010 -> YCbCr 4:2:0: one thread per 2x2 quad (chroma is the mean of the four converted pixels,
// accumulated in (row, col) order like the reference).  Other formats: one thread per pixel.
#include "pixel_io.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint8_t scale_to_8bit(float v) {  // jpegr.cpp:1979-1983
  int i = (int)roundf(v * 255.0f);
  return (uint8_t)min(max(i, 0), 255);
}
__device__ __forceinline__ uint32_t put8(float v) {  // put*Pixel: *255, +0.5, clip, truncate
  v *= 255.0f;
  v += 0.5f;
  v = (v < 0.0f) ? 0.0f : ((v > 255.0f) ? 255.0f : v);
  return (uint32_t)v;
}
// srgbOetf (gainmapmath.cpp:139-148): std::pow(float,float) => powf
__device__ __forceinline__ float srgb_oetf(float e) {
  if (e <= 0.0031308f) return 12.92f * e;
  return (1.0f + 0.055f) * powf(e, 1.0f / 2.4f) - 0.055f;
}

// one HDR pixel -> gamma-encoded Display-P3 SDR rgb
__device__ __forceinline__ Color3 tone_map_pixel(const ToneMapParams& p, uint32_t x, uint32_t y) {
  Color3 g = fetch_pixel(p.hdr, x, y);
  if (!p.hdr_is_rgb) g = yuv_to_rgb(g.r, g.g, g.b, p.hdr_yuv);
  Color3 l = g;
  if (p.hdr_inv_lut) {
    if (p.hdr_inv_n == kInvOetfN) {
      l.r = p.hdr_inv_lut[lut_index_f64<kInvOetfN>(g.r)];
      l.g = p.hdr_inv_lut[lut_index_f64<kInvOetfN>(g.g)];
      l.b = p.hdr_inv_lut[lut_index_f64<kInvOetfN>(g.b)];
    } else {
      l.r = p.hdr_inv_lut[lut_index_f32<kSrgbN>(g.r)];
      l.g = p.hdr_inv_lut[lut_index_f32<kSrgbN>(g.g)];
      l.b = p.hdr_inv_lut[lut_index_f32<kSrgbN>(g.b)];
    }
  }
  if (p.hdr_is_hlg) {
    l.r = powf(l.r, 1.2f); l.g = powf(l.g, 1.2f); l.b = powf(l.b, 1.2f);
  }
  // globalTonemap (jpegr.cpp:1951-1977)
  float c0 = l.r, c1 = l.g, c2 = l.b;
  const float hr = p.headroom;
  if (p.is_normalized) { c0 *= hr; c1 *= hr; c2 *= hr; }
  float mx = c0;
  if (c1 > mx) mx = c1;
  if (c2 > mx) mx = c2;
  float ms = 1.0f + mx / (hr * hr);  // ReinhardMap
  ms /= 1.0f + mx;
  ms = ms * mx;
  Color3 o;
  o.r = c0 > 0.0f ? c0 * ms / mx : 0.0f;
  o.g = c1 > 0.0f ? c1 * ms / mx : 0.0f;
  o.b = c2 > 0.0f ? c2 * ms / mx : 0.0f;
  if (p.gamut_on) o = mat3_apply(o, p.gamut);
  o.r = clamp01(o.r); o.g = clamp01(o.g); o.b = clamp01(o.b);
  Color3 og = {srgb_oetf(o.r), srgb_oetf(o.g), srgb_oetf(o.b)};
  return og;
}

__global__ __launch_bounds__(kBlock) void tonemap_p010_kernel(const ToneMapParams p) {
  const uint32_t qw = p.hdr.w / 2, qh = p.hdr.h / 2;
  const size_t total = (size_t)qw * qh;
  uint8_t* yp = (uint8_t*)p.sdr.p[0];
  uint8_t* up = (uint8_t*)p.sdr.p[1];
  uint8_t* vp = (uint8_t*)p.sdr.p[2];
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const uint32_t qy = (uint32_t)(i / qw), qx = (uint32_t)(i - (size_t)qy * qw);
    float su = 0.0f, sv = 0.0f;
    uint32_t yb[2][2];
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 2; c++) {
        Color3 og = tone_map_pixel(p, qx * 2 + c, qy * 2 + r);
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
    yp[(size_t)(qy * 2) * sy + qx * 2] = (uint8_t)yb[0][0];
    yp[(size_t)(qy * 2) * sy + qx * 2 + 1] = (uint8_t)yb[0][1];
    yp[(size_t)(qy * 2 + 1) * sy + qx * 2] = (uint8_t)yb[1][0];
    yp[(size_t)(qy * 2 + 1) * sy + qx * 2 + 1] = (uint8_t)yb[1][1];
    up[(size_t)qy * p.sdr.stride[1] + qx] = scale_to_8bit(su);
    vp[(size_t)qy * p.sdr.stride[2] + qx] = scale_to_8bit(sv);
  }
}

__global__ __launch_bounds__(kBlock) void tonemap_pixel_kernel(const ToneMapParams p) {
  const uint32_t w = p.hdr.w, h = p.hdr.h;
  const size_t total = (size_t)w * h;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const uint32_t y = (uint32_t)(i / w), x = (uint32_t)(i - (size_t)y * w);
    const Color3 og = tone_map_pixel(p, x, y);
    if (p.sdr.fmt == UHDR_IMG_FMT_32bppRGBA8888) {  // putRgba8888Pixel (gainmapmath.cpp:538-552)
      ((uint32_t*)p.sdr.p[0])[x + (size_t)y * p.sdr.stride[0]] =
          put8(og.r) | (put8(og.g) << 8) | (put8(og.b) << 16) | (255u << 24);
    } else {  // 4:4:4: p3RgbToYuv, +0.5 chroma offset, putYuv444Pixel (gainmapmath.cpp:579-596)
      Color3 yuv = rgb_to_yuv(og, p.p3);
      yuv.g += 0.5f;
      yuv.b += 0.5f;
      ((uint8_t*)p.sdr.p[0])[x + (size_t)y * p.sdr.stride[0]] = (uint8_t)put8(yuv.r);
      ((uint8_t*)p.sdr.p[1])[x + (size_t)y * p.sdr.stride[1]] = (uint8_t)put8(yuv.g);
      ((uint8_t*)p.sdr.p[2])[x + (size_t)y * p.sdr.stride[2]] = (uint8_t)put8(yuv.b);
    }
  }
}

}  // namespace

hipError_t launch_tone_map(const ToneMapParams& p, hipStream_t s) {
  if (p.hdr.fmt == UHDR_IMG_FMT_24bppYCbCrP010) {
    size_t total = (size_t)(p.hdr.w / 2) * (p.hdr.h / 2);
    int grid = (int)min((total + kBlock - 1) / kBlock, (size_t)4096);
    hipLaunchKernelGGL(tonemap_p010_kernel, dim3(max(grid, 1)), dim3(kBlock), 0, s, p);
  } else {
    size_t total = (size_t)p.hdr.w * p.hdr.h;
    int grid = (int)min((total + kBlock - 1) / kBlock, (size_t)4096);
    hipLaunchKernelGGL(tonemap_pixel_kernel, dim3(max(grid, 1)), dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

}  // namespace uhdr
