// Device-side pixel fetch for every input format the hot path reads
// (get*Pixel, /root/reference/lib/src/gainmapmath.cpp:354-492) and the s x s box sampler
// (samplePixels, gainmapmath.cpp:494-504).
#pragma once
#include "uhdr_types.h"

namespace uhdr {

// Returns the pixel as the reference's Color: (y,u,v) for YCbCr formats, (r,g,b) for RGB ones.
// FMT >= 0 fixes the format at compile time (the encode kernels are instantiated per format so the
// switch folds away); FMT < 0 reads it from the view.
//
// The reference normalises RGBA8888 / RGB888 samples with a float DIVISION by 255.0f and full-range
// 10-bit samples by 1023.0f (gainmapmath.cpp:461-481, 438-441); x * (1/255.0f) is not the same
// float for every x.  UnormTables holds those quotients for every possible sample (filled with the
// device's own correctly rounded division, see fill_unorm_tables): one LDS read instead of an
// 11-instruction IEEE division per channel.  Pass nullptr to divide in place.
struct UnormTables {
  float u8[256];    // i / 255.0f
  float u10[1024];  // i / 1023.0f
};
__device__ __forceinline__ void fill_unorm_tables(UnormTables& t, uint32_t tid, uint32_t nthreads) {
  for (uint32_t i = tid; i < 256; i += nthreads) t.u8[i] = (float)i / 255.0f;
  for (uint32_t i = tid; i < 1024; i += nthreads) t.u10[i] = (float)i / 1023.0f;
}

template <int FMT = -1>
__device__ __forceinline__ Color3 fetch_pixel(const ImageView& im, uint32_t x, uint32_t y, const UnormTables* ut = nullptr) {
  Color3 c = {0.f, 0.f, 0.f};
  const int fmt_ = FMT >= 0 ? FMT : im.fmt;
  switch (fmt_) {
    case UHDR_IMG_FMT_24bppYCbCr444:
    case UHDR_IMG_FMT_16bppYCbCr422:
    case UHDR_IMG_FMT_12bppYCbCr420: {
      const uint32_t hf = fmt_ == UHDR_IMG_FMT_24bppYCbCr444 ? 1 : 2;
      const uint32_t vf = fmt_ == UHDR_IMG_FMT_12bppYCbCr420 ? 2 : 1;
      const int yy = ((const uint8_t*)im.p[0])[x + (size_t)y * im.stride[0]];
      const int uu = ((const uint8_t*)im.p[1])[x / hf + (size_t)(y / vf) * im.stride[1]];
      const int vv = ((const uint8_t*)im.p[2])[x / hf + (size_t)(y / vf) * im.stride[2]];
      c.r = (float)yy * (1 / 255.0f);
      c.g = (float)(uu - 128) * (1 / 255.0f);
      c.b = (float)(vv - 128) * (1 / 255.0f);
      break;
    }
    case UHDR_IMG_FMT_8bppYCbCr400:
      c.r = (float)((const uint8_t*)im.p[0])[x + (size_t)y * im.stride[0]] * (1 / 255.0f);
      break;
    case UHDR_IMG_FMT_24bppYCbCrP010:
    case UHDR_IMG_FMT_30bppYCbCr444: {
      int yy, uu, vv;
      if (fmt_ == UHDR_IMG_FMT_24bppYCbCrP010) {
        const uint16_t* yp = (const uint16_t*)im.p[0];
        const uint16_t* cp = (const uint16_t*)im.p[1];
        const size_t ui = (size_t)(y >> 1) * im.stride[1] + (x & ~1u);
        yy = yp[(size_t)y * im.stride[0] + x] >> 6;
        uu = cp[ui] >> 6;
        vv = cp[ui + 1] >> 6;
      } else {
        yy = ((const uint16_t*)im.p[0])[(size_t)y * im.stride[0] + x];
        uu = ((const uint16_t*)im.p[1])[(size_t)y * im.stride[1] + x];
        vv = ((const uint16_t*)im.p[2])[(size_t)y * im.stride[2] + x];
      }
      if (im.range == UHDR_CR_FULL_RANGE) {
        if (ut) {  // 10-bit samples: P010 carries them in the top bits (>> 6 above), 4:4:4 may exceed 1023 only if corrupt
          c.r = ut->u10[yy & 1023];
          c.g = ut->u10[uu & 1023] - 0.5f;
          c.b = ut->u10[vv & 1023] - 0.5f;
        } else {
          c.r = (float)yy / 1023.0f;
          c.g = (float)uu / 1023.0f - 0.5f;
          c.b = (float)vv / 1023.0f - 0.5f;
        }
      } else {
        c.r = (float)(yy - 64) * (1 / 876.0f);
        c.g = (float)(uu - 64) * (1 / 896.0f) - 0.5f;
        c.b = (float)(vv - 64) * (1 / 896.0f) - 0.5f;
      }
      break;
    }
    case UHDR_IMG_FMT_24bppRGB888: {
      const uint8_t* q = (const uint8_t*)im.p[0] + (size_t)x * 3 + (size_t)y * im.stride[0] * 3;
      if (ut) {
        c.r = ut->u8[q[0]]; c.g = ut->u8[q[1]]; c.b = ut->u8[q[2]];
      } else {
        c.r = (float)q[0] / 255.0f;
        c.g = (float)q[1] / 255.0f;
        c.b = (float)q[2] / 255.0f;
      }
      break;
    }
    case UHDR_IMG_FMT_32bppRGBA8888: {
      const uint32_t v = ((const uint32_t*)im.p[0])[x + (size_t)y * im.stride[0]];
      if (ut) {
        c.r = ut->u8[v & 0xff]; c.g = ut->u8[(v >> 8) & 0xff]; c.b = ut->u8[(v >> 16) & 0xff];
      } else {
        c.r = (float)(v & 0xff) / 255.0f;
        c.g = (float)((v >> 8) & 0xff) / 255.0f;
        c.b = (float)((v >> 16) & 0xff) / 255.0f;
      }
      break;
    }
    case UHDR_IMG_FMT_32bppRGBA1010102: {
      const uint32_t v = ((const uint32_t*)im.p[0])[x + (size_t)y * im.stride[0]];
      if (ut) {
        c.r = ut->u10[v & 0x3ff]; c.g = ut->u10[(v >> 10) & 0x3ff]; c.b = ut->u10[(v >> 20) & 0x3ff];
      } else {
        c.r = (float)(v & 0x3ff) / 1023.0f;
        c.g = (float)((v >> 10) & 0x3ff) / 1023.0f;
        c.b = (float)((v >> 20) & 0x3ff) / 1023.0f;
      }
      break;
    }
    case UHDR_IMG_FMT_64bppRGBAHalfFloat: {
      const uint2 v = ((const uint2*)im.p[0])[x + (size_t)y * im.stride[0]];
      c.r = sanitize_linear(half_to_float_ref(v.x & 0xffff));
      c.g = sanitize_linear(half_to_float_ref(v.x >> 16));
      c.b = sanitize_linear(half_to_float_ref(v.y & 0xffff));
      break;
    }
    default: break;
  }
  return c;
}

// samplePixels: sum over the s x s box (dy outer, dx inner), then one divide per channel.
template <int FMT = -1>
__device__ __forceinline__ Color3 sample_box(const ImageView& im, uint32_t s, uint32_t x, uint32_t y, const UnormTables* ut = nullptr) {
  if (s == 1) {  // e = 0 + p; e / 1.0f  == p bit for bit (0.0f + p == p, p / 1 == p)
    return fetch_pixel<FMT>(im, x, y, ut);
  }
  Color3 e = {0.0f, 0.0f, 0.0f};
  for (uint32_t dy = 0; dy < s; ++dy)
    for (uint32_t dx = 0; dx < s; ++dx) {
      const Color3 q = fetch_pixel<FMT>(im, x * s + dx, y * s + dy, ut);
      e.r += q.r;
      e.g += q.g;
      e.b += q.b;
    }
  const float d = (float)(s * s);
  e.r /= d;
  e.g /= d;
  e.b /= d;
  return e;
}

__device__ __forceinline__ bool is_rgb_fmt(int fmt) {  // isPixelFormatRgb, gainmapmath.cpp:1274-1277
  return fmt == UHDR_IMG_FMT_64bppRGBAHalfFloat || fmt == UHDR_IMG_FMT_32bppRGBA8888 ||
         fmt == UHDR_IMG_FMT_32bppRGBA1010102;
}

}  // namespace uhdr

namespace uhdr {

// ---- 2x2 quad fetch for the two chroma-subsampled layouts -----------------------------------------------
// One lane reads the four luma samples of a quad with one load per row (P010: a dword = two 16-bit
// samples; 4:2:0: a 16-bit load = two bytes) and the quad's chroma once (P010: one dword = the U, V
// pair; 4:2:0: one byte from each plane), instead of three scalar loads per pixel; a wave then covers
// 128 consecutive pixels of two rows with fully coalesced loads.  The normalisation arithmetic is
// fetch_pixel's, so px[k] is bit-identical to fetch_pixel(x0 + (k & 1), y0 + (k >> 1)).
// Layout contract (checked by the launchers): even x0 / y0, row pitches and plane bases that keep the
// vector loads aligned.
struct QuadYuv {
  Color3 px[4];  // (row 0, col 0), (row 0, col 1), (row 1, col 0), (row 1, col 1): y, u, v as fetch_pixel returns them
};

__device__ __forceinline__ QuadYuv fetch_quad_p010(const ImageView& im, uint32_t qx, uint32_t qy, const UnormTables* ut) {
  const uint16_t* yp = (const uint16_t*)im.p[0];
  const uint16_t* cp = (const uint16_t*)im.p[1];
  const uint32_t ya = *(const uint32_t*)(yp + (size_t)(2 * qy) * im.stride[0] + 2 * qx);
  const uint32_t yb = *(const uint32_t*)(yp + (size_t)(2 * qy + 1) * im.stride[0] + 2 * qx);
  const uint32_t uv = *(const uint32_t*)(cp + (size_t)qy * im.stride[1] + 2 * qx);
  const int ys[4] = {(int)((ya & 0xffff) >> 6), (int)(ya >> 22), (int)((yb & 0xffff) >> 6), (int)(yb >> 22)};
  const int uu = (int)((uv & 0xffff) >> 6), vv = (int)(uv >> 22);
  QuadYuv q;
  float cu, cv;
  if (im.range == UHDR_CR_FULL_RANGE) {
    if (ut) { cu = ut->u10[uu] - 0.5f; cv = ut->u10[vv] - 0.5f; }
    else { cu = (float)uu / 1023.0f - 0.5f; cv = (float)vv / 1023.0f - 0.5f; }
  } else {
    cu = (float)(uu - 64) * (1 / 896.0f) - 0.5f;
    cv = (float)(vv - 64) * (1 / 896.0f) - 0.5f;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    float yf;
    if (im.range == UHDR_CR_FULL_RANGE) yf = ut ? ut->u10[ys[k]] : (float)ys[k] / 1023.0f;
    else yf = (float)(ys[k] - 64) * (1 / 876.0f);
    q.px[k] = {yf, cu, cv};
  }
  return q;
}

__device__ __forceinline__ QuadYuv fetch_quad_420(const ImageView& im, uint32_t qx, uint32_t qy) {
  const uint8_t* yp = (const uint8_t*)im.p[0];
  const uint32_t ya = *(const uint16_t*)(yp + (size_t)(2 * qy) * im.stride[0] + 2 * qx);
  const uint32_t yb = *(const uint16_t*)(yp + (size_t)(2 * qy + 1) * im.stride[0] + 2 * qx);
  const int uu = ((const uint8_t*)im.p[1])[(size_t)qy * im.stride[1] + qx];
  const int vv = ((const uint8_t*)im.p[2])[(size_t)qy * im.stride[2] + qx];
  const float cu = (float)(uu - 128) * (1 / 255.0f), cv = (float)(vv - 128) * (1 / 255.0f);
  QuadYuv q;
  q.px[0] = {(float)(int)(ya & 0xff) * (1 / 255.0f), cu, cv};
  q.px[1] = {(float)(int)(ya >> 8) * (1 / 255.0f), cu, cv};
  q.px[2] = {(float)(int)(yb & 0xff) * (1 / 255.0f), cu, cv};
  q.px[3] = {(float)(int)(yb >> 8) * (1 / 255.0f), cu, cv};
  return q;
}

// do the vector loads of fetch_quad_* stay aligned for this image?
__host__ __device__ inline bool quad_layout_ok(const ImageView& im) {
  if (im.w % 2 || im.h % 2) return false;
  if (im.fmt == UHDR_IMG_FMT_24bppYCbCrP010)
    return im.stride[0] % 2 == 0 && im.stride[1] % 2 == 0 && ((uintptr_t)im.p[0] % 4 == 0) && ((uintptr_t)im.p[1] % 4 == 0);
  if (im.fmt == UHDR_IMG_FMT_12bppYCbCr420) return im.stride[0] % 2 == 0 && ((uintptr_t)im.p[0] % 2 == 0);
  return false;
}

}  // namespace uhdr
