// toneMap on gfx950: HDR rendition -> SDR rendition (extended Reinhard on max-RGB).
// Reference: /root/reference/lib/src/jpegr.cpp:1945-1983 (ReinhardMap, globalTonemap, ScaleTo8Bit)
// and the loop at jpegr.cpp:2147-2203.
//
// P010 -> YCbCr 4:2:0: one thread per 2x2 quad (chroma is the mean of the four converted pixels,
// accumulated in (row, col) order like the reference).  Other formats: one thread per pixel.
//
// The kernels are instantiated per HDR format; the HDR linearisation table (inverse OETF with, for
// HLG, hlgOotfApprox folded in by the host -- see generate_gainmap.hip) and the float64 tables
// behind srgbOetf's pow (exact_math.h: the direct table, 18 KB) are staged in LDS.  A workgroup walks tiles of 256
// consecutive quads / pixels of one row.  Round 4: Markstein divisions, v_cvt_rpi table indices and the direct pow
// table (encode_core.h) -- P010 4K 51 -> see DESIGN.md 5.2.
#include "encode_core.h"
#include "lds_copy.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;

// jpegr.cpp:1979-1983: clamp((int)std::round(v * 255), 0, 255).  std::round is half away from zero, floor(t + 0.5)
// (v_cvt_rpi_i32_f32, exact) is half up: they differ for negative t only, where both land at or below 0 and clamp to 0
// (t in (-0.5, 0): both 0 / -0; t <= -0.5: both negative).
__device__ __forceinline__ uint32_t scale_to_8bit(float v) {
  int i = rpi(v * 255.0f), o;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o) : "v"(i), "v"(255));
  return (uint32_t)o;
}
struct ToneLds {
  float hdr[kInvOetfN];  // RGBA1010102 input with p.lin10: the first 1024 entries hold code -> linear value
  double powt[kPowDirDoubles];  // exact_math.h: pow_direct_f32
  UnormTables unorm;
};
// one HDR sample (as fetch_pixel returns it) -> gamma-encoded Display-P3 SDR rgb
__device__ __forceinline__ Color3 tone_map_sample(const ToneMapParams& p, const ToneLds& L, Color3 g) {
  if (!p.hdr_is_rgb) g = yuv_to_rgb(g.r, g.g, g.b, p.hdr_yuv);
  const Color3 l = linearise_hdr(g, L.hdr, p.hdr_inv_lut != nullptr, p.hdr_inv_n == kInvOetfN);
  return tone_curve(l, p, L.powt);
}
template <int HDRF>
__device__ __forceinline__ Color3 tone_map_pixel(const ToneMapParams& p, const ToneLds& L, uint32_t x, uint32_t y) {
  return tone_map_sample(p, L, fetch_pixel<HDRF>(p.hdr, x, y, &L.unorm));
}

// A WAVE walks tiles of 64 consecutive quads of one quad row (4K: 30 tiles per row, none ragged), 512-thread workgroups
// share one 39 KB table set (four per CU = 32 waves).  The gamut conversion and the presence of a linearisation table are
// template parameters and the table size a float factor, so the four pixels of a quad are ONE basic block: the
// compiler issues their twelve table reads together instead of waiting for each.
constexpr int kQuadBlock = 512;
template <int GAMUT, bool LUT>
__global__ __launch_bounds__(kQuadBlock) void tonemap_p010_kernel(const ToneMapParams p) {
  __shared__ ToneLds L;
  if (LUT) copy_to_lds(L.hdr, p.hdr_inv_lut, (uint32_t)p.hdr_inv_n, threadIdx.x, kQuadBlock);
  stage_pow_tab(L.powt, p.math_tab, threadIdx.x, kQuadBlock);
  fill_unorm_tables(L.unorm, threadIdx.x, kQuadBlock);
  __syncthreads();
  const uint32_t qw = p.hdr.w / 2, qh = p.hdr.h / 2;
  const uint32_t tiles_x = (qw + 63) / 64, tiles = tiles_x * qh;
  const float inv_tx = 1.0f / (float)tiles_x;
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (kQuadBlock / 64) + (threadIdx.x >> 6));
  const uint32_t nwaves = gridDim.x * (kQuadBlock / 64);
  uint8_t* yp = (uint8_t*)p.sdr.p[0];
  uint8_t* up = (uint8_t*)p.sdr.p[1];
  uint8_t* vp = (uint8_t*)p.sdr.p[2];
  const bool vec_in = quad_layout_ok(p.hdr);
  const float lut_scale = (float)(p.hdr_inv_n - 1);  // lut_index: x * (N - 1), round half up (encode_core.h: rpi)
  for (uint32_t t = wave; t < tiles; t += nwaves) {
    uint32_t qy = (uint32_t)((float)t * inv_tx);  // t / tiles_x for t < 2^24: the float estimate is off by at most one
    if (qy * tiles_x > t) qy--;
    if ((qy + 1) * tiles_x <= t) qy++;
    const uint32_t qx = (t - qy * tiles_x) * 64 + lane;
    if (qx >= qw) continue;
    QuadYuv hq;
    if (vec_in) {  // coalesced: one dword of luma per row, the (U, V) pair once
      hq = fetch_quad_p010(p.hdr, qx, qy, &L.unorm);
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) hq.px[k] = fetch_pixel<UHDR_IMG_FMT_24bppYCbCrP010>(p.hdr, qx * 2 + (k & 1), qy * 2 + (k >> 1), &L.unorm);
    }
    Color3 l[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const Color3 g = yuv_to_rgb(hq.px[k].r, hq.px[k].g, hq.px[k].b, p.hdr_yuv);  // in [0, 1]
      l[k] = g;
      if (LUT) l[k] = Color3{L.hdr[rpi(g.r * lut_scale)], L.hdr[rpi(g.g * lut_scale)], L.hdr[rpi(g.b * lut_scale)]};
    }
    float su = 0.0f, sv = 0.0f;
    uint32_t yb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {  // (row, column) order: the chroma sums accumulate like the reference's
      const Color3 og = tone_curve<GAMUT>(l[k], p, L.powt);
      Color3 yuv = rgb_to_yuv(og, p.p3);
      yuv.g += 0.5f;
      yuv.b += 0.5f;
      yb[k] = scale_to_8bit(yuv.r);
      su += yuv.g;
      sv += yuv.b;
    }
    su *= 0.25f;  // x / 4.0f == x * 0.25f exactly
    sv *= 0.25f;
    const size_t sy = p.sdr.stride[0];
    uint8_t* y0 = yp + (size_t)(qy * 2) * sy + qx * 2;
    if (((sy | (uintptr_t)yp) & 1) == 0) {  // two luma bytes per row as one 16-bit store
      *(uint16_t*)y0 = (uint16_t)(yb[0] | (yb[1] << 8));
      *(uint16_t*)(y0 + sy) = (uint16_t)(yb[2] | (yb[3] << 8));
    } else {
      y0[0] = (uint8_t)yb[0]; y0[1] = (uint8_t)yb[1];
      y0[sy] = (uint8_t)yb[2]; y0[sy + 1] = (uint8_t)yb[3];
    }
    up[(size_t)qy * p.sdr.stride[1] + qx] = (uint8_t)scale_to_8bit(su);
    vp[(size_t)qy * p.sdr.stride[2] + qx] = (uint8_t)scale_to_8bit(sv);
  }
}

template <int HDRF>
__global__ __launch_bounds__(kBlock) void tonemap_pixel_kernel(const ToneMapParams p) {
  __shared__ ToneLds L;
  // RGBA8888 output: clamped linear value -> sRGB byte through the step table, which then takes the pow table's place
  // (a call uses one or the other; kStepTabMax entries of 8 bytes fit the 18 KB of the pow table)
  static_assert(kStepTabMax * sizeof(uint2) <= sizeof(L.powt), "the sRGB byte table shares the pow table's storage");
  uint2* const s_srgb8 = (uint2*)L.powt;
  const bool code_lin = HDRF == UHDR_IMG_FMT_32bppRGBA1010102 && p.lin10 != nullptr;
  const bool bytes_tab = p.sdr.fmt == UHDR_IMG_FMT_32bppRGBA8888 && p.srgb8.tab != nullptr;
  if (code_lin) {
    copy_to_lds(L.hdr, p.lin10, 1024u, threadIdx.x, kBlock);
  } else if (p.hdr_inv_lut) {
    copy_to_lds(L.hdr, p.hdr_inv_lut, (uint32_t)p.hdr_inv_n, threadIdx.x, kBlock);
  }
  if (bytes_tab) stage_step_tab(s_srgb8, p.srgb8, threadIdx.x, kBlock);
  else stage_pow_tab(L.powt, p.math_tab, threadIdx.x, kBlock);
  fill_unorm_tables(L.unorm, threadIdx.x, kBlock);
  __syncthreads();
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
      tone_curve_bytes(l, p, L.powt, s_srgb8, r8, g8, b8);
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

// per_cu: resident workgroups per CU (39 KB of LDS tables each): four, of 512 threads (P010 kernel) or of 256 (pixel kernel)
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
    const uint32_t wave_tiles = ((qw + 63) / 64) * qh;
    const int grid = tone_grid((wave_tiles + kQuadBlock / 64 - 1) / (kQuadBlock / 64), 4);
    const bool lut = p.hdr_inv_lut != nullptr;
    if (p.gamut_on) {
      if (lut) hipLaunchKernelGGL((tonemap_p010_kernel<1, true>), dim3(grid), dim3(kQuadBlock), 0, s, p);
      else hipLaunchKernelGGL((tonemap_p010_kernel<1, false>), dim3(grid), dim3(kQuadBlock), 0, s, p);
    } else {
      if (lut) hipLaunchKernelGGL((tonemap_p010_kernel<0, true>), dim3(grid), dim3(kQuadBlock), 0, s, p);
      else hipLaunchKernelGGL((tonemap_p010_kernel<0, false>), dim3(grid), dim3(kQuadBlock), 0, s, p);
    }
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
