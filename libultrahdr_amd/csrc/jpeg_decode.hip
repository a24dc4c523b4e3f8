// JPEG decode stage on gfx950: dequantize + 8x8 inverse DCT + range limit, and
// libjpeg's two colour conversions for 3-channel gain maps -- all bit-exact integer work.
//
// Like the forward DCT this arithmetic is NOT in the reference tree: JpegDecoderHelper
// (/root/reference/lib/src/jpegdecoderhelper.cpp:169-535) sets dct_method = JDCT_ISLOW and pulls
// raw data (base image, Y400 map) or RGB scanlines (3-channel map) out of libjpeg; the encoder hands
// an RGB map to libjpeg as JCS_RGB (jpegencoderhelper.cpp:165-167, 212-225).  What runs there is the
// public Loeffler-Ligtenberg-Moschytz integer IDCT of jidctint.c (CONST_BITS 13, PASS1_BITS 2:
// column pass from the dequantized coefficients, row pass, descale by 2^18, +128, range-limit
// table indexed modulo 1024) and jccolor.c / jdcolor.c's 16-bit fixed-point colour transforms.
//
// IDCT mapping (mirror image of fdct_quant.hip): one wavefront = 8 horizontally adjacent blocks.
// Lane (row r, block b) loads one coefficient row (8 x int16 = 16 bytes: a wave reads 1 KiB of
// contiguous JBLOCKs), dequantizes it and parks it in LDS (rows padded to 9 words); lane
// (block, column) runs the column pass; lane (row, block) runs the row pass on its workspace row,
// range-limits and stores 8 samples as one 8-byte store (8 lanes = 64 contiguous bytes of a row).
// When every dequantized coefficient of the tile is below 2^13 in magnitude -- always, for
// coefficients that came out of an 8-bit forward DCT -- all butterfly operands are below 2^23 and
// the multiplies are the full-rate 24-bit ones; otherwise (corrupt streams) the tile takes the
// 32-bit multiply path, whose wrap-around equals libjpeg's INT32 arithmetic.
#include "idct_core.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

using namespace idct;

constexpr int kBlock = 256;

struct DequantArgs {
  uint16_t q[64];  // natural order
};

__global__ __launch_bounds__(kBlock) void idct_dequant_kernel(const int16_t* __restrict__ coef, int bw, int bh,
                                                              const DequantArgs qa, uint8_t* __restrict__ plane,
                                                              size_t stride) {
  __shared__ int s_ws[kBlock / 64][8 * 8 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* ws = s_ws[wv];
  const int groups_x = (bw + 7) >> 3, total = groups_x * bh;
  const int gwave = blockIdx.x * (kBlock / 64) + wv, nwaves = gridDim.x * (kBlock / 64);
  const int rr = lane >> 3, rb = lane & 7;  // load / row-pass / store role: (row, block)
  int q[8];
#pragma unroll
  for (int c = 0; c < 8; c++) q[c] = qa.q[rr * 8 + c];
  const bool vec_store = ((stride | (uintptr_t)plane) & 7) == 0;

  for (int t = gwave; t < total; t += nwaves) {
    const int by = t / groups_x, gx = t - by * groups_x;
    const int bx = gx * 8 + rb;
    int v[8];
    int big = 0;
    load_dequant_row(coef, bw, by, bx, rr, q, v, big);
    uint32_t s[8];
    idct_wave(ws, v, big, rr, rb, s);
    if (bx < bw) {
      const uint32_t lo = s[0] | (s[1] << 8) | (s[2] << 16) | (s[3] << 24);
      const uint32_t hi = s[4] | (s[5] << 8) | (s[6] << 16) | (s[7] << 24);
      uint8_t* dst = plane + (size_t)(by * 8 + rr) * stride + (size_t)bx * 8;
      if (vec_store) {
        *(uint2*)dst = make_uint2(lo, hi);
      } else {
#pragma unroll
        for (int c = 0; c < 4; c++) { dst[c] = (uint8_t)(lo >> (8 * c)); dst[4 + c] = (uint8_t)(hi >> (8 * c)); }
      }
    }
  }
}

// ---- libjpeg colour conversions (16-bit fixed point) -----------------------------------------------
#define FIX16(x) ((int)((x) * 65536.0 + 0.5))

// jccolor.c rgb_ycc_convert.  (The 6b / libjpeg-turbo constants and IJG 9's longer ones give the same
// result for every 8-bit (r, g, b): checked exhaustively, tests/test_host_logic.py.)
__device__ __forceinline__ void rgb_to_ycc_px(uint32_t r, uint32_t g, uint32_t b, uint32_t& y, uint32_t& cb, uint32_t& cr) {
  const int half = 1 << 15, off = 128 << 16;
  const int ri = (int)r, gi = (int)g, bi = (int)b;
  y = (uint32_t)((FIX16(0.29900) * ri + FIX16(0.58700) * gi + FIX16(0.11400) * bi + half) >> 16);
  cb = (uint32_t)(((-FIX16(0.16874)) * ri + (-FIX16(0.33126)) * gi + FIX16(0.50000) * bi + off + half - 1) >> 16);
  cr = (uint32_t)((FIX16(0.50000) * ri + (-FIX16(0.41869)) * gi + (-FIX16(0.08131)) * bi + off + half - 1) >> 16);
}

struct JpegColorParams {
  const uint8_t* rgb;   // packed RGB888 / RGBA8888
  uint8_t* rgb_out;
  const uint8_t* p[3];  // planes (ycc -> rgb input)
  uint8_t* po[3];       // planes (rgb -> ycc output)
  uint32_t w, h, rgb_stride_px, stride[3];
  int bpp;              // 3 or 4
  int k_cr_g, k_cb_g;   // ycc -> rgb green constants (libjpeg variant)
};

// one lane = 4 horizontally adjacent pixels when VEC (w % 4 == 0, 4-byte aligned rows), else 1 pixel
template <int BPP, bool VEC>
__global__ __launch_bounds__(kBlock) void jpeg_rgb_to_ycc_kernel(const JpegColorParams p) {
  const uint32_t per_row = VEC ? p.w / 4 : p.w;
  const uint32_t tiles_x = (per_row + kBlock - 1) / kBlock, tiles = tiles_x * p.h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, j = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (j >= per_row) continue;
    const uint8_t* src = p.rgb + (size_t)y * p.rgb_stride_px * BPP;
    if constexpr (VEC) {
      uint32_t r[4], g[4], b[4];
      if constexpr (BPP == 4) {
        const uint4 v = *(const uint4*)(src + j * 16);
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) { r[k] = w4[k] & 0xff; g[k] = (w4[k] >> 8) & 0xff; b[k] = (w4[k] >> 16) & 0xff; }
      } else {
        const uint32_t* s3 = (const uint32_t*)(src + j * 12);
        const uint32_t a = s3[0], bq = s3[1], c = s3[2];  // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
        r[0] = a & 0xff; g[0] = (a >> 8) & 0xff; b[0] = (a >> 16) & 0xff;
        r[1] = a >> 24; g[1] = bq & 0xff; b[1] = (bq >> 8) & 0xff;
        r[2] = (bq >> 16) & 0xff; g[2] = bq >> 24; b[2] = c & 0xff;
        r[3] = (c >> 8) & 0xff; g[3] = (c >> 16) & 0xff; b[3] = c >> 24;
      }
      uint32_t yo = 0, cbo = 0, cro = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t yy, cb, cr;
        rgb_to_ycc_px(r[k], g[k], b[k], yy, cb, cr);
        yo |= yy << (8 * k); cbo |= cb << (8 * k); cro |= cr << (8 * k);
      }
      *(uint32_t*)(p.po[0] + (size_t)y * p.stride[0] + j * 4) = yo;
      *(uint32_t*)(p.po[1] + (size_t)y * p.stride[1] + j * 4) = cbo;
      *(uint32_t*)(p.po[2] + (size_t)y * p.stride[2] + j * 4) = cro;
    } else {
      uint32_t yy, cb, cr;
      rgb_to_ycc_px(src[j * BPP], src[j * BPP + 1], src[j * BPP + 2], yy, cb, cr);
      p.po[0][(size_t)y * p.stride[0] + j] = (uint8_t)yy;
      p.po[1][(size_t)y * p.stride[1] + j] = (uint8_t)cb;
      p.po[2][(size_t)y * p.stride[2] + j] = (uint8_t)cr;
    }
  }
}

// jdcolor.c ycc_rgb_convert: r = y + ((FIX(1.402) v + half) >> 16), b likewise with 1.772 u,
// g = y + ((-k_cb_g u + half - k_cr_g v) >> 16), each clamped to [0, 255]
__device__ __forceinline__ uint32_t clamp255(int v) { return (uint32_t)min(max(v, 0), 255); }
__device__ __forceinline__ uint32_t ycc_to_rgb_px(uint32_t y, uint32_t cb, uint32_t cr, int k_cr_g, int k_cb_g) {
  const int half = 1 << 15;
  const int yy = (int)y, u = (int)cb - 128, v = (int)cr - 128;
  const uint32_t r = clamp255(yy + ((FIX16(1.40200) * v + half) >> 16));
  const uint32_t g = clamp255(yy + (((-k_cb_g) * u + half + (-k_cr_g) * v) >> 16));
  const uint32_t b = clamp255(yy + ((FIX16(1.77200) * u + half) >> 16));
  return r | (g << 8) | (b << 16) | (255u << 24);
}

template <int BPP, bool VEC>
__global__ __launch_bounds__(kBlock) void jpeg_ycc_to_rgb_kernel(const JpegColorParams p) {
  const uint32_t per_row = VEC ? p.w / 4 : p.w;
  const uint32_t tiles_x = (per_row + kBlock - 1) / kBlock, tiles = tiles_x * p.h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, j = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (j >= per_row) continue;
    uint8_t* dst = p.rgb_out + (size_t)y * p.rgb_stride_px * BPP;
    if constexpr (VEC) {
      const uint32_t yv = *(const uint32_t*)(p.p[0] + (size_t)y * p.stride[0] + j * 4);
      const uint32_t uv = *(const uint32_t*)(p.p[1] + (size_t)y * p.stride[1] + j * 4);
      const uint32_t vv = *(const uint32_t*)(p.p[2] + (size_t)y * p.stride[2] + j * 4);
      uint32_t px[4];
#pragma unroll
      for (int k = 0; k < 4; k++)
        px[k] = ycc_to_rgb_px((yv >> (8 * k)) & 0xff, (uv >> (8 * k)) & 0xff, (vv >> (8 * k)) & 0xff, p.k_cr_g, p.k_cb_g);
      if constexpr (BPP == 4) {
        *(uint4*)(dst + j * 16) = make_uint4(px[0], px[1], px[2], px[3]);
      } else {
        uint32_t* d3 = (uint32_t*)(dst + j * 12);
        d3[0] = (px[0] & 0xffffff) | (px[1] << 24);
        d3[1] = ((px[1] >> 8) & 0xffff) | (px[2] << 16);
        d3[2] = ((px[2] >> 16) & 0xff) | (px[3] << 8);
      }
    } else {
      const uint32_t px = ycc_to_rgb_px(p.p[0][(size_t)y * p.stride[0] + j], p.p[1][(size_t)y * p.stride[1] + j],
                                        p.p[2][(size_t)y * p.stride[2] + j], p.k_cr_g, p.k_cb_g);
      dst[j * BPP] = (uint8_t)px; dst[j * BPP + 1] = (uint8_t)(px >> 8); dst[j * BPP + 2] = (uint8_t)(px >> 16);
      if constexpr (BPP == 4) dst[j * BPP + 3] = 255;
    }
  }
}

// ---- 3-channel gain map: dequant + IDCT of Y, Cb, Cr and ycc_rgb_convert in one pass ------------------
// The decode-side mirror of fdct_quant_rgb_kernel: a 3-channel map is a 4:4:4 JPEG (the encoder hands it to
// libjpeg as JCS_RGB, jpegencoderhelper.cpp:165-167), and the decoder asks libjpeg for RGB / RGBA scanlines
// (jpegdecoderhelper.cpp:400-470).  Unfused that is three IDCT launches and a colour-conversion launch,
// 6 + 3 + 3 + 4 = 16 B/px of traffic; here a wave runs the three inverse transforms of its eight blocks back to
// back through the same LDS workspace, keeps the 3 x 8 samples of its row in registers, converts and stores
// 8 pixels (32 or 24 contiguous bytes per lane, 256 / 192 per eight lanes): 6 B/px in, 4 (3) B/px out.
struct IdctRgbArgs {
  const int16_t* coef[3];
  uint8_t* rgb;
  size_t pitch;       // bytes
  uint32_t w, h;      // pixels actually stored (<= blocks * 8)
  int bw, bh;
  int k_cr_g, k_cb_g;
  uint16_t q[2][64];  // luma, chroma (natural order)
};

template <int BPP>
__global__ __launch_bounds__(kBlock) void idct_dequant_rgb_kernel(const IdctRgbArgs a) {
  __shared__ int s_ws[kBlock / 64][8 * 8 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* ws = s_ws[wv];
  const int bw = a.bw;
  const int groups_x = (bw + 7) >> 3, total = groups_x * a.bh;
  const int gwave = blockIdx.x * (kBlock / 64) + wv, nwaves = gridDim.x * (kBlock / 64);
  const int rr = lane >> 3, rb = lane & 7;
  int ql[8], qc[8];
#pragma unroll
  for (int c = 0; c < 8; c++) { ql[c] = a.q[0][rr * 8 + c]; qc[c] = a.q[1][rr * 8 + c]; }
  const bool vec_ok = ((a.pitch | (uintptr_t)a.rgb) & (BPP == 4 ? 15 : 7)) == 0;

  for (int t = gwave; t < total; t += nwaves) {
    const int by = t / groups_x, gx = t - by * groups_x;
    const int bx = gx * 8 + rb;
    // all three coefficient rows are requested before the first transform starts
    int vy[8], vb[8], vr[8];
    int big_y = 0, big_b = 0, big_r = 0;
    load_dequant_row(a.coef[0], bw, by, bx, rr, ql, vy, big_y);
    load_dequant_row(a.coef[1], bw, by, bx, rr, qc, vb, big_b);
    load_dequant_row(a.coef[2], bw, by, bx, rr, qc, vr, big_r);
    uint32_t sy[8], sb[8], sr[8];
    idct_wave(ws, vy, big_y, rr, rb, sy);
    idct_wave(ws, vb, big_b, rr, rb, sb);
    idct_wave(ws, vr, big_r, rr, rb, sr);
    const uint32_t y = (uint32_t)(by * 8 + rr), x0 = (uint32_t)bx * 8;
    if (bx < bw && y < a.h && x0 < a.w) {
      uint32_t px[8];
#pragma unroll
      for (int c = 0; c < 8; c++) px[c] = ycc_to_rgb_px(sy[c], sb[c], sr[c], a.k_cr_g, a.k_cb_g);
      uint8_t* dst = a.rgb + (size_t)y * a.pitch + (size_t)x0 * BPP;
      if (vec_ok && x0 + 8 <= a.w) {
        if constexpr (BPP == 4) {
          *(uint4*)dst = make_uint4(px[0], px[1], px[2], px[3]);
          *(uint4*)(dst + 16) = make_uint4(px[4], px[5], px[6], px[7]);
        } else {
          uint32_t d[6];
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const uint32_t* q4 = px + 4 * h;
            d[3 * h + 0] = (q4[0] & 0xffffff) | (q4[1] << 24);
            d[3 * h + 1] = ((q4[1] >> 8) & 0xffff) | (q4[2] << 16);
            d[3 * h + 2] = ((q4[2] >> 16) & 0xff) | (q4[3] << 8);
          }
          *(uint2*)dst = make_uint2(d[0], d[1]);
          *(uint2*)(dst + 8) = make_uint2(d[2], d[3]);
          *(uint2*)(dst + 16) = make_uint2(d[4], d[5]);
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; c++) {
          if (x0 + c < a.w) {
            dst[c * BPP] = (uint8_t)px[c]; dst[c * BPP + 1] = (uint8_t)(px[c] >> 8); dst[c * BPP + 2] = (uint8_t)(px[c] >> 16);
            if constexpr (BPP == 4) dst[c * BPP + 3] = 255;
          }
        }
      }
    }
  }
}

int resident_grid(uint32_t tiles, int per_cu) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const uint32_t r = (uint32_t)(cus * per_cu);
  return (int)(tiles < r ? (tiles ? tiles : 1u) : r);
}

inline bool al(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

}  // namespace

hipError_t launch_idct_dequant(const int16_t* coef, int bw, int bh, const uint16_t* qt_host, uint8_t* plane,
                               size_t stride, hipStream_t s) {
  DequantArgs qa;
  for (int i = 0; i < 64; i++) qa.q[i] = qt_host[i];
  const int total = ((bw + 7) / 8) * bh;
  const int grid = resident_grid((uint32_t)(total + 3) / 4, 8);
  hipLaunchKernelGGL(idct_dequant_kernel, dim3(grid), dim3(kBlock), 0, s, coef, bw, bh, qa, plane, stride);
  return hipGetLastError();
}

hipError_t launch_idct_dequant_rgb(const int16_t* coef_y, const int16_t* coef_cb, const int16_t* coef_cr, int bw, int bh,
                                   const uint16_t* qt_luma_host, const uint16_t* qt_chroma_host, int variant,
                                   const ImageViewMut& rgb, hipStream_t s) {
  IdctRgbArgs a = {};
  a.coef[0] = coef_y; a.coef[1] = coef_cb; a.coef[2] = coef_cr;
  a.rgb = (uint8_t*)rgb.p[0];
  const int bpp = rgb.fmt == UHDR_IMG_FMT_32bppRGBA8888 ? 4 : 3;
  a.pitch = (size_t)rgb.stride[0] * bpp;
  a.w = rgb.w; a.h = rgb.h; a.bw = bw; a.bh = bh;
  a.k_cr_g = variant ? FIX16(0.714136286) : FIX16(0.71414);
  a.k_cb_g = variant ? FIX16(0.344136286) : FIX16(0.34414);
  for (int i = 0; i < 64; i++) { a.q[0][i] = qt_luma_host[i]; a.q[1][i] = qt_chroma_host[i]; }
  const int total = ((bw + 7) / 8) * bh;
  const int grid = resident_grid((uint32_t)(total + 3) / 4, 8);
  if (bpp == 4) hipLaunchKernelGGL((idct_dequant_rgb_kernel<4>), dim3(grid), dim3(kBlock), 0, s, a);
  else hipLaunchKernelGGL((idct_dequant_rgb_kernel<3>), dim3(grid), dim3(kBlock), 0, s, a);
  return hipGetLastError();
}

// rgb: 24bppRGB888 or 32bppRGBA8888 -> ycc: three 8-bit planes of the same size
hipError_t launch_jpeg_rgb_to_ycc(const ImageView& rgb, const ImageViewMut& ycc, hipStream_t s) {
  JpegColorParams p = {};
  p.rgb = (const uint8_t*)rgb.p[0];
  p.w = rgb.w; p.h = rgb.h; p.rgb_stride_px = rgb.stride[0];
  p.bpp = rgb.fmt == UHDR_IMG_FMT_32bppRGBA8888 ? 4 : 3;
  for (int c = 0; c < 3; c++) { p.po[c] = (uint8_t*)ycc.p[c]; p.stride[c] = ycc.stride[c]; }
  const bool vec = (p.w % 4 == 0) && al(p.rgb, p.bpp == 4 ? 16 : 4) && ((size_t)p.rgb_stride_px * p.bpp) % (p.bpp == 4 ? 16 : 4) == 0 &&
                   al(p.po[0], 4) && al(p.po[1], 4) && al(p.po[2], 4) && p.stride[0] % 4 == 0 && p.stride[1] % 4 == 0 && p.stride[2] % 4 == 0;
  const uint32_t per_row = vec ? p.w / 4 : p.w;
  const int grid = resident_grid(((per_row + kBlock - 1) / kBlock) * p.h, 8);
  if (p.bpp == 4) {
    if (vec) hipLaunchKernelGGL((jpeg_rgb_to_ycc_kernel<4, true>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((jpeg_rgb_to_ycc_kernel<4, false>), dim3(grid), dim3(kBlock), 0, s, p);
  } else {
    if (vec) hipLaunchKernelGGL((jpeg_rgb_to_ycc_kernel<3, true>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((jpeg_rgb_to_ycc_kernel<3, false>), dim3(grid), dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

// variant 0: libjpeg 6b / libjpeg-turbo constants (the reference's pinned libjpeg-turbo); 1: IJG 9
hipError_t launch_jpeg_ycc_to_rgb(const ImageView& ycc, const ImageViewMut& rgb, int variant, hipStream_t s) {
  JpegColorParams p = {};
  p.rgb_out = (uint8_t*)rgb.p[0];
  p.w = ycc.w; p.h = ycc.h; p.rgb_stride_px = rgb.stride[0];
  p.bpp = rgb.fmt == UHDR_IMG_FMT_32bppRGBA8888 ? 4 : 3;
  for (int c = 0; c < 3; c++) { p.p[c] = (const uint8_t*)ycc.p[c]; p.stride[c] = ycc.stride[c]; }
  p.k_cr_g = variant ? FIX16(0.714136286) : FIX16(0.71414);
  p.k_cb_g = variant ? FIX16(0.344136286) : FIX16(0.34414);
  const bool vec = (p.w % 4 == 0) && al(p.rgb_out, p.bpp == 4 ? 16 : 4) && ((size_t)p.rgb_stride_px * p.bpp) % (p.bpp == 4 ? 16 : 4) == 0 &&
                   al(p.p[0], 4) && al(p.p[1], 4) && al(p.p[2], 4) && p.stride[0] % 4 == 0 && p.stride[1] % 4 == 0 && p.stride[2] % 4 == 0;
  const uint32_t per_row = vec ? p.w / 4 : p.w;
  const int grid = resident_grid(((per_row + kBlock - 1) / kBlock) * p.h, 8);
  if (p.bpp == 4) {
    if (vec) hipLaunchKernelGGL((jpeg_ycc_to_rgb_kernel<4, true>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((jpeg_ycc_to_rgb_kernel<4, false>), dim3(grid), dim3(kBlock), 0, s, p);
  } else {
    if (vec) hipLaunchKernelGGL((jpeg_ycc_to_rgb_kernel<3, true>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((jpeg_ycc_to_rgb_kernel<3, false>), dim3(grid), dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

}  // namespace uhdr
