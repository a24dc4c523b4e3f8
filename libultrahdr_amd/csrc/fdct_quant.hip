// 8x8 forward DCT + quantize on gfx950, bit-exact to libjpeg's JDCT_ISLOW path.
//
// The arithmetic is NOT in the reference tree: libultrahdr hands planes to libjpeg
// (/root/reference/lib/src/jpegencoderhelper.cpp:187-198 sets jpeg_set_quality(q, TRUE) and
// dct_method = JDCT_ISLOW; :297 jpeg_write_raw_data).  What runs there is the public
// Loeffler-Ligtenberg-Moschytz integer DCT (libjpeg jfdctint.c: CONST_BITS 13, PASS1_BITS 2, a
// row pass scaled by 4 then a column pass, round-to-nearest descales) followed by jcdctmgr.c's
// round-half-away division by (quantval << 3).  Integers only => results are exact.
//
// Mapping: one wavefront = 8 horizontally adjacent blocks (64 x 8 samples).  Row pass: lane
// (row r = lane/8, block b = lane%8) loads its 8 samples with one 8-byte load -- 8 consecutive
// lanes read 64 contiguous bytes -- and runs the 1-D butterfly in registers.  The 8x8 tiles are
// transposed through LDS (row-padded to 9 words: conflict-free column reads), the column pass and
// the quantizer run with lane = (block, column), and a second LDS transpose lets every lane emit
// one coefficient ROW (8 x int16 = 16 bytes), so a wave writes 1 KiB contiguous in libjpeg's
// JBLOCK order.  The quantizer divides by multiplying with ceil(2^32 / q) (exact for the 19-bit
// magnitudes that occur; q <= 2040); divisors and their reciprocals are computed on the host and
// travel in the kernel arguments.  Every butterfly operand is below 2^23 in magnitude (samples
// +-128 -> row pass <= 2^13 -> column pass <= 2^16; constants < 2^15), so the multiplies are the
// full-rate 24-bit ones (v_mul_i32_i24 returns the low 32 bits of the exact product).
// A wave keeps walking tiles (grid = resident waves) so the set-up is paid once.
#include <string.h>

#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;  // 4 waves = 32 blocks per workgroup iteration

#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

template <int PASS>
__device__ __forceinline__ void fdct_1d(const int in[8], int out[8]) {
  constexpr int sh = PASS == 0 ? 13 - 2 : 13 + 2;
  int t0 = in[0] + in[7], t7 = in[0] - in[7], t1 = in[1] + in[6], t6 = in[1] - in[6];
  int t2 = in[2] + in[5], t5 = in[2] - in[5], t3 = in[3] + in[4], t4 = in[3] - in[4];
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  if (PASS == 0) {
    out[0] = (t10 + t11) * 4;
    out[4] = (t10 - t11) * 4;
  } else {
    out[0] = descale(t10 + t11, 2);
    out[4] = descale(t10 - t11, 2);
  }
  int z1 = __mul24(t12 + t13, FIX_0_541196100);
  out[2] = descale(z1 + __mul24(t13, FIX_0_765366865), sh);
  out[6] = descale(z1 + __mul24(t12, -FIX_1_847759065), sh);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = __mul24(z3 + z4, FIX_1_175875602);
  t4 = __mul24(t4, FIX_0_298631336);
  t5 = __mul24(t5, FIX_2_053119869);
  t6 = __mul24(t6, FIX_3_072711026);
  t7 = __mul24(t7, FIX_1_501321110);
  z1 = __mul24(z1, -FIX_0_899976223);
  z2 = __mul24(z2, -FIX_2_562915447);
  z3 = __mul24(z3, -FIX_1_961570560);
  z4 = __mul24(z4, -FIX_0_390180644);
  z3 += z5;
  z4 += z5;
  out[7] = descale(t4 + z1 + z3, sh);
  out[5] = descale(t5 + z2 + z4, sh);
  out[3] = descale(t6 + z2 + z3, sh);
  out[1] = descale(t7 + z1 + z4, sh);
}

struct QuantArgs {
  uint32_t qv[64];  // quantval << 3, natural order
  uint32_t qm[64];  // ceil(2^32 / qv)
};

// The eight samples of row y, columns x0 .. x0 + 7 of a block that reaches beyond the plane's w x h valid samples, by
// JpegEncoderHelper::compressYCbCr's rules (jpegencoderhelper.cpp:246-309; FdctEdge in uhdr_types.h):
//   the caller's row pitch covers the block-aligned width (col_mode 0): columns >= w are the bytes that sit there (the
//     helper hands libjpeg pointers into the caller's rows), rows >= h are a constant row of `fill` (mPlanesMCURow);
//   it does not (col_mode 1): the helper copies every row into a scratch MCU row whose tail is `fill`; rows >= h are never
//     written, they still hold what the PREVIOUS MCU row copied there -- row y - mcu_rows of the plane (zeros in the first).
__device__ __forceinline__ void load8_edge(const uint8_t* plane, size_t stride, int x0, int y, const FdctEdge& e, int in[8]) {
  int ys = y;
  bool row_const = false;
  if (y >= e.h) {
    if (e.col_mode == 0) row_const = true;
    else ys = y - e.mcu_rows;
  }
#pragma unroll
  for (int c = 0; c < 8; c++) {
    const int x = x0 + c;
    int v;
    if (row_const) v = e.fill;
    else if (x >= e.w) v = e.col_mode == 0 ? (int)plane[(size_t)ys * stride + x] : e.fill;
    else v = ys < 0 ? 0 : (int)plane[(size_t)ys * stride + x];
    in[c] = v - 128;
  }
}

__global__ __launch_bounds__(kBlock) void fdct_quant_kernel(const uint8_t* __restrict__ plane,
                                                            size_t stride, int bw, int bh,
                                                            const QuantArgs qa,
                                                            int16_t* __restrict__ coef, const FdctEdge edge) {
  // per wave: 8 blocks x 8 rows x 9 (padded) words
  __shared__ int s_ws[kBlock / 64][8 * 8 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* ws = s_ws[wv];
  const int groups_x = (bw + 7) >> 3;              // groups of 8 blocks per block-row
  const int total = groups_x * bh;
  const int gwave = blockIdx.x * (kBlock / 64) + wv;
  const int nwaves = gridDim.x * (kBlock / 64);

  // column-pass role: lane = (block cb, column cc); its 8 divisors never change
  const int cb = lane >> 3, cc = lane & 7;
  uint32_t qv[8], qm[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    qv[r] = qa.qv[r * 8 + cc];
    qm[r] = qa.qm[r * 8 + cc];
  }
  // row-pass / store role: lane = (row rr, block rb)
  const int rr = lane >> 3, rb = lane & 7;

  for (int t = gwave; t < total; t += nwaves) {
    const int by = t / groups_x, gx = t - by * groups_x;
    const int bx = gx * 8 + rb;
    int in[8], out[8];
    if (bx < bw && edge.on && (bx * 8 + 8 > edge.w || by * 8 + 8 > edge.h)) {  // a partial block of the right / bottom edge
      load8_edge(plane, stride, bx * 8, by * 8 + rr, edge, in);
      fdct_1d<0>(in, out);
    } else if (bx < bw) {
      const uint8_t* src = plane + (size_t)(by * 8 + rr) * stride + (size_t)bx * 8;
      uint32_t lo, hi;
      if (((uintptr_t)src & 3) == 0) {
        lo = ((const uint32_t*)src)[0];
        hi = ((const uint32_t*)src)[1];
      } else {
        lo = src[0] | (src[1] << 8) | (src[2] << 16) | ((uint32_t)src[3] << 24);
        hi = src[4] | (src[5] << 8) | (src[6] << 16) | ((uint32_t)src[7] << 24);
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        in[c] = (int)((lo >> (8 * c)) & 0xff) - 128;
        in[4 + c] = (int)((hi >> (8 * c)) & 0xff) - 128;
      }
      fdct_1d<0>(in, out);
    } else {
#pragma unroll
      for (int c = 0; c < 8; c++) out[c] = 0;
    }
#pragma unroll
    for (int c = 0; c < 8; c++) ws[rb * 72 + rr * 9 + c] = out[c];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave have landed
    // column pass
#pragma unroll
    for (int r = 0; r < 8; r++) in[r] = ws[cb * 72 + r * 9 + cc];
    fdct_1d<1>(in, out);
#pragma unroll
    for (int r = 0; r < 8; r++) {  // jcdctmgr.c forward_DCT quantizer: sign(v) * ((|v| + q/2) / q)
      // branch-free: |v| through the sign mask, the quotient through the reciprocal (0 for |v| + q/2 < q by
      // itself: a * ceil(2^32 / q) < 2^32 whenever a < q < 65536), the sign put back the same way
      const int v = out[r];
      const int sgn = v >> 31;
      const uint32_t a = (uint32_t)((v ^ sgn) - sgn) + (qv[r] >> 1);
      const uint32_t q = __umulhi(a, qm[r]);
      out[r] = (int)(q ^ (uint32_t)sgn) - sgn;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < 8; r++) ws[cb * 72 + r * 9 + cc] = out[r];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (bx < bw) {
      int v[8];
#pragma unroll
      for (int c = 0; c < 8; c++) v[c] = ws[rb * 72 + rr * 9 + c];
      uint4 o;
      o.x = (uint32_t)(v[0] & 0xffff) | ((uint32_t)v[1] << 16);
      o.y = (uint32_t)(v[2] & 0xffff) | ((uint32_t)v[3] << 16);
      o.z = (uint32_t)(v[4] & 0xffff) | ((uint32_t)v[5] << 16);
      o.w = (uint32_t)(v[6] & 0xffff) | ((uint32_t)v[7] << 16);
      *(uint4*)(coef + ((size_t)by * bw + bx) * 64 + rr * 8) = o;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- 3-channel gain map: libjpeg's JCS_RGB -> YCbCr conversion + FDCT + quantize in one pass -----------------
// What jpeg_write_scanlines does to an RGB888 gain map (jpegencoderhelper.cpp:165-167, 212-225): jccolor.c
// rgb_ycc_convert, then the islow FDCT and the quantizer per component (luma table for Y, chroma table for
// Cb / Cr, 4:4:4).  Unfused that is uhdr_hip_jpeg_rgb_to_ycc (3 B in, 3 B out) followed by three
// uhdr_hip_fdct_quant launches (1 B in, 2 B out each): 15 B/px.  Here a lane reads its 8 pixels once
// (24 or 32 bytes), converts them in registers and runs the three transforms back to back through the same
// LDS workspace: 3 (4) B/px in, 6 B/px out.
struct QuantArgs2 {
  QuantArgs y, c;
};
#define FIX16(x) ((int)((x) * 65536.0 + 0.5))

template <int BPP>
__global__ __launch_bounds__(kBlock) void fdct_quant_rgb_kernel(const uint8_t* __restrict__ rgb, size_t pitch /* bytes */, int bw, int bh,
                                                                const QuantArgs2 qa, int16_t* __restrict__ coef_y,
                                                                int16_t* __restrict__ coef_cb, int16_t* __restrict__ coef_cr, int vw, int vh) {
  // vw x vh: the image's valid pixels.  Blocks that reach beyond them repeat the last column / row -- what libjpeg's
  // jpeg_write_scanlines pipeline does to an RGB gain map (jcsample.c expand_right_edge, jcprepct.c expand_bottom_edge;
  // the colour conversion is per pixel, so replicating before or after it is the same).
  __shared__ int s_ws[kBlock / 64][8 * 8 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* ws = s_ws[wv];
  const int groups_x = (bw + 7) >> 3, total = groups_x * bh;
  const int gwave = blockIdx.x * (kBlock / 64) + wv, nwaves = gridDim.x * (kBlock / 64);
  const int cb = lane >> 3, cc = lane & 7;  // column-pass role
  const int rr = lane >> 3, rb = lane & 7;  // row-pass / store role
  uint32_t qvy[8], qmy[8], qvc[8], qmc[8];
#pragma unroll
  for (int r = 0; r < 8; r++) {
    qvy[r] = qa.y.qv[r * 8 + cc]; qmy[r] = qa.y.qm[r * 8 + cc];
    qvc[r] = qa.c.qv[r * 8 + cc]; qmc[r] = qa.c.qm[r * 8 + cc];
  }
  for (int t = gwave; t < total; t += nwaves) {
    const int by = t / groups_x, gx = t - by * groups_x;
    const int bx = gx * 8 + rb;
    int comp[3][8];
    const bool partial = bx * 8 + 8 > vw || by * 8 + 8 > vh;
    if (bx < bw && partial) {
      const int y = min(by * 8 + rr, vh - 1);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint8_t* q = rgb + (size_t)y * pitch + (size_t)min(bx * 8 + k, vw - 1) * BPP;
        const int r = q[0], g = q[1], b = q[2];
        const int half = 1 << 15, off = 128 << 16;
        comp[0][k] = ((__mul24(FIX16(0.29900), r) + __mul24(FIX16(0.58700), g) + __mul24(FIX16(0.11400), b) + half) >> 16) - 128;
        comp[1][k] = ((__mul24(-FIX16(0.16874), r) + __mul24(-FIX16(0.33126), g) + __mul24(FIX16(0.50000), b) + off + half - 1) >> 16) - 128;
        comp[2][k] = ((__mul24(FIX16(0.50000), r) + __mul24(-FIX16(0.41869), g) + __mul24(-FIX16(0.08131), b) + off + half - 1) >> 16) - 128;
      }
    } else if (bx < bw) {
      const uint8_t* src = rgb + (size_t)(by * 8 + rr) * pitch + (size_t)bx * 8 * BPP;
      uint32_t w[2 * BPP];
      if constexpr (BPP == 4) {
        const uint4 a = ((const uint4*)src)[0], b = ((const uint4*)src)[1];
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
      } else {
        const uint2 a = ((const uint2*)src)[0], b = ((const uint2*)src)[1], c = ((const uint2*)src)[2];
        w[0] = a.x; w[1] = a.y; w[2] = b.x; w[3] = b.y; w[4] = c.x; w[5] = c.y;
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        int r, g, b;
        if constexpr (BPP == 4) {
          r = w[k] & 0xff; g = (w[k] >> 8) & 0xff; b = (w[k] >> 16) & 0xff;
        } else {  // byte 3k, 3k+1, 3k+2 of the 24-byte run
          auto byte_at = [&](int i) { return (int)((w[i >> 2] >> (8 * (i & 3))) & 0xff); };
          r = byte_at(3 * k); g = byte_at(3 * k + 1); b = byte_at(3 * k + 2);
        }
        // jccolor.c rgb_ycc_convert (see jpeg_decode.hip: both published constant sets give these bytes)
        const int half = 1 << 15, off = 128 << 16;
        comp[0][k] = ((__mul24(FIX16(0.29900), r) + __mul24(FIX16(0.58700), g) + __mul24(FIX16(0.11400), b) + half) >> 16) - 128;
        comp[1][k] = ((__mul24(-FIX16(0.16874), r) + __mul24(-FIX16(0.33126), g) + __mul24(FIX16(0.50000), b) + off + half - 1) >> 16) - 128;
        comp[2][k] = ((__mul24(FIX16(0.50000), r) + __mul24(-FIX16(0.41869), g) + __mul24(-FIX16(0.08131), b) + off + half - 1) >> 16) - 128;
      }
    }
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      int in[8], out[8];
      if (bx < bw) {
        fdct_1d<0>(comp[ci], out);
      } else {
#pragma unroll
        for (int c = 0; c < 8; c++) out[c] = 0;
      }
#pragma unroll
      for (int c = 0; c < 8; c++) ws[rb * 72 + rr * 9 + c] = out[c];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
      for (int r = 0; r < 8; r++) in[r] = ws[cb * 72 + r * 9 + cc];
      fdct_1d<1>(in, out);
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const int v = out[r];
        const int sgn = v >> 31;
        const uint32_t qv = ci == 0 ? qvy[r] : qvc[r], qm = ci == 0 ? qmy[r] : qmc[r];
        const uint32_t a = (uint32_t)((v ^ sgn) - sgn) + (qv >> 1);
        const uint32_t q = __umulhi(a, qm);
        out[r] = (int)(q ^ (uint32_t)sgn) - sgn;
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int r = 0; r < 8; r++) ws[cb * 72 + r * 9 + cc] = out[r];
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_s_waitcnt(0xc07f);
      if (bx < bw) {
        int v[8];
#pragma unroll
        for (int c = 0; c < 8; c++) v[c] = ws[rb * 72 + rr * 9 + c];
        uint4 o;
        o.x = (uint32_t)(v[0] & 0xffff) | ((uint32_t)v[1] << 16);
        o.y = (uint32_t)(v[2] & 0xffff) | ((uint32_t)v[3] << 16);
        o.z = (uint32_t)(v[4] & 0xffff) | ((uint32_t)v[5] << 16);
        o.w = (uint32_t)(v[6] & 0xffff) | ((uint32_t)v[7] << 16);
        int16_t* dst = ci == 0 ? coef_y : (ci == 1 ? coef_cb : coef_cr);
        *(uint4*)(dst + ((size_t)by * bw + bx) * 64 + rr * 8) = o;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

static void fill_quant_args(const uint16_t* qt_host, QuantArgs* qa) {
  for (int i = 0; i < 64; i++) {
    qa->qv[i] = (uint32_t)qt_host[i] << 3;
    qa->qm[i] = (uint32_t)((0x100000000ull + qa->qv[i] - 1) / qa->qv[i]);
  }
}

}  // namespace

hipError_t launch_fdct_quant(const uint8_t* plane, size_t stride, int bw, int bh,
                             const uint16_t* qt_host, int16_t* coef, hipStream_t s, const FdctEdge* edge) {
  FdctEdge e;
  memset(&e, 0, sizeof e);
  if (edge) e = *edge;
  QuantArgs qa;
  for (int i = 0; i < 64; i++) {
    qa.qv[i] = (uint32_t)qt_host[i] << 3;
    qa.qm[i] = (uint32_t)((0x100000000ull + qa.qv[i] - 1) / qa.qv[i]);
  }
  const int total = ((bw + 7) / 8) * bh;
  static const int resident = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 2048;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cus * 8;
  }();
  int grid = (total + 3) / 4;
  if (grid > resident) grid = resident;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(fdct_quant_kernel, dim3(grid), dim3(kBlock), 0, s, plane, stride, bw, bh, qa, coef, e);
  return hipGetLastError();
}

// bpp 3 (RGB888) or 4 (RGBA8888, alpha ignored); pitch in bytes; rows and base 8- (bpp 3) / 16-byte (bpp 4) aligned
hipError_t launch_fdct_quant_rgb(const uint8_t* rgb, size_t pitch, int bpp, int bw, int bh, const uint16_t* qt_luma_host,
                                 const uint16_t* qt_chroma_host, int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr, hipStream_t s, int vw, int vh) {
  if (vw <= 0) vw = bw * 8;
  if (vh <= 0) vh = bh * 8;
  QuantArgs2 qa;
  fill_quant_args(qt_luma_host, &qa.y);
  fill_quant_args(qt_chroma_host, &qa.c);
  const int total = ((bw + 7) / 8) * bh;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  int grid = (total + 3) / 4;
  if (grid > cus * 6) grid = cus * 6;
  if (grid < 1) grid = 1;
  if (bpp == 4) hipLaunchKernelGGL((fdct_quant_rgb_kernel<4>), dim3(grid), dim3(kBlock), 0, s, rgb, pitch, bw, bh, qa, coef_y, coef_cb, coef_cr, vw, vh);
  else hipLaunchKernelGGL((fdct_quant_rgb_kernel<3>), dim3(grid), dim3(kBlock), 0, s, rgb, pitch, bw, bh, qa, coef_y, coef_cb, coef_cr, vw, vh);
  return hipGetLastError();
}

}  // namespace uhdr
