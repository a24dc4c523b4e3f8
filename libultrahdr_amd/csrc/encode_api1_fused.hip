// API-1 encode chain with its round trips removed (MI355X extension, round 4; not reference operators).
//
// JpegR::encodeJPEGR API-1 (/root/reference/lib/src/jpegr.cpp:253-316) runs generateGainMap, compresses the map, converts the
// base image's YUV encoding (convertYuv, jpegr.cpp:436-518) and compresses it.  As separate kernels that was seven launches:
// pass 1, min/max reduce, pass 2 (writes 3 B/px of map), rgb->ycc + FDCT of the map (reads them back), convertYuv (rewrites
// 1.5 B/px), three FDCT launches over the base planes (read them back; a 4K chroma plane is too small to fill the part).
// Two kernels here replace five of them:
//
//   map_blocks_kernel   pass 2 of generateGainMap (ratio -> byte through the per-channel step tables, generate_gainmap.hip)
//                       + libjpeg's rgb_ycc_convert + the three FDCT / quantize transforms in one pass: 12 B/px of gain ratios in,
//                       6 B/px of coefficients out (+ 3 B/px if the caller wants the 8-bit map itself)
//   base_blocks_kernel  convertYuv + FDCT / quantize of Y, Cb, Cr in ONE launch: a wave takes two 16 x 16 MCUs, converts their
//                       128 quads (transformYuv420's arithmetic, gainmapmath.cpp:686-748) into an LDS tile and transforms
//                       the 8 luma + 4 chroma blocks from there; the converted planes never exist in HBM
//
// Coefficients are bit-identical to the unfused chain (tests/test_gpu_parity.py::test_api1_fused_chain_equals_the_operators).
#include <string.h>
#include "lds_copy.h"

#include "encode_core.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;  // 4 waves

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

// libjpeg jfdctint.c, one 8-point pass (see fdct_quant.hip for the derivation and the 24-bit multiply argument)
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

struct QuantPair {  // divisors (quantval << 3) and their reciprocals ceil(2^32 / q), natural order: luma table, chroma table
  uint32_t qv[2][64], qm[2][64];
};
static void fill_quant(const uint16_t* qt_luma, const uint16_t* qt_chroma, QuantPair* q) {
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 64; i++) {
      q->qv[t][i] = (uint32_t)(t ? qt_chroma : qt_luma)[i] << 3;
      q->qm[t][i] = (uint32_t)((0x100000000ull + q->qv[t][i] - 1) / q->qv[t][i]);
    }
}

// One wave, up to eight blocks: row pass results `out` of lane (row rr, block rb) -> transposed through the wave's LDS workspace
// -> column pass + quantizer with lane (block cb, column cc) -> transposed back -> lane (rr, rb) holds one coefficient row of
// its block in v[0..7].  `active`: the lane's block exists.  (jcdctmgr.c forward_DCT's quantizer, branch free: fdct_quant.hip)
// q: this table's 64 {divisor, reciprocal} pairs in LDS (natural order) -- the lane's eight pairs of column cc are read where they
// are used instead of living in 16 registers per table (map_blocks_kernel: 133 -> ~100 VGPRs)
__device__ __forceinline__ void column_pass_and_quantize(int* ws, int lane, const int row_out[8], const uint2* q, int v[8]) {
  const int cb = lane >> 3, cc = lane & 7, rr = lane >> 3, rb = lane & 7;
  int in[8], out[8];
#pragma unroll
  for (int c = 0; c < 8; c++) ws[rb * 72 + rr * 9 + c] = row_out[c];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS writes of this wave have landed
#pragma unroll
  for (int r = 0; r < 8; r++) in[r] = ws[cb * 72 + r * 9 + cc];
  fdct_1d<1>(in, out);
#pragma unroll
  for (int r = 0; r < 8; r++) {
    const int x = out[r];
    const int sgn = x >> 31;
    const uint2 qe = q[r * 8 + cc];
    const uint32_t a = (uint32_t)((x ^ sgn) - sgn) + (qe.x >> 1);
    const uint32_t qq = __umulhi(a, qe.y);
    out[r] = (int)(qq ^ (uint32_t)sgn) - sgn;
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 8; r++) ws[cb * 72 + r * 9 + cc] = out[r];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
  for (int c = 0; c < 8; c++) v[c] = ws[rb * 72 + rr * 9 + c];
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void store_coef_row(int16_t* dst, const int v[8]) {
  uint4 o;
  o.x = (uint32_t)(v[0] & 0xffff) | ((uint32_t)v[1] << 16);
  o.y = (uint32_t)(v[2] & 0xffff) | ((uint32_t)v[3] << 16);
  o.z = (uint32_t)(v[4] & 0xffff) | ((uint32_t)v[5] << 16);
  o.w = (uint32_t)(v[6] & 0xffff) | ((uint32_t)v[7] << 16);
  *(uint4*)dst = o;
}

// ---- the gain map: ratio plane -> coefficients ------------------------------------------------------------------------------------
struct MapBlocksParams {
  const float* ratio;      // map_w * map_h * nch gain ratios (pass 1)
  const AffineDev* dev;    // final range + step tables (minmax_table_kernel)
  const double* math_tab;  // per-sample evaluation when a channel has no table
  uint8_t* map_out;        // optional: the 8-bit map, nch bytes per pixel
  uint32_t out_stride;     // pixels
  int bw, bh;              // blocks (map_w / 8, map_h / 8)
  int16_t* coef[3];
  QuantPair q;
};

#define FIX16(x) ((int)((x) * 65536.0 + 0.5))
constexpr int kMapBlock = 512;  // eight waves share one copy of the step tables (24 KB): three workgroups = 24 waves per CU
template <int NCH>
__global__ __launch_bounds__(kMapBlock) void map_blocks_kernel(const MapBlocksParams p) {
  constexpr int kBlock = kMapBlock;  // (shadows the file's 256 inside this kernel)
  __shared__ int s_ws[kBlock / 64][8 * 8 * 9];
  __shared__ uint2 s_tab[NCH][kAffTabMax];
  __shared__ uint2 s_q[2][64];
  if (threadIdx.x < 128) s_q[threadIdx.x >> 6][threadIdx.x & 63] = uint2{p.q.qv[threadIdx.x >> 6][threadIdx.x & 63], p.q.qm[threadIdx.x >> 6][threadIdx.x & 63]};
  StepTab st[3];
  bool tabs = true;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const AffineTabDev& td = p.dev->tab[c];
    st[c].tab = nullptr;
    st[c].n = td.n; st[c].base8 = td.base8; st[c].shm3 = td.shm3; st[c].lo_bits = td.lo_bits; st[c].hi_bits = td.hi_bits;
    tabs = tabs && td.ok != 0;
  }
  if (tabs) {
    const uint2* src = (const uint2*)((const char*)p.dev + kAffineTablesOff);
#pragma unroll
    for (int c = 0; c < NCH; c++)
      copy_to_lds(s_tab[c], src + (size_t)c * kAffTabMax, st[c].n * 2u, threadIdx.x, kBlock);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* ws = s_ws[wv];
  const int groups_x = (p.bw + 7) >> 3, total = groups_x * p.bh;
  const int gwave = blockIdx.x * (kBlock / 64) + wv, nwaves = gridDim.x * (kBlock / 64);
  const int rr = lane >> 3, rb = lane & 7;
  const uint32_t map_w = (uint32_t)p.bw * 8;
  for (int t = gwave; t < total; t += nwaves) {
    const int by = t / groups_x, gx = t - by * groups_x;
    const int bx = gx * 8 + rb;
    const bool active = bx < p.bw;
    int comp[3][8];
    if (active) {
      const uint32_t y = by * 8 + rr, x0 = bx * 8;
      const float4* src = (const float4*)(p.ratio + ((size_t)y * map_w + x0) * NCH);
      float4 g[2 * NCH];
#pragma unroll
      for (int k = 0; k < 2 * NCH; k++) g[k] = src[k];
      uint32_t w[2 * NCH];  // the 8 * NCH map bytes of this row, packed as they come
      if (tabs) {
#pragma unroll
        for (int k = 0; k < 2 * NCH; k++) {
          const float f[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
          uint32_t acc = 0;
#pragma unroll
          for (int i = 0; i < 4; i++) acc |= step_code(f[i], s_tab[(4 * k + i) % NCH], st[(4 * k + i) % NCH]) << (8 * i);
          w[k] = acc;
        }
      } else {  // a channel without a table (gamma is 1 here, the launcher checks): the per-sample evaluation of generate_gainmap.hip,
                // as a rolled loop that parks its words in the wave's (idle) LDS workspace -- its float64 arithmetic must not
                // set the register count of the table path
        const float* gp = (const float*)src;
#pragma unroll 1
        for (int k = 0; k < 2 * NCH; k++) {
          uint32_t acc = 0;
#pragma unroll 1
          for (int i = 0; i < 4; i++) {
            const int c = (4 * k + i) % NCH;
            const float lg = gain_log2_of_ratio(gp[4 * k + i], p.math_tab);
            float m = div_by_rcp64(lg - p.dev->mn[c], p.dev->range_rcp[c]);
            m *= 255.0f;
            float t2 = m + 0.5f;
            t2 = (t2 < 0.0f) ? 0.0f : ((t2 > 255.0f) ? 255.0f : t2);
            acc |= (uint32_t)t2 << (8 * i);
          }
          ws[lane * 9 + k] = (int)acc;
        }
#pragma unroll
        for (int k = 0; k < 2 * NCH; k++) w[k] = (uint32_t)ws[lane * 9 + k];
      }
      if (p.map_out) {
        uint8_t* o = p.map_out + ((size_t)y * p.out_stride + x0) * NCH;
#pragma unroll
        for (int k = 0; k < NCH; k++) *(uint2*)(o + 8 * k) = uint2{w[2 * k], w[2 * k + 1]};
      }
      auto byte_at = [&](int e) -> int { return (int)((w[e >> 2] >> (8 * (e & 3))) & 0xffu); };
#pragma unroll
      for (int k = 0; k < 8; k++) {
        if (NCH == 3) {  // jccolor.c rgb_ycc_convert (fdct_quant.hip / jpeg_decode.hip: both published constant sets give these bytes)
          const int r = byte_at(3 * k), gg = byte_at(3 * k + 1), bb = byte_at(3 * k + 2);
          const int half = 1 << 15, off = 128 << 16;
          comp[0][k] = ((__mul24(FIX16(0.29900), r) + __mul24(FIX16(0.58700), gg) + __mul24(FIX16(0.11400), bb) + half) >> 16) - 128;
          comp[1][k] = ((__mul24(-FIX16(0.16874), r) + __mul24(-FIX16(0.33126), gg) + __mul24(FIX16(0.50000), bb) + off + half - 1) >> 16) - 128;
          comp[2][k] = ((__mul24(FIX16(0.50000), r) + __mul24(-FIX16(0.41869), gg) + __mul24(-FIX16(0.08131), bb) + off + half - 1) >> 16) - 128;
        } else {
          comp[0][k] = byte_at(k) - 128;
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < NCH; ci++) {
      int out[8], v[8];
      if (active) {
        fdct_1d<0>(comp[ci], out);
      } else {
#pragma unroll
        for (int c = 0; c < 8; c++) out[c] = 0;
      }
      column_pass_and_quantize(ws, lane, out, s_q[ci == 0 ? 0 : 1], v);
      if (active) store_coef_row(p.coef[ci] + ((size_t)by * p.bw + bx) * 64 + rr * 8, v);
    }
  }
}

// ---- the base image: convertYuv + FDCT of Y, Cb, Cr in one launch -------------------------------------------------------------------
struct BaseBlocksParams {
  const uint8_t* y;
  const uint8_t* u;
  const uint8_t* v;
  uint32_t sy, su, sv;     // strides in bytes
  int mcus_x, mcus_y;      // 16 x 16 MCUs (w / 16, h / 16)
  int convert;             // 0: the planes go to the FDCT as they are
  Mat3 c;                  // convertYuv's coefficients (host_tables.cpp: yuv_encoding_matrix)
  int16_t* coef[3];
  QuantPair q;
};

__device__ __forceinline__ uint32_t st8(float v) { return (uint32_t)__builtin_amdgcn_fmed3f(v, 0.0f, 255.0f); }  // static_cast<uint8_t>(CLIP3(v, 0, 255)), convert.hip

__global__ __launch_bounds__(kBlock) void base_blocks_kernel(const BaseBlocksParams p) {
  __shared__ int s_ws[kBlock / 64][8 * 8 * 9];
  // per wave: the converted samples of two MCUs -- luma 16 rows x 32, Cb and Cr 8 rows x 16 each
  __shared__ __attribute__((aligned(16))) uint8_t s_y[kBlock / 64][16 * 32];
  __shared__ __attribute__((aligned(16))) uint8_t s_c[kBlock / 64][2][8 * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* ws = s_ws[wv];
  uint8_t* ty = s_y[wv];
  uint8_t* tcb = s_c[wv][0];
  uint8_t* tcr = s_c[wv][1];
  const int pairs_x = (p.mcus_x + 1) >> 1, total = pairs_x * p.mcus_y;
  const int gwave = blockIdx.x * (kBlock / 64) + wv, nwaves = gridDim.x * (kBlock / 64);
  const int rr = lane >> 3, rb = lane & 7;
  __shared__ uint2 s_q[2][64];
  if (threadIdx.x < 128) s_q[threadIdx.x >> 6][threadIdx.x & 63] = uint2{p.q.qv[threadIdx.x >> 6][threadIdx.x & 63], p.q.qm[threadIdx.x >> 6][threadIdx.x & 63]};
  __syncthreads();
  const int bw_y = p.mcus_x * 2, bw_c = p.mcus_x;
  for (int t = gwave; t < total; t += nwaves) {
    const int my = t / pairs_x, mp = t - my * pairs_x;
    const int mx0 = mp * 2;
    const int n_mcu = (mx0 + 1 < p.mcus_x) ? 2 : 1;
    // ---- phase A: the pair's 128 quads (8 quad rows x 16 quad columns), two per lane, through convertYuv into the tile ----
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int q = lane + 64 * h, qr = q >> 4, qc = q & 15;  // quad row / column inside the pair
      if (qc < n_mcu * 8) {
        const size_t gy = (size_t)my * 16 + qr * 2, gx = (size_t)mx0 * 16 + qc * 2;
        const uint32_t r0 = *(const uint16_t*)(p.y + gy * p.sy + gx), r1 = *(const uint16_t*)(p.y + (gy + 1) * p.sy + gx);
        const uint32_t ub = p.u[(gy >> 1) * p.su + (gx >> 1)], vb = p.v[(gy >> 1) * p.sv + (gx >> 1)];
        uint32_t o0 = r0, o1 = r1, ou = ub, ov = vb;
        if (p.convert) {  // transformYuv420 (gainmapmath.cpp:686-748), the arithmetic of convert.hip: transform_yuv420_kernel
          const float u = (float)((int)ub - 128) * (1 / 255.0f), v = (float)((int)vb - 128) * (1 / 255.0f);
          const Color3 a = mat3_apply({(float)(r0 & 0xff) * (1 / 255.0f), u, v}, p.c);
          const Color3 b = mat3_apply({(float)(r0 >> 8) * (1 / 255.0f), u, v}, p.c);
          const Color3 c = mat3_apply({(float)(r1 & 0xff) * (1 / 255.0f), u, v}, p.c);
          const Color3 d = mat3_apply({(float)(r1 >> 8) * (1 / 255.0f), u, v}, p.c);
          const float nu = (((a.g + b.g) + c.g) + d.g) / 4.0f;
          const float nv = (((a.b + b.b) + c.b) + d.b) / 4.0f;
          o0 = st8(a.r * 255.0f + 0.5f) | (st8(b.r * 255.0f + 0.5f) << 8);
          o1 = st8(c.r * 255.0f + 0.5f) | (st8(d.r * 255.0f + 0.5f) << 8);
          ou = st8(nu * 255.0f + 128.0f + 0.5f);
          ov = st8(nv * 255.0f + 128.0f + 0.5f);
        }
        *(uint16_t*)(ty + (qr * 2) * 32 + qc * 2) = (uint16_t)o0;
        *(uint16_t*)(ty + (qr * 2 + 1) * 32 + qc * 2) = (uint16_t)o1;
        tcb[qr * 16 + qc] = (uint8_t)ou;
        tcr[qr * 16 + qc] = (uint8_t)ov;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- phase B: the 8 luma blocks.  Block rb = (MCU m, block row byy, block column bxx) ------------------------------------
    {
      const int m = rb >> 2, byy = (rb >> 1) & 1, bxx = rb & 1;
      const bool active = m < n_mcu;
      int in[8], out[8], v[8];
      if (active) {
        const uint8_t* src = ty + (byy * 8 + rr) * 32 + m * 16 + bxx * 8;
        const uint32_t lo = ((const uint32_t*)src)[0], hi = ((const uint32_t*)src)[1];
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
      column_pass_and_quantize(ws, lane, out, s_q[0], v);
      if (active) {
        const size_t bx = (size_t)(mx0 + m) * 2 + bxx, by = (size_t)my * 2 + byy;
        store_coef_row(p.coef[0] + (by * bw_y + bx) * 64 + rr * 8, v);
      }
    }
    // ---- phase C: the 4 chroma blocks (Cb of MCU 0, 1, Cr of MCU 0, 1): lanes with rb < 4 ------------------------------------
    {
      const int comp = (rb >> 1) & 1, m = rb & 1;
      const bool active = rb < 4 && m < n_mcu;
      int in[8], out[8], v[8];
      if (active) {
        const uint8_t* src = (comp ? tcr : tcb) + rr * 16 + m * 8;
        const uint32_t lo = ((const uint32_t*)src)[0], hi = ((const uint32_t*)src)[1];
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
      column_pass_and_quantize(ws, lane, out, s_q[1], v);
      if (active) store_coef_row(p.coef[1 + comp] + ((size_t)my * bw_c + (mx0 + m)) * 64 + rr * 8, v);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int resident_grid(int total_wave_items, int per_cu) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  int grid = (total_wave_items + kBlock / 64 - 1) / (kBlock / 64);
  if (grid > cus * per_cu) grid = cus * per_cu;
  return grid < 1 ? 1 : grid;
}

}  // namespace

// ratio plane (pass 1) + AffineDev (launch_minmax_table) -> quantized coefficients of the map's JPEG (+ the 8-bit map when map_out != null).
// nch 3: packed RGB semantics (rgb_ycc_convert, 4:4:4, luma table for Y, chroma table for Cb / Cr); nch 1: a Y400 map.
hipError_t launch_map_blocks(const float* ratio, const AffineDev* dev, const double* math_tab, int nch, int bw, int bh, const uint16_t* qt_luma,
                             const uint16_t* qt_chroma, int16_t* const coef[3], uint8_t* map_out, uint32_t out_stride, hipStream_t s) {
  MapBlocksParams p;
  memset(&p, 0, sizeof p);
  p.ratio = ratio; p.dev = dev; p.math_tab = math_tab; p.map_out = map_out; p.out_stride = out_stride; p.bw = bw; p.bh = bh;
  for (int i = 0; i < nch; i++) p.coef[i] = coef[i];
  fill_quant(qt_luma, qt_chroma, &p.q);
  int grid = (((bw + 7) / 8) * bh + kMapBlock / 64 - 1) / (kMapBlock / 64);
  const int cap = resident_grid(1 << 30, 3);  // 44 KB of LDS (tables + workspaces) per 512 threads: three workgroups per CU
  if (grid > cap) grid = cap;
  if (nch == 3) hipLaunchKernelGGL((map_blocks_kernel<3>), dim3(grid), dim3(kMapBlock), 0, s, p);
  else hipLaunchKernelGGL((map_blocks_kernel<1>), dim3(grid), dim3(kMapBlock), 0, s, p);
  return hipGetLastError();
}

// 4:2:0 planes (w, h multiples of 16) -> [convertYuv with matrix c ->] quantized coefficients of Y, Cb, Cr
hipError_t launch_base_blocks(const ImageView& yuv420, const Mat3* c, const uint16_t* qt_luma, const uint16_t* qt_chroma, int16_t* const coef[3],
                              hipStream_t s) {
  BaseBlocksParams p;
  memset(&p, 0, sizeof p);
  p.y = (const uint8_t*)yuv420.p[0]; p.u = (const uint8_t*)yuv420.p[1]; p.v = (const uint8_t*)yuv420.p[2];
  p.sy = yuv420.stride[0]; p.su = yuv420.stride[1]; p.sv = yuv420.stride[2];
  p.mcus_x = (int)(yuv420.w / 16); p.mcus_y = (int)(yuv420.h / 16);
  p.convert = c ? 1 : 0;
  if (c) p.c = *c;
  for (int i = 0; i < 3; i++) p.coef[i] = coef[i];
  fill_quant(qt_luma, qt_chroma, &p.q);
  const int grid = resident_grid(((p.mcus_x + 1) / 2) * p.mcus_y, 8);
  hipLaunchKernelGGL(base_blocks_kernel, dim3(grid), dim3(kBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace uhdr
