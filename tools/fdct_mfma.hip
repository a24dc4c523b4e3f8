// BASELINE north_star: "MFMA only if the 8x8 DCT is cast as a dense matmul and rocprof shows it wins over the LDS butterfly".
// This is that experiment (round 3): libjpeg's JDCT_ISLOW forward DCT + quantizer on the matrix cores,
// bit-exact, timed against the shipping butterfly kernel (libultrahdr_amd/csrc/fdct_quant.hip, included below) on the same
// planes.  tools/profile_fdct.sh runs both under rocprofv3; the verdict is in profiles/r03_fdct_mfma_vs_butterfly.txt.
//
// Each islow pass is an exact integer linear map followed by DESCALE (jfdctint.c: every product and sum is exact in int32):
//     pass 1 (rows)     t[r][u]   = (sum_x M[u][x] * d[r][x] + 2^10) >> 11
//     pass 2 (columns)  c[v][u]   = (sum_r M[v][r] * t[r][u] + 2^14) >> 15
// with ONE matrix M for both (the 13-bit LL&M constants summed per output; rows 0 and 4 are +-8192 so that the common shift
// reproduces "(t10 +- t11) << PASS1_BITS" and "DESCALE(t10 +- t11, PASS1_BITS)" exactly).  |M| <= 11363 needs two signed
// base-256 digits, the pass-2 inputs (|t| < 2^15) two more, samples (level shifted) one:
//     pass 1 = 2 digit products, pass 2 = 4, on v_mfma_i32_16x16x32_i8 (A 16 x 32, B 32 x 16, int32 accumulate).
// Mapping (K = 8 of the instruction's 32 is all an 8-point transform can use):
//     pass 1: A = the rows of two blocks (16 x 8 samples), B = [M_lo^T | M_hi^T] (8 x 16): one MFMA gives both digit products
//             of two blocks in separate column halves; t = (C[u] + (C[u + 8] << 8)) >> 11 with the rounding term in C's input
//     pass 2: A = [M_lo ; M_hi] (16 x 8), B = one digit of t for two blocks (8 x 16): two MFMAs (t_lo, t_hi) per block pair;
//             c = C_lo[v] + ((C_lo[v + 8] + C_hi[v]) << 8) + (C_hi[v + 8] << 16)
// Operand layouts change between the passes (C: four rows per lane, B: eight k per lane), through the wave's LDS slice.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "fdct_quant.hip"

using namespace uhdr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef short s2 __attribute__((ext_vector_type(2)));

struct MfmaArgs {
  long mdig[16];     // lane n < 16: the 8 bytes digit(n / 8) of M[n % 8][0..7]
  uint32_t qv[64];   // quantval << 3
  uint32_t qm[64];   // ceil(2^32 / qv)
};

__global__ __launch_bounds__(256) void fdct_quant_mfma_kernel(const uint8_t* __restrict__ plane, size_t stride, int bw, int bh, const MfmaArgs ma,
                                                              int16_t* __restrict__ coef) {
  __shared__ __attribute__((aligned(16))) uint2 s_rows[4][64];        // level-shifted sample rows: [blk * 8 + row]
  __shared__ __attribute__((aligned(16))) int16_t s_t[4][8 * 8 * 8];  // pass-1 output, transposed: [blk][u][r]
  __shared__ __attribute__((aligned(16))) int16_t s_o[4][8 * 64];     // quantized coefficients: [blk][v * 8 + u]
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint2* rows = s_rows[wv];
  int16_t* tb = s_t[wv];
  int16_t* ob = s_o[wv];
  const int groups_x = bw >> 3, total = groups_x * bh;  // bw is a multiple of 8 here
  const int gwave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
  const int rr = lane >> 3, rb = lane & 7;              // load / store role: (row, block)
  const int col = lane & 15, quad = lane >> 4;          // MFMA C role: column, row group
  const long mconst = lane < 16 ? ma.mdig[lane] : 0L;
  const int round1 = col < 8 ? 1024 : 0;
  // quantizer constants of this lane's pass-2 outputs: v = 4 * (quad & 1) + r, u = col & 7 (lanes 0..31 hold results)
  uint32_t qv[4], qm[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int v = 4 * (quad & 1) + r, u = col & 7;
    qv[r] = ma.qv[v * 8 + u];
    qm[r] = ma.qm[v * 8 + u];
  }
  for (int t = gwave; t < total; t += nwaves) {
    const int by = t / groups_x, gx = t - by * groups_x, bx = gx * 8 + rb;
    {
      const uint8_t* src = plane + (size_t)(by * 8 + rr) * stride + (size_t)bx * 8;
      uint2 d = *(const uint2*)src;
      d.x ^= 0x80808080u;  // sample - 128 as a signed byte
      d.y ^= 0x80808080u;
      rows[rb * 8 + rr] = d;
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- pass 1: two blocks per MFMA ---------------------------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < 4; p++) {
      long a = 0;
      if (lane < 16) {
        const uint2 d = rows[(2 * p + (lane >> 3)) * 8 + (lane & 7)];
        a = (long)(((unsigned long)d.y << 32) | d.x);
      }
      const v4i c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, mconst, (v4i){round1, round1, round1, round1}, 0, 0, 0);
      int tt[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int hi = __builtin_amdgcn_update_dpp(0, c[r], 0x108 /* row_shl:8: lane i reads lane i + 8 */, 0xf, 0xf, true);
        tt[r] = (c[r] + (hi << 8)) >> 11;
      }
      if (col < 8) {  // rows 4 * quad + r of the pair: block 2p + quad / 2, rows 4 * (quad & 1) + r; transposed store [blk][u][r]
        const uint2 w = make_uint2((uint32_t)(tt[0] & 0xffff) | ((uint32_t)tt[1] << 16), (uint32_t)(tt[2] & 0xffff) | ((uint32_t)tt[3] << 16));
        *(uint2*)(tb + ((2 * p + (quad >> 1)) * 8 + col) * 8 + 4 * (quad & 1)) = w;
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // ---- pass 2: two MFMAs per block pair ------------------------------------------------------------------------------
#pragma unroll
    for (int p = 0; p < 4; p++) {
      long blo = 0, bhi = 0;
      if (lane < 16) {  // column n = (block lane / 8, u = lane % 8): its eight t[r], split into balanced base-256 digits
        const uint4 q = *(const uint4*)(tb + ((2 * p + (lane >> 3)) * 8 + (lane & 7)) * 8);
        const uint32_t l0 = __builtin_amdgcn_perm(q.y, q.x, 0x06040200u), l1 = __builtin_amdgcn_perm(q.w, q.z, 0x06040200u);
        const s2 k = {128, 128};
        const uint32_t hx = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, q.x) + k), hy = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, q.y) + k);
        const uint32_t hz = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, q.z) + k), hw = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, q.w) + k);
        const uint32_t h0 = __builtin_amdgcn_perm(hy, hx, 0x07050301u), h1 = __builtin_amdgcn_perm(hw, hz, 0x07050301u);
        blo = (long)(((unsigned long)l1 << 32) | l0);
        bhi = (long)(((unsigned long)h1 << 32) | h0);
      }
      const v4i z = {0, 0, 0, 0};
      const v4i clo = __builtin_amdgcn_mfma_i32_16x16x32_i8(mconst, blo, z, 0, 0, 0);
      const v4i chi = __builtin_amdgcn_mfma_i32_16x16x32_i8(mconst, bhi, z, 0, 0, 0);
      int out[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int clo_up = __shfl_xor(clo[r], 32, 64), chi_up = __shfl_xor(chi[r], 32, 64);  // the M_hi rows live 32 lanes up
        const int acc = clo[r] + ((clo_up + chi[r]) << 8) + (chi_up << 16);
        const int v = (acc + 16384) >> 15;
        const int sgn = v >> 31;
        const uint32_t ab = (uint32_t)((v ^ sgn) - sgn) + (qv[r] >> 1);
        const uint32_t qq = __umulhi(ab, qm[r]);
        out[r] = (int)(qq ^ (uint32_t)sgn) - sgn;
      }
      if (lane < 32) {  // v = 4 * quad + r (quad 0 / 1), column (block col / 8, u = col % 8)
#pragma unroll
        for (int r = 0; r < 4; r++) ob[(2 * p + (col >> 3)) * 64 + (4 * quad + r) * 8 + (col & 7)] = (int16_t)out[r];
      }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    *(uint4*)(coef + ((size_t)by * bw + bx) * 64 + rr * 8) = *(const uint4*)(ob + rb * 64 + rr * 8);
    __builtin_amdgcn_wave_barrier();
  }
}

// the exact linear part of jfdctint.c's 1-D pass (pre-DESCALE), rows 0 / 4 scaled to share the other rows' shift
static void lin_1d(const long in[8], long out[8]) {
  long t0 = in[0] + in[7], t7 = in[0] - in[7], t1 = in[1] + in[6], t6 = in[1] - in[6];
  long t2 = in[2] + in[5], t5 = in[2] - in[5], t3 = in[3] + in[4], t4 = in[3] - in[4];
  const long t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  out[0] = (t10 + t11) * 8192;
  out[4] = (t10 - t11) * 8192;
  long z1 = (t12 + t13) * FIX_0_541196100;
  out[2] = z1 + t13 * FIX_0_765366865;
  out[6] = z1 + t12 * -FIX_1_847759065;
  z1 = t4 + t7;
  long z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const long z5 = (z3 + z4) * FIX_1_175875602;
  t4 *= FIX_0_298631336; t5 *= FIX_2_053119869; t6 *= FIX_3_072711026; t7 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
  z3 += z5; z4 += z5;
  out[7] = t4 + z1 + z3; out[5] = t5 + z2 + z4; out[3] = t6 + z2 + z3; out[1] = t7 + z1 + z4;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  uint16_t qt[64];
  for (int i = 0; i < 64; i++) qt[i] = (uint16_t)(1 + (i * 7) % 23);  // small divisors: every coefficient position is exercised
  MfmaArgs ma;
  for (int n = 0; n < 16; n++) {
    unsigned long w = 0;
    for (int x = 0; x < 8; x++) {
      long e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, o[8];
      e[x] = 1;
      lin_1d(e, o);
      const long m = o[n % 8], hi = (m + 128) >> 8, lo = m - hi * 256;
      const long d = n < 8 ? lo : hi;
      if (d < -128 || d > 127) { printf("digit out of range\n"); return 1; }
      w |= (unsigned long)(uint8_t)(int8_t)d << (8 * x);
    }
    ma.mdig[n] = (long)w;
  }
  for (int i = 0; i < 64; i++) {
    ma.qv[i] = (uint32_t)qt[i] << 3;
    ma.qm[i] = (uint32_t)((0x100000000ull + ma.qv[i] - 1) / ma.qv[i]);
  }
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int size = 0; size < 2; size++) {
    const int w = size ? 7680 : 3840, h = size ? 4320 : 2160, bw = w / 8, bh = h / 8;
    std::vector<uint8_t> img((size_t)w * h);
    srand(7 + size);
    for (size_t i = 0; i < img.size(); i++) {
      const int y = (int)(i / w), x = (int)(i % w);
      int v = 128 + (int)(110 * sinf(x / 37.f) * cosf(y / 23.f)) + (rand() % 41) - 20;
      if ((x / 64 + y / 64) % 7 == 0) v = rand() & 255;  // some white-noise tiles: the extreme coefficient magnitudes
      img[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
    uint8_t* dimg[2]; int16_t *c_ref, *c_mf;
    for (int s = 0; s < 2; s++) { CK(hipMalloc(&dimg[s], img.size())); CK(hipMemcpy(dimg[s], img.data(), img.size(), hipMemcpyHostToDevice)); }
    CK(hipMalloc(&c_ref, img.size() * 2)); CK(hipMalloc(&c_mf, img.size() * 2));
    CK(hipMemset(c_mf, 0x55, img.size() * 2));
    CK(launch_fdct_quant(dimg[0], w, bw, bh, qt, c_ref, st));
    const int grid = 256 * 8;
    hipLaunchKernelGGL(fdct_quant_mfma_kernel, dim3(grid), dim3(256), 0, st, dimg[0], (size_t)w, bw, bh, ma, c_mf);
    CK(hipStreamSynchronize(st));
    std::vector<int16_t> a(img.size()), b(img.size());
    CK(hipMemcpy(a.data(), c_ref, a.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), c_mf, b.size() * 2, hipMemcpyDeviceToHost));
    size_t bad = 0, first = 0;
    for (size_t i = 0; i < a.size(); i++) if (a[i] != b[i]) { if (!bad) first = i; bad++; }
    printf("%dx%d: MFMA coefficients vs butterfly: %zu of %zu differ%s\n", w, h, bad, a.size(), bad ? "" : " (bit-exact)");
    if (bad) printf("  first at %zu: butterfly %d mfma %d\n", first, a[first], b[first]);
    // timing: clock ramp first, then iters launches of each
    for (int k = 0; k < 2; k++) {
      auto launch = [&](int i) {
        if (k == 0) CK(launch_fdct_quant(dimg[i & 1], w, bw, bh, qt, c_ref, st));
        else hipLaunchKernelGGL(fdct_quant_mfma_kernel, dim3(grid), dim3(256), 0, st, dimg[i & 1], (size_t)w, bw, bh, ma, c_mf);
      };
      for (int i = 0; i < iters; i++) launch(i);
      CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int i = 0; i < iters; i++) launch(i);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1e3 / iters;
      printf("  %-10s %7.2f us per plane  (%.0f GB/s algorithmic: 1 B in + 2 B out per sample)\n", k == 0 ? "butterfly" : "mfma", us, 3.0 * w * h / us / 1e3);
    }
    for (int s = 0; s < 2; s++) CK(hipFree(dimg[s]));
    CK(hipFree(c_ref)); CK(hipFree(c_mf));
  }
  return 0;
}
