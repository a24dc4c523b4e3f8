// The libjpeg JDCT_ISLOW inverse DCT as wave-level building blocks (jidctint.c: CONST_BITS 13, PASS1_BITS 2:
// column pass from the dequantized coefficients, row pass, descale by 2^18, +128, range-limit table indexed
// modulo 1024).  Shared by the stand-alone decode kernels (jpeg_decode.hip) and the applyGainMap variant that
// consumes the base image in coefficient form (apply_gainmap.hip).  See jpeg_decode.hip for the provenance notes.
#pragma once
#include "uhdr_types.h"

namespace uhdr {
namespace idct {

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

template <bool M24>
__device__ __forceinline__ int mulc(int a, int c) {
  if constexpr (M24) return __mul24(a, c);
  else return (int)((uint32_t)a * (uint32_t)c);
}
// jidctint.c jpeg_idct_islow, one 1-D pass.  PASS 0: columns, descale by 2^11 -> workspace values.  PASS 1: rows,
// descale by 2^18, then range_limit[(x + 128) & 1023] (jdmaster.c prepare_range_limit_table: identity on 0..255, 255
// up to 639, 0 from 640) -> the 8-bit samples.
// Two rewrites that leave every bit unchanged under libjpeg's wrap-around INT32 arithmetic (modular addition is
// associative, so a constant may be added to the even part's DC terms instead of to each of the eight outputs --
// libjpeg-turbo folds its rounding "fudge factor" the same way):
//   * the rounding constant 2^(sh-1) of DESCALE rides on tmp0 / tmp1: descale is a bare arithmetic shift;
//   * the row pass also carries 512 << 18, so that t = bits 18..27 of the sum = (x + 512) & 1023 = (v + 384) & 1023
//     for the table index v = (x + 128) & 1023, and range_limit[v] = clamp(t - 384, 0, 255)  (v <= 255 -> t - 384 = v;
//     256 <= v < 640 -> 256..639 -> 255; v >= 640 -> t wraps to 0..383 -> negative -> 0): bit-field extract,
//     subtract, median -- three instructions per sample instead of nine.
template <int PASS, bool M24>
__device__ __forceinline__ void idct_1d(const int in[8], int out[8]) {
  constexpr int sh = PASS == 0 ? 13 - 2 : 13 + 2 + 3;
  constexpr uint32_t fudge = (1u << (sh - 1)) + (PASS == 1 ? (512u << 18) : 0u);
  int z2 = in[2], z3 = in[6];
  int z1 = mulc<M24>(z2 + z3, FIX_0_541196100);
  int tmp2 = z1 + mulc<M24>(z3, -FIX_1_847759065);
  int tmp3 = z1 + mulc<M24>(z2, FIX_0_765366865);
  z2 = in[0]; z3 = in[4];
  int tmp0 = (int)(((uint32_t)(z2 + z3) << 13) + fudge);
  int tmp1 = (int)(((uint32_t)(z2 - z3) << 13) + fudge);
  const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int z4 = tmp1 + tmp3;
  const int z5 = mulc<M24>(z3 + z4, FIX_1_175875602);
  tmp0 = mulc<M24>(tmp0, FIX_0_298631336);
  tmp1 = mulc<M24>(tmp1, FIX_2_053119869);
  tmp2 = mulc<M24>(tmp2, FIX_3_072711026);
  tmp3 = mulc<M24>(tmp3, FIX_1_501321110);
  z1 = mulc<M24>(z1, -FIX_0_899976223);
  z2 = mulc<M24>(z2, -FIX_2_562915447);
  z3 = mulc<M24>(z3, -FIX_1_961570560);
  z4 = mulc<M24>(z4, -FIX_0_390180644);
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  const int sum[8] = {tmp10 + tmp3, tmp11 + tmp2, tmp12 + tmp1, tmp13 + tmp0, tmp13 - tmp0, tmp12 - tmp1, tmp11 - tmp2, tmp10 - tmp3};
#pragma unroll
  for (int k = 0; k < 8; k++) {
    if constexpr (PASS == 0) {
      out[k] = sum[k] >> sh;
    } else {
      const int t = (int)(((uint32_t)sum[k] >> 18) & 1023u) - 384;
      out[k] = min(max(t, 0), 255);
    }
  }
}

// Lane (row rr, block rb) fetches and dequantizes its coefficient row of block (by, bx); `big` collects the
// magnitude bits that rule out the 24-bit multiply path.
__device__ __forceinline__ void load_dequant_row(const int16_t* __restrict__ coef, int bw, int by, int bx, int rr,
                                                 const int q[8], int v[8], int& big, bool row_ok = true) {
  if (bx < bw && row_ok) {
    const uint4 raw = *(const uint4*)(coef + ((size_t)by * bw + bx) * 64 + rr * 8);
    const uint32_t w4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int c = 0; c < 4; c++) {
      v[2 * c] = __mul24((int)(int16_t)(w4[c] & 0xffff), q[2 * c]);  // |coef| < 2^15, q < 2^16: exact
      v[2 * c + 1] = __mul24((int)(int16_t)(w4[c] >> 16), q[2 * c + 1]);
    }
    // all |v| < 2^13?  (three-operand max / min: one instruction per sample)
    const int mx = max(max(max(v[0], v[1]), max(v[2], v[3])), max(max(v[4], v[5]), max(v[6], v[7])));
    const int mn = min(min(min(v[0], v[1]), min(v[2], v[3])), min(min(v[4], v[5]), min(v[6], v[7])));
    big |= (int)(mx > 8191) | (int)(mn < -8191);
  } else {
#pragma unroll
    for (int c = 0; c < 8; c++) v[c] = 0;
  }
}

// One wavefront, eight blocks: dequantized rows in (lane role (row rr, block rb)) -> that lane's eight
// range-limited output samples.  ws is the wave's private 8 x 8 x 9-word workspace; on return it may be reused.
__device__ __forceinline__ void idct_wave(int* ws, const int v[8], int big, int rr, int rb, uint32_t s[8]) {
  const int cb = rr, cc = rb;  // column-pass role: (block, column)
  const bool fast = __builtin_amdgcn_ballot_w64(big != 0) == 0;  // wave-uniform
#pragma unroll
  for (int c = 0; c < 8; c++) ws[rb * 72 + rr * 9 + c] = v[c];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes have landed
  int in[8], out[8];
#pragma unroll
  for (int r = 0; r < 8; r++) in[r] = ws[cb * 72 + r * 9 + cc];
  if (fast) idct_1d<0, true>(in, out); else idct_1d<0, false>(in, out);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int r = 0; r < 8; r++) ws[cb * 72 + r * 9 + cc] = out[r];
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
  for (int c = 0; c < 8; c++) in[c] = ws[rb * 72 + rr * 9 + c];
  if (fast) idct_1d<1, true>(in, out); else idct_1d<1, false>(in, out);
#pragma unroll
  for (int c = 0; c < 8; c++) s[c] = (uint32_t)out[c];
  __builtin_amdgcn_wave_barrier();
}

}  // namespace idct
}  // namespace uhdr
