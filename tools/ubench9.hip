// Issue cost of the VALU / LDS instructions the encode kernels are made of, on gfx950 (round 4).
// Each case: an unrolled loop of one instruction over 8 independent register chains, 8 waves per SIMD on every SIMD of
// the part, time per wave-instruction per SIMD relative to v_fma_f32.  Also: the semantics of v_cvt_rpi_i32_f32 and the
// exactness of the refined reciprocal / quotient sequences (encode_fast.h), swept exhaustively.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench9 tools/ubench9.hip && tools/ubench9
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <cmath>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

constexpr int kIters = 2048;

#define CHAIN8(INS)                                                                                                                    \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                                                   \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])                          \
               : "v"(a), "v"(b)                                                                                                               \
               : "vcc", "s10", "s11");

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %8\n"
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %9\n"
#define I_MED3(k) "v_med3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_LOG(k) "v_log_f32 %" #k ", %" #k "\n"
#define I_EXP(k) "v_exp_f32 %" #k ", %" #k "\n"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define I_CVTI(k) "v_cvt_i32_f32 %" #k ", %" #k "\n"
#define I_CVTRPI(k) "v_cvt_rpi_i32_f32 %" #k ", %" #k "\n"
#define I_CVTF(k) "v_cvt_f32_i32 %" #k ", %" #k "\n"
#define I_CND(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define I_LSHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 7\n"
#define I_LSHLADD(k) "v_lshl_add_u32 %" #k ", %" #k ", 3, %8\n"
#define I_MAX3(k) "v_max3_f32 %" #k ", %" #k ", %8, %9\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %8\n"
#define I_DIVSCALE(k) "v_div_scale_f32 %" #k ", vcc, %" #k ", %8, %9\n"
#define I_DIVFIXUP(k) "v_div_fixup_f32 %" #k ", %" #k ", %8, %9\n"
#define I_CND64(k) "v_cndmask_b32_e64 %" #k ", %" #k ", %8, s[10:11]\n"
#define I_CMP(k) "v_cmp_gt_f32_e32 vcc, %" #k ", %8\n"
#define I_CMPCND(k) "v_cmp_gt_f32_e32 vcc, %" #k ", %8\nv_cndmask_b32_e32 %" #k ", %" #k ", %9, vcc\n"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %8\n"
#define I_SUB(k) "v_sub_f32 %" #k ", %" #k ", %8\n"
#define I_RNDNE(k) "v_rndne_f32 %" #k ", %" #k "\n"
#define I_CVTU(k) "v_cvt_u32_f32 %" #k ", %" #k "\n"
#define I_UBYTE(k) "v_cvt_f32_ubyte1 %" #k ", %" #k "\n"
#define I_PERM(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define I_LSHR(k) "v_lshrrev_b32 %" #k ", 3, %" #k "\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define I_OR3(k) "v_or3_b32 %" #k ", %" #k ", %8, %9\n"
#define I_MADU24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define I_CVTPKU8(k) "v_cvt_pk_u8_f32 %" #k ", %" #k ", 1, %8\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %8, %9\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %8\n"
#define I_SUBREV_SDWA(k) "v_add_f32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"

template <int OP>
__global__ __launch_bounds__(512) void k32(float* out, float a, float b) {
  float r[8];
  for (int i = 0; i < 8; i++) r[i] = a + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < kIters; it++) {
    if (OP == 0) CHAIN8(I_FMA)
    if (OP == 1) CHAIN8(I_MUL)
    if (OP == 2) CHAIN8(I_ADD)
    if (OP == 3) CHAIN8(I_MED3)
    if (OP == 4) CHAIN8(I_RCP)
    if (OP == 5) CHAIN8(I_LOG)
    if (OP == 6) CHAIN8(I_EXP)
    if (OP == 7) CHAIN8(I_CVTI)
    if (OP == 8) CHAIN8(I_CVTRPI)
    if (OP == 9) CHAIN8(I_CVTF)
    if (OP == 10) CHAIN8(I_CND)
    if (OP == 11) CHAIN8(I_LSHL)
    if (OP == 12) CHAIN8(I_AND)
    if (OP == 13) CHAIN8(I_BFE)
    if (OP == 14) CHAIN8(I_LSHLADD)
    if (OP == 15) CHAIN8(I_MAX3)
    if (OP == 16) CHAIN8(I_MULLO)
    if (OP == 17) CHAIN8(I_SQRT)
    if (OP == 18) CHAIN8(I_DIVSCALE)
    if (OP == 19) CHAIN8(I_DIVFIXUP)
    if (OP == 20) CHAIN8(I_CND64)
    if (OP == 21) CHAIN8(I_CMP)
    if (OP == 22) CHAIN8(I_CMPCND)
    if (OP == 23) CHAIN8(I_MAX)
    if (OP == 24) CHAIN8(I_SUB)
    if (OP == 25) CHAIN8(I_RNDNE)
    if (OP == 26) CHAIN8(I_CVTU)
    if (OP == 27) CHAIN8(I_UBYTE)
    if (OP == 28) CHAIN8(I_PERM)
    if (OP == 29) CHAIN8(I_LSHR)
    if (OP == 30) CHAIN8(I_ADDU)
    if (OP == 31) CHAIN8(I_OR3)
    if (OP == 32) CHAIN8(I_MADU24)
    if (OP == 33) CHAIN8(I_CVTPKU8)
    if (OP == 34) CHAIN8(I_FMAC)
    if (OP == 35) CHAIN8(I_MOV)
    if (OP == 36) CHAIN8(I_SUBREV_SDWA)
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += r[i];
  if (s == 12345.678f) out[0] = s;
}

typedef float f2 __attribute__((ext_vector_type(2)));
#define CHAIN8P(INS)                                                                                                                   \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                                                   \
               : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])                          \
               : "v"(a), "v"(b));
#define P_FMA(k) "v_pk_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define P_MUL(k) "v_pk_mul_f32 %" #k ", %" #k ", %8\n"
#define P_ADD(k) "v_pk_add_f32 %" #k ", %" #k ", %9\n"
#define D_FMA(k) "v_fma_f64 %" #k ", %" #k ", %8, %9\n"
#define D_MUL(k) "v_mul_f64 %" #k ", %" #k ", %8\n"
#define D_ADD(k) "v_add_f64 %" #k ", %" #k ", %9\n"
#define D_RCP(k) "v_rcp_f64 %" #k ", %" #k "\n"

template <int OP>
__global__ __launch_bounds__(512) void k64(float* out, float a_, float b_) {
  if (OP < 3) {
    f2 r[8];
    const f2 a = {a_, a_}, b = {b_, b_};
    for (int i = 0; i < 8; i++) r[i] = (f2){a_ + threadIdx.x * 1e-3f + i, a_ + i};
    for (int it = 0; it < kIters; it++) {
      if (OP == 0) CHAIN8P(P_FMA)
      if (OP == 1) CHAIN8P(P_MUL)
      if (OP == 2) CHAIN8P(P_ADD)
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += r[i].x + r[i].y;
    if (s == 12345.678f) out[0] = s;
  } else {
    double r[8];
    const double a = a_, b = b_;
    for (int i = 0; i < 8; i++) r[i] = a_ + threadIdx.x * 1e-3 + i;
    for (int it = 0; it < kIters; it++) {
      if (OP == 3) CHAIN8P(D_FMA)
      if (OP == 4) CHAIN8P(D_MUL)
      if (OP == 5) CHAIN8P(D_ADD)
      if (OP == 6) CHAIN8P(D_RCP)
    }
    double s = 0;
    for (int i = 0; i < 8; i++) s += r[i];
    if (s == 12345.678) out[0] = (float)s;
  }
}

// conversions between f32 and f64: chains alternate f32 -> f64 -> f32
__global__ __launch_bounds__(512) void kcvt(float* out, float a) {
  float r[8];
  for (int i = 0; i < 8; i++) r[i] = a + threadIdx.x * 1e-3f + i;
  for (int it = 0; it < kIters; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double d;
      asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(r[i]));
      asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(r[i]) : "v"(d));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += r[i];
  if (s == 12345.678f) out[0] = s;
}

// LDS gathers: random (bank-conflicting) vs linear addresses, b32 / b64 / b128
template <int W, bool RANDOM>
__global__ __launch_bounds__(512) void klds(float* out, uint32_t seed) {
  __shared__ uint32_t tab[8192];
  for (int i = threadIdx.x; i < 8192; i += 512) tab[i] = i * 2654435761u;
  __syncthreads();
  uint32_t addr[8];
  for (int i = 0; i < 8; i++) {
    uint32_t h = (threadIdx.x * 8 + i) * 2654435761u + seed;
    addr[i] = RANDOM ? ((h >> 8) % (8192 - 4)) : ((threadIdx.x & 63) * (W / 4) + i * 256) % (8192 - 4);
    addr[i] = (addr[i] * 4) & ~(W - 1);
  }
  uint32_t acc = 0;
  for (int it = 0; it < kIters / 4; it++) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      if (W == 4) {
        uint32_t v;
        asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr[i]));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc ^= v;
      } else if (W == 8) {
        uint2 v;
        asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr[i]));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc ^= v.x;
      } else {
        uint4 v;
        asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr[i]));
        asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
        acc ^= v.x;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// ---- exactness sweeps ------------------------------------------------------------------------------------------------
// (1) v_cvt_rpi_i32_f32(x) == (int)floor((double)x + 0.5) for every float in [0, 2^23)
__global__ void sweep_rpi(unsigned long long* bad, uint32_t* first) {
  const uint64_t n = 0x4B000000ull;  // bit patterns of [0, 2^23]
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i <= n; i += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((uint32_t)i);
    int got;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(got) : "v"(x));
    const int want = (int)floor((double)x + 0.5);
    if (got != want) {
      if (atomicAdd(bad, 1ull) == 0) *first = (uint32_t)i;
    }
  }
}
// (2) the refined reciprocal r1 = fma(fma(-b, r0, 1), r0, r0), r0 = v_rcp_f32(b): is it RN(1/b)?  every normal positive float
__global__ void sweep_rcp(unsigned long long* cnt /* [0] r1 != RN(1/b), [1] r0 != RN(1/b), [2] max ulp err of r0 */, uint32_t* first) {
  for (uint64_t i = 0x00800000ull + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < 0x7F000000ull; i += (uint64_t)gridDim.x * blockDim.x) {
    const float b = __uint_as_float((uint32_t)i);
    float r0;
    asm volatile("v_rcp_f32 %0, %1" : "=v"(r0) : "v"(b));
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    const float want = (float)(1.0 / (double)b);  // RN24(RN53(1/b)) == RN24(1/b): 1/b is never within 2^-53 of a float midpoint
    if (want < 1.1754944e-38f) continue;          // sub-normal reciprocal: out of the kernels' range
    if (r1 != want) {
      if (atomicAdd(&cnt[0], 1ull) == 0) *first = (uint32_t)i;
    }
    if (r0 != want) atomicAdd(&cnt[1], 1ull);
    const long long d = llabs((long long)__float_as_uint(r0) - (long long)__float_as_uint(want));
    atomicMax(&cnt[2], (unsigned long long)d);
  }
}
// (3) quotient sequences against the compiler's IEEE division on random pairs from the kernels' ranges
__device__ __forceinline__ uint32_t rng(uint32_t& s) {
  s ^= s << 13; s ^= s >> 17; s ^= s << 5;
  return s;
}
__global__ void sweep_div(unsigned long long* cnt /* [0] seq8 mismatches, [1] seq6 (one correction) mismatches, [2] pairs */, uint32_t* first, uint32_t seed,
                          int emin, int emax) {
  uint32_t s = seed ^ ((blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u) ^ 0x9E3779B9u;
  if (!s) s = 1;
  unsigned long long bad8 = 0, bad6 = 0;
  for (int it = 0; it < 4096; it++) {
    const uint32_t ma = rng(s) & 0x7fffffu, mb = rng(s) & 0x7fffffu;
    const uint32_t ea = 127 + emin + rng(s) % (uint32_t)(emax - emin + 1), eb = 127 + emin + rng(s) % (uint32_t)(emax - emin + 1);
    const float a = __uint_as_float((ea << 23) | ma), b = __uint_as_float((eb << 23) | mb);
    const float want = a / b;
    float r0;
    asm volatile("v_rcp_f32 %0, %1" : "=v"(r0) : "v"(b));
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r = __builtin_fmaf(e, r0, r0);
    float q = a * r;
    float e2 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e2, r, q);
    const float q6 = q;
    e2 = __builtin_fmaf(-b, q, a);
    q = __builtin_fmaf(e2, r, q);
    if (q != want) {
      bad8++;
      if (atomicAdd(&cnt[3], 1ull) == 0) { first[0] = __float_as_uint(a); first[1] = __float_as_uint(b); }
    }
    if (q6 != want) bad6++;
  }
  atomicAdd(&cnt[0], bad8);
  atomicAdd(&cnt[1], bad6);
  atomicAdd(&cnt[2], 4096ull);
}

template <typename F>
static float time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  int cus = 0;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  float* out;
  CK(hipMalloc(&out, 64));
  const int grid = cus * 4;  // 4 x 512 threads = 32 waves per CU = 8 per SIMD
  // clock ramp
  for (int i = 0; i < 400; i++) hipLaunchKernelGGL(k32<0>, dim3(grid), dim3(512), 0, 0, out, 1.0001f, 0.5f);
  CK(hipDeviceSynchronize());
  const double wave_instr_per_simd = 8.0 * kIters * 8;  // 8 waves per SIMD, 8 instructions per iteration
  const float base = time_ms([&] { hipLaunchKernelGGL(k32<0>, dim3(grid), dim3(512), 0, 0, out, 1.0001f, 0.5f); }, 5);
  printf("CUs %d; v_fma_f32: %.3f ms -> %.2f ns per wave-instruction per SIMD (x 2.4 GHz = %.2f cycles)\n", cus, base, base * 1e6 / wave_instr_per_simd,
         base * 1e6 / wave_instr_per_simd * 2.4);
#define T32(OP, NAME)                                                                                                        \
  {                                                                                                                          \
    const float t = time_ms([&] { hipLaunchKernelGGL(k32<OP>, dim3(grid), dim3(512), 0, 0, out, 1.0001f, 0.5f); }, 5);      \
    printf("  %-22s %.3f ms  = %.2f x v_fma_f32\n", NAME, t, t / base);                                                      \
  }
  T32(1, "v_mul_f32") T32(2, "v_add_f32") T32(3, "v_med3_f32") T32(4, "v_rcp_f32") T32(5, "v_log_f32") T32(6, "v_exp_f32") T32(17, "v_sqrt_f32")
  T32(7, "v_cvt_i32_f32") T32(8, "v_cvt_rpi_i32_f32") T32(9, "v_cvt_f32_i32") T32(10, "v_cndmask_b32") T32(11, "v_lshlrev_b32") T32(12, "v_and_b32")
  T32(13, "v_bfe_u32") T32(14, "v_lshl_add_u32") T32(15, "v_max3_f32") T32(16, "v_mul_lo_u32") T32(18, "v_div_scale_f32") T32(19, "v_div_fixup_f32")
  T32(20, "v_cndmask_b32_e64 sgpr") T32(21, "v_cmp_gt_f32 vcc") T32(22, "v_cmp+v_cndmask pair") T32(23, "v_max_f32") T32(24, "v_sub_f32") T32(25, "v_rndne_f32")
  T32(26, "v_cvt_u32_f32") T32(27, "v_cvt_f32_ubyte1") T32(28, "v_perm_b32") T32(29, "v_lshrrev_b32") T32(30, "v_add_u32") T32(31, "v_or3_b32")
  T32(32, "v_mad_u32_u24") T32(33, "v_cvt_pk_u8_f32") T32(34, "v_fmac_f32") T32(35, "v_mov_b32") T32(36, "v_add_f32_sdwa")
#define T64(OP, NAME)                                                                                                        \
  {                                                                                                                          \
    const float t = time_ms([&] { hipLaunchKernelGGL(k64<OP>, dim3(grid), dim3(512), 0, 0, out, 1.0001f, 0.5f); }, 5);      \
    printf("  %-22s %.3f ms  = %.2f x v_fma_f32\n", NAME, t, t / base);                                                      \
  }
  T64(0, "v_pk_fma_f32") T64(1, "v_pk_mul_f32") T64(2, "v_pk_add_f32") T64(3, "v_fma_f64") T64(4, "v_mul_f64") T64(5, "v_add_f64") T64(6, "v_rcp_f64")
  {
    const float t = time_ms([&] { hipLaunchKernelGGL(kcvt, dim3(grid), dim3(512), 0, 0, out, 1.0001f); }, 5);
    printf("  %-22s %.3f ms  = %.2f x v_fma_f32 per PAIR (f32->f64->f32)\n", "v_cvt_f64_f32+back", t, t / base);
  }
  const double lds_instr_per_simd = 8.0 * (kIters / 4) * 8;
#define TL(W, R, NAME)                                                                                                              \
  {                                                                                                                                 \
    const float t = time_ms([&] { hipLaunchKernelGGL((klds<W, R>), dim3(grid), dim3(512), 0, 0, out, 12345u); }, 5);                \
    printf("  %-22s %.3f ms  = %.2f ns per wave-instruction per SIMD (%.2f x v_fma_f32)\n", NAME, t, t * 1e6 / lds_instr_per_simd,  \
           (t / lds_instr_per_simd) / (base / wave_instr_per_simd));                                                                \
  }
  TL(4, false, "ds_read_b32 linear") TL(4, true, "ds_read_b32 random") TL(8, false, "ds_read_b64 linear") TL(8, true, "ds_read_b64 random")
  TL(16, false, "ds_read_b128 linear") TL(16, true, "ds_read_b128 random")

  unsigned long long* cnt;
  uint32_t* first;
  CK(hipMalloc(&cnt, 64));
  CK(hipMalloc(&first, 64));
  unsigned long long h[8];
  uint32_t hf[4];
  CK(hipMemset(cnt, 0, 64)); CK(hipMemset(first, 0, 64));
  hipLaunchKernelGGL(sweep_rpi, dim3(cus * 8), dim3(256), 0, 0, cnt, first);
  CK(hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost));
  printf("v_cvt_rpi_i32_f32 vs floor((double)x + 0.5), every float in [0, 2^23]: %llu mismatches (first bits 0x%08x)\n", h[0], hf[0]);
  CK(hipMemset(cnt, 0, 64)); CK(hipMemset(first, 0, 64));
  hipLaunchKernelGGL(sweep_rcp, dim3(cus * 8), dim3(256), 0, 0, cnt, first);
  CK(hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost));
  printf("refined reciprocal vs RN(1/b), every normal float: %llu mismatches (first b bits 0x%08x); raw v_rcp_f32: %llu not correctly rounded, max %llu ulp\n", h[0],
         hf[0], h[1], h[2]);
  for (int range = 0; range < 3; range++) {
    const int emin = range == 0 ? -30 : (range == 1 ? -60 : -120), emax = range == 0 ? 20 : (range == 1 ? 60 : 120);
    CK(hipMemset(cnt, 0, 64)); CK(hipMemset(first, 0, 64));
    for (int rep = 0; rep < 16; rep++) hipLaunchKernelGGL(sweep_div, dim3(cus * 8), dim3(256), 0, 0, cnt, first, 777u + rep * 7919u, emin, emax);
    CK(hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(hf, first, 16, hipMemcpyDeviceToHost));
    printf("quotient sequences vs a / b, exponents [%d, %d], %llu random pairs: two corrections %llu mismatches (first a 0x%08x b 0x%08x), one correction %llu\n", emin,
           emax, h[2], h[0], hf[0], hf[1], h[1]);
  }
  return 0;
}
