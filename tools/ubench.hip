// Micro-benchmarks used to steer kernel design (run on the GPU box: tools/ubench).
//  1. VALU issue cost of the instruction kinds the apply kernel is made of (cycles per wave64
//     instruction per SIMD, with 8 waves per SIMD resident).
//  2. Store / load+store streaming patterns of the quad kernel vs a plain contiguous fill.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 2048;

#define OP_KERNEL(name, decl, body)                                           \
  __global__ __launch_bounds__(256) void name(float* out, float seed) {       \
    decl;                                                                     \
    for (int i = 0; i < ITERS; i++) {                                         \
      body body body body body body body body                                \
    }                                                                         \
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3;                  \
  }

#define DECL_F float r0 = seed + threadIdx.x, r1 = r0 * 1.1f, r2 = r0 * 1.2f, r3 = r0 * 1.3f
// 4 independent instructions per "body" -> 32 per loop iteration
OP_KERNEL(k_add_f32, DECL_F, asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)
OP_KERNEL(k_mul_f32, DECL_F, asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)
OP_KERNEL(k_fma_f32, DECL_F, asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)
OP_KERNEL(k_cvt_u32, DECL_F, asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
OP_KERNEL(k_cvt_ubyte, DECL_F, asm volatile("v_cvt_f32_ubyte0 %0, %0\n v_cvt_f32_ubyte1 %1, %1\n v_cvt_f32_ubyte0 %2, %2\n v_cvt_f32_ubyte1 %3, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
OP_KERNEL(k_lshl, DECL_F, asm volatile("v_lshlrev_b32 %0, 2, %0\n v_lshlrev_b32 %1, 2, %1\n v_lshlrev_b32 %2, 2, %2\n v_lshlrev_b32 %3, 2, %3" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
OP_KERNEL(k_add_u32, DECL_F, asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)
OP_KERNEL(k_med3, DECL_F, asm volatile("v_med3_f32 %0, %0, 0, %4\n v_med3_f32 %1, %1, 0, %4\n v_med3_f32 %2, %2, 0, %4\n v_med3_f32 %3, %3, 0, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)
OP_KERNEL(k_max_clamp, DECL_F, asm volatile("v_max_f32 %0, %0, %0 clamp\n v_max_f32 %1, %1, %1 clamp\n v_max_f32 %2, %2, %2 clamp\n v_max_f32 %3, %3, %3 clamp" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3));)
OP_KERNEL(k_lshl_or, DECL_F, asm volatile("v_lshl_or_b32 %0, %0, 16, %4\n v_lshl_or_b32 %1, %1, 16, %4\n v_lshl_or_b32 %2, %2, 16, %4\n v_lshl_or_b32 %3, %3, 16, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)
OP_KERNEL(k_mad_u24, DECL_F, asm volatile("v_mad_u32_u24 %0, %0, 4, %4\n v_mad_u32_u24 %1, %1, 4, %4\n v_mad_u32_u24 %2, %2, 4, %4\n v_mad_u32_u24 %3, %3, 4, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)
OP_KERNEL(k_pkrtz, DECL_F, asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %4\n v_cvt_pkrtz_f16_f32 %1, %1, %4\n v_cvt_pkrtz_f16_f32 %2, %2, %4\n v_cvt_pkrtz_f16_f32 %3, %3, %4" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(seed));)

typedef float f2 __attribute__((ext_vector_type(2)));
#define DECL_F2 f2 q0 = {seed + threadIdx.x, seed}, q1 = q0 * 1.1f, q2 = q0 * 1.2f, q3 = q0 * 1.3f; f2 sv = {seed, seed}; float r0, r1, r2, r3
#define FIN_F2 r0 = q0.x + q0.y; r1 = q1.x + q1.y; r2 = q2.x + q2.y; r3 = q3.x + q3.y;
#define OP_KERNEL2(name, body)                                                \
  __global__ __launch_bounds__(256) void name(float* out, float seed) {       \
    DECL_F2;                                                                  \
    for (int i = 0; i < ITERS; i++) {                                         \
      body body body body body body body body                                \
    }                                                                         \
    FIN_F2 out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3;           \
  }
OP_KERNEL2(k_pk_mul, asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(sv));)
OP_KERNEL2(k_pk_add, asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(sv));)
OP_KERNEL2(k_pk_fma, asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(sv));)

// LDS gather: random-ish vs broadcast-friendly indices
__global__ __launch_bounds__(256) void k_lds_gather(float* out, float seed, int spread) {
  __shared__ float tab[1024];
  for (int i = threadIdx.x; i < 1024; i += 256) tab[i] = i * seed;
  __syncthreads();
  uint32_t a = (threadIdx.x * 2654435761u) >> 8, b = a * 7 + 1, c = a * 13 + 5, d = a * 29 + 3;
  float r = 0;
  for (int i = 0; i < ITERS; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      float v0 = tab[(a >> 3) % spread], v1 = tab[(b >> 3) % spread], v2 = tab[(c >> 3) % spread], v3 = tab[(d >> 3) % spread];
      r += v0 + v1 + v2 + v3;
      a = a * 1664525u + 1013904223u; b += a; c ^= b; d += c;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

// ---- memory patterns -------------------------------------------------------------------------
// contiguous fill: each thread 16 B, grid-stride
__global__ __launch_bounds__(256) void k_fill(uint4* dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = make_uint4(i, 1, 2, 3);
}
// quad-kernel store pattern: wave writes 1 KiB in row y and 1 KiB in row y+1 (pitch = w*8)
__global__ __launch_bounds__(256) void k_quad_store(uint8_t* dst, uint32_t w, uint32_t h, uint32_t iters) {
  const uint32_t lane = threadIdx.x & 63, strips = w / 128, total = strips * (h / 2);
  const uint32_t nw = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t pitch = w * 8;
  for (uint32_t i = 0; i < iters; i++) {
    uint32_t t = min(w0 + i * nw, total - 1);
    uint32_t qy = t / strips, sx = t - qy * strips;
    uint32_t x = (sx * 64 + lane) * 2;
    uint4 v = make_uint4(t, lane, i, 7);
    *(uint4*)(dst + (size_t)(2 * qy) * pitch + x * 8) = v;
    *(uint4*)(dst + (size_t)(2 * qy + 1) * pitch + x * 8) = v;
  }
}
// same, but a wave covers 4 rows x 64 px... (1 KiB per row needs 128 px; variant: 4 rows per iteration)
__global__ __launch_bounds__(256) void k_quad_store4(uint8_t* dst, uint32_t w, uint32_t h, uint32_t iters) {
  const uint32_t lane = threadIdx.x & 63, strips = w / 128, total = strips * (h / 4);
  const uint32_t nw = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t pitch = w * 8;
  for (uint32_t i = 0; i < iters; i++) {
    uint32_t t = min(w0 + i * nw, total - 1);
    uint32_t qy = t / strips, sx = t - qy * strips;
    uint32_t x = (sx * 64 + lane) * 2;
    uint4 v = make_uint4(t, lane, i, 7);
#pragma unroll
    for (int r = 0; r < 4; r++) *(uint4*)(dst + (size_t)(4 * qy + r) * pitch + x * 8) = v;
  }
}
// quad-kernel load+store pattern without compute
__global__ __launch_bounds__(256) void k_quad_copy(const uint8_t* yp, const uint8_t* up, const uint8_t* vp, uint8_t* dst, uint32_t w, uint32_t h, uint32_t iters) {
  const uint32_t lane = threadIdx.x & 63, strips = w / 128, total = strips * (h / 2);
  const uint32_t nw = gridDim.x * 4, w0 = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t pitch = w * 8;
  for (uint32_t i = 0; i < iters; i++) {
    uint32_t t = min(w0 + i * nw, total - 1);
    uint32_t qy = t / strips, sx = t - qy * strips;
    uint32_t x = (sx * 64 + lane) * 2;
    uint32_t a = *(const uint16_t*)(yp + (size_t)(2 * qy) * w + x), b = *(const uint16_t*)(yp + (size_t)(2 * qy + 1) * w + x);
    uint32_t u = up[(size_t)qy * (w / 2) + x / 2], v = vp[(size_t)qy * (w / 2) + x / 2];
    *(uint4*)(dst + (size_t)(2 * qy) * pitch + x * 8) = make_uint4(a, u, v, 7);
    *(uint4*)(dst + (size_t)(2 * qy + 1) * pitch + x * 8) = make_uint4(b, u, v, 7);
  }
}

template <typename F>
float time_ms(F f, int reps = 5) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  float* out; CK(hipMalloc(&out, 2048 * 256 * 4));
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const double clk = pr.clockRate * 1e3;  // Hz
  printf("device %s CUs %d clock %.0f MHz\n", pr.gcnArchName, pr.multiProcessorCount, clk / 1e6);
  const double simds = pr.multiProcessorCount * 4.0;
#define RUN(k, n_per_iter) { float ms = time_ms([&] { hipLaunchKernelGGL(k, dim3(2048), dim3(256), 0, 0, out, 1.0f); });          \
    double winst = 2048.0 * 4 * ITERS * n_per_iter; /* wave-instructions */                                                     \
    printf("%-14s %8.3f ms  %.2f cycles per wave-instr per SIMD (at %.0f MHz nominal)\n", #k, ms, ms * 1e-3 * clk / (winst / simds), clk / 1e6); }
  RUN(k_add_f32, 32) RUN(k_mul_f32, 32) RUN(k_fma_f32, 32) RUN(k_cvt_u32, 32) RUN(k_cvt_ubyte, 32) RUN(k_lshl, 32) RUN(k_add_u32, 32)
  RUN(k_med3, 32) RUN(k_max_clamp, 32) RUN(k_lshl_or, 32) RUN(k_mad_u24, 32) RUN(k_pkrtz, 32) RUN(k_pk_mul, 32) RUN(k_pk_add, 32) RUN(k_pk_fma, 32)
  for (int spread : {1, 16, 64, 1024}) {
    float ms = time_ms([&] { hipLaunchKernelGGL(k_lds_gather, dim3(2048), dim3(256), 0, 0, out, 1.0f, spread); });
    double winst = 2048.0 * 4 * ITERS * 32;
    printf("lds_gather spread=%-5d %8.3f ms  %.2f cycles per ds_read_b32 per CU\n", spread, ms, ms * 1e-3 * clk / (winst / pr.multiProcessorCount));
  }
  // memory
  const uint32_t w = 7680, h = 4320;
  const size_t dbytes = (size_t)w * h * 8;
  uint8_t *dst, *dst2, *yp, *up, *vp;
  CK(hipMalloc(&dst, dbytes)); CK(hipMalloc(&dst2, dbytes)); CK(hipMalloc(&yp, (size_t)w * h)); CK(hipMalloc(&up, (size_t)w * h / 4)); CK(hipMalloc(&vp, (size_t)w * h / 4));
  CK(hipMemset(yp, 1, (size_t)w * h)); CK(hipMemset(up, 2, (size_t)w * h / 4)); CK(hipMemset(vp, 3, (size_t)w * h / 4));
  int k = 0;
  for (int grid : {1024, 2048, 4096}) {
    float ms = time_ms([&] { hipLaunchKernelGGL(k_fill, dim3(grid), dim3(256), 0, 0, (uint4*)((k++ & 1) ? dst : dst2), dbytes / 16); });
    printf("fill contiguous grid=%d: %.1f us  %.0f GB/s\n", grid, ms * 1e3, dbytes / (ms * 1e-3) / 1e9);
  }
  for (int grid : {1024, 1792, 2048}) {
    uint32_t total = (w / 128) * (h / 2), iters = (total + grid * 4 - 1) / (grid * 4);
    float ms = time_ms([&] { hipLaunchKernelGGL(k_quad_store, dim3(grid), dim3(256), 0, 0, (k++ & 1) ? dst : dst2, w, h, iters); });
    printf("quad store pattern grid=%d iters=%u: %.1f us  %.0f GB/s\n", grid, iters, ms * 1e3, dbytes / (ms * 1e-3) / 1e9);
    uint32_t total4 = (w / 128) * (h / 4), iters4 = (total4 + grid * 4 - 1) / (grid * 4);
    ms = time_ms([&] { hipLaunchKernelGGL(k_quad_store4, dim3(grid), dim3(256), 0, 0, (k++ & 1) ? dst : dst2, w, h, iters4); });
    printf("quad store 4-row grid=%d iters=%u: %.1f us  %.0f GB/s\n", grid, iters4, ms * 1e3, dbytes / (ms * 1e-3) / 1e9);
    ms = time_ms([&] { hipLaunchKernelGGL(k_quad_copy, dim3(grid), dim3(256), 0, 0, yp, up, vp, (k++ & 1) ? dst : dst2, w, h, iters); });
    printf("quad load+store grid=%d: %.1f us  %.0f GB/s (algorithmic 9.5 B/px)\n", grid, ms * 1e3, 9.5 * w * h / (ms * 1e-3) / 1e9);
  }
  return 0;
}
