// Store-flavour study for applyGainMap's access pattern (8K frame, map C, F16 out; same ownership as the shipping kernel:
// two 2x2 quads per lane 128 pixels apart, tools/ubench4.hip k_multi<Q = 2>): which cache-policy bits on the 16-byte
// output stores give the lowest time?  MODE 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc1 nt, 6 sc0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <int MODE> __device__ __forceinline__ void st16(uint8_t* base, uint32_t off, u4 v) {
  if constexpr (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(off), "v"(v), "s"(base) : "memory");
  if constexpr (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, %2 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
  if constexpr (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, %2 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
  if constexpr (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1" ::"v"(off), "v"(v), "s"(base) : "memory");
  if constexpr (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
  if constexpr (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, %2 sc1 nt" ::"v"(off), "v"(v), "s"(base) : "memory");
  if constexpr (MODE == 6) asm volatile("global_store_dwordx4 %0, %1, %2 sc0" ::"v"(off), "v"(v), "s"(base) : "memory");
}

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256) void k_multi(const uint8_t* __restrict__ y, const uint8_t* __restrict__ u, const uint8_t* __restrict__ v,
                                               const uint8_t* __restrict__ m, uint8_t* __restrict__ d, uint32_t w, uint32_t h, uint32_t groups) {
  constexpr int Q = 2;
  const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t strips = w / (128 * Q), qh = h / 2;
  if (wave >= strips * groups) return;
  const uint32_t g0 = wave / strips, sx = wave - g0 * strips;
  for (uint32_t qy = g0; qy < qh; qy += groups) {
    const uint32_t row = qy * 2;
    uint32_t yv[Q][2], cu[Q], cv[Q];
    u2 mm[Q][2];
#pragma unroll
    for (int q = 0; q < Q; q++) {
      const uint32_t xc = sx * 128 * Q + q * 128 + lane * 2;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        yv[q][r] = *(const uint16_t*)(y + (size_t)(row + r) * w + xc);
        mm[q][r] = *(const u2*)(m + ((size_t)(row + r) * w + xc) * 4);
      }
      cu[q] = u[(size_t)qy * (w / 2) + xc / 2];
      cv[q] = v[(size_t)qy * (w / 2) + xc / 2];
    }
#pragma unroll
    for (int a = 0; a < 2 * Q; a++) {
      const int r = a / Q, q = a % Q;
      const uint32_t xc = sx * 128 * Q + q * 128 + lane * 2;
      const u4 val = {yv[q][r] ^ cu[q], mm[q][r].x ^ cv[q], mm[q][r].y, yv[q][r] + q};
      st16<MODE>(d, (uint32_t)(((size_t)(row + r) * w + xc) * 8), val);
    }
  }
}

template <typename F>
float time_us(F f, int reps = 10) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); f(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const uint32_t w = 7680, h = 4320;
  const size_t px = (size_t)w * h;
  uint8_t *y[2], *u[2], *v[2], *m[2], *d[2];
  for (int s = 0; s < 2; s++) {
    CK(hipMalloc(&y[s], px)); CK(hipMalloc(&u[s], px / 4)); CK(hipMalloc(&v[s], px / 4)); CK(hipMalloc(&m[s], px * 4)); CK(hipMalloc(&d[s], px * 8));
    CK(hipMemset(y[s], 1, px)); CK(hipMemset(u[s], 2, px / 4)); CK(hipMemset(v[s], 3, px / 4)); CK(hipMemset(m[s], 4, px * 4));
  }
  const char* names[] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt", "sc0"};
#define RUN(MODE, bpc) { \
    const uint32_t strips = w / 256; const uint32_t groups = (256u * bpc * 4) / strips; int flip = 0; \
    const uint32_t grid = (strips * groups + 3) / 4; \
    float us = time_us([&] { const int s = (flip ^= 1); hipLaunchKernelGGL((k_multi<MODE, 0>), dim3(grid), dim3(256), 0, 0, y[s], u[s], v[s], m[s], d[s], w, h, groups); }); \
    printf("stores %-11s blocks/CU=%d: %7.1f us  %6.0f GB/s (%.1f%% of 8 TB/s)\n", names[MODE], bpc, us, px * 13.5 / us / 1e3, px * 13.5 / us / 1e3 / 80.0); }
  for (int rep = 0; rep < 3; rep++) {
    RUN(0, 8) RUN(1, 8) RUN(2, 8) RUN(3, 8) RUN(4, 8) RUN(5, 8) RUN(6, 8)
  }
  return 0;
}
