// HBM ceiling for the applyGainMap traffic mix: streams R bytes in and W bytes out per "pixel" with
// 16-byte accesses and no arithmetic, for several read:write ratios and grid sizes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each wave iteration: NL u4 loads and NS u4 stores per lane (contiguous 1 KiB per wave-access)
template <int NL, int NS, int NTL = 0, int NTS = 0>
__global__ __launch_bounds__(256) void k_mix(const u4* __restrict__ src, u4* __restrict__ dst, uint32_t iters_total) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
  u4 acc = {0, 0, 0, 0};
  for (uint32_t it = wave; it < iters_total; it += nw) {
    u4 v[NL > 0 ? NL : 1];
#pragma unroll
    for (int k = 0; k < NL; k++) {
      const u4* a = &src[((size_t)it * NL + k) * 64 + lane];
      if (NTL) v[k] = __builtin_nontemporal_load(a); else v[k] = *a;
    }
#pragma unroll
    for (int k = 0; k < NL; k++) { acc.x ^= v[k].x; acc.y += v[k].y; acc.z ^= v[k].z; acc.w += v[k].w; }
#pragma unroll
    for (int k = 0; k < NS; k++) {
      u4* a = &dst[((size_t)it * NS + k) * 64 + lane];
      if (NTS) __builtin_nontemporal_store(acc, a); else *a = acc;
    }
  }
  if (NS == 0 && acc.x == 0x12345678u) dst[wave * 64 + lane] = acc;
}

template <typename F>
float time_us(F f, int reps = 6) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const size_t cap = (size_t)1200 << 20;  // two sets of each, beyond the 256 MiB infinity cache
  u4 *src, *dst;
  CK(hipMalloc(&src, cap)); CK(hipMalloc(&dst, cap)); CK(hipMemset(src, 1, cap)); CK(hipMemset(dst, 0, cap));
  const double px = 7680.0 * 4320.0;
#define RUN(NL, NS, blocks) RUNX(NL, NS, 0, 0, blocks)
#define RUNX(NL, NS, NTL, NTS, blocks) { \
    const double total = px * 13.5; /* bytes of one 8K map-C decode */ \
    const uint32_t iters = (uint32_t)(total / ((NL + NS) * 1024.0)); \
    int flip = 0; \
    float us = time_us([&] { const size_t off = (flip ^= 1) ? 0 : cap / 2 / 16; \
      hipLaunchKernelGGL((k_mix<NL, NS, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, src + off, dst + off, iters); }); \
    printf("loads:stores %2d:%2d nt(l,s)=%d,%d blocks %5d: %7.1f us  %6.0f GB/s\n", NL, NS, NTL, NTS, blocks, us, (double)iters * (NL + NS) * 1024.0 / us / 1e3); }
  for (int blocks : {2048, 8192}) {
    RUN(16, 0, blocks) RUN(0, 16, blocks) RUN(8, 8, blocks) RUN(11, 16, blocks) RUN(2, 3, blocks)
    RUNX(16, 0, 1, 0, blocks) RUNX(0, 16, 0, 1, blocks) RUNX(11, 16, 0, 1, blocks) RUNX(11, 16, 1, 0, blocks) RUNX(11, 16, 1, 1, blocks) RUNX(2, 3, 1, 1, blocks)
  }
  return 0;
}
