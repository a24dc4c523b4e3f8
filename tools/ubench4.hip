// Access-pattern study for applyGainMap (8K frame, map C = RGBA8888 full-resolution map, F16 out):
// same bytes, same wave -> (column strip, row group) ownership as apply_quad_kernel, trivial math.
//   P = pixels per lane per row (2 = the shipping 2x2 quad, 4 = a 4x2 tile: every load instruction of a
//   wave then covers whole 128-byte lines), NTL / NTS = nontemporal loads / stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template <typename T, int NT> __device__ __forceinline__ T ld(const void* p) {
  if constexpr (NT) return __builtin_nontemporal_load((const T*)p); else return *(const T*)p;
}
template <typename T, int NT> __device__ __forceinline__ void st(void* p, T v) {
  if constexpr (NT) __builtin_nontemporal_store(v, (T*)p); else *(T*)p = v;
}

template <int P, int NTL, int NTS, int STORE_INTERLEAVE>
__global__ __launch_bounds__(256) void k_pat(const uint8_t* __restrict__ y, const uint8_t* __restrict__ u, const uint8_t* __restrict__ v,
                                             const uint8_t* __restrict__ m, uint8_t* __restrict__ d, uint32_t w, uint32_t h, uint32_t groups) {
  const uint32_t lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t strips = w / (64 * P), qh = h / 2;
  if (wave >= strips * groups) return;
  const uint32_t g0 = wave / strips, sx = wave - g0 * strips;
  const uint32_t xc = (sx * 64 + lane) * P;
  for (uint32_t qy = g0; qy < qh; qy += groups) {
    const uint32_t row = qy * 2;
    uint32_t acc[2] = {0, 0};
    uint32_t mm[2][P];
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if constexpr (P == 2) acc[r] = ld<uint16_t, NTL>(y + (size_t)(row + r) * w + xc);
      else acc[r] = ld<uint32_t, NTL>(y + (size_t)(row + r) * w + xc);
      if constexpr (P == 2) { const u2 a = ld<u2, NTL>(m + ((size_t)(row + r) * w + xc) * 4); mm[r][0] = a.x; mm[r][1] = a.y; }
      else { const u4 a = ld<u4, NTL>(m + ((size_t)(row + r) * w + xc) * 4); mm[r][0] = a.x; mm[r][1] = a.y; mm[r][2] = a.z; mm[r][3] = a.w; }
    }
    uint32_t cu, cv;
    if constexpr (P == 2) { cu = ld<uint8_t, NTL>(u + (size_t)qy * (w / 2) + xc / 2); cv = ld<uint8_t, NTL>(v + (size_t)qy * (w / 2) + xc / 2); }
    else { cu = ld<uint16_t, NTL>(u + (size_t)qy * (w / 2) + xc / 2); cv = ld<uint16_t, NTL>(v + (size_t)qy * (w / 2) + xc / 2); }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      uint8_t* o = d + ((size_t)(row + r) * w + xc) * 8;
#pragma unroll
      for (int k = 0; k < P / 2; k++) {
        const u4 val = {acc[r] ^ cu, mm[r][2 * k] ^ cv, mm[r][2 * k + 1], acc[r] + k};
        // STORE_INTERLEAVE: lane writes its 16-byte pieces so that each wave store instruction covers a
        // contiguous 1 KiB (lane l writes piece k at (k * 64 + l) * 16 of the wave's row segment)
        if constexpr (STORE_INTERLEAVE && P == 4) {
          uint8_t* seg = d + ((size_t)(row + r) * w + sx * 64 * P) * 8;
          st<u4, NTS>(seg + ((size_t)k * 64 + lane) * 16, val);
        } else {
          st<u4, NTS>(o + k * 16, val);
        }
      }
    }
  }
}

// Q: two (or more) 2x2 quads per lane, 128 pixels apart: loads stay 2-pixel wide (as in the shipping
// kernel), but a wave owns a 128*Q-pixel strip and its store instructions of one row are back to back
template <int Q, int NTL, int NTS, int QMAJOR = 0>
__global__ __launch_bounds__(256) void k_multi(const uint8_t* __restrict__ y, const uint8_t* __restrict__ u, const uint8_t* __restrict__ v,
                                               const uint8_t* __restrict__ m, uint8_t* __restrict__ d, uint32_t w, uint32_t h, uint32_t groups) {
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
        yv[q][r] = ld<uint16_t, NTL>(y + (size_t)(row + r) * w + xc);
        mm[q][r] = ld<u2, NTL>(m + ((size_t)(row + r) * w + xc) * 4);
      }
      cu[q] = ld<uint8_t, NTL>(u + (size_t)qy * (w / 2) + xc / 2);
      cv[q] = ld<uint8_t, NTL>(v + (size_t)qy * (w / 2) + xc / 2);
    }
#pragma unroll
    for (int a = 0; a < 2 * Q; a++) {
      const int r = QMAJOR ? a % 2 : a / Q, q = QMAJOR ? a / 2 : a % Q;  // QMAJOR: quad 0 rows 0,1 then quad 1 rows 0,1
      const uint32_t xc = sx * 128 * Q + q * 128 + lane * 2;
      const u4 val = {yv[q][r] ^ cu[q], mm[q][r].x ^ cv[q], mm[q][r].y, yv[q][r] + q};
      st<u4, NTS>(d + ((size_t)(row + r) * w + xc) * 8, val);
    }
  }
}

template <typename F>
float time_us(F f, int reps = 8) {
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
#define RUN(P, NTL, NTS, IL, blocks_per_cu) { \
    const uint32_t strips = w / (64 * P); const uint32_t groups = (256u * blocks_per_cu * 4) / strips; int flip = 0; \
    const uint32_t grid = (strips * groups + 3) / 4; \
    float us = time_us([&] { const int s = (flip ^= 1); hipLaunchKernelGGL((k_pat<P, NTL, NTS, IL>), dim3(grid), dim3(256), 0, 0, y[s], u[s], v[s], m[s], d[s], w, h, groups); }); \
    printf("P=%d nt(l,s)=%d,%d store_interleave=%d blocks/CU=%d: %7.1f us  %6.0f GB/s (%.1f%% of 8 TB/s)\n", P, NTL, NTS, IL, blocks_per_cu, us, px * 13.5 / us / 1e3, px * 13.5 / us / 1e3 / 80.0); }
#define RUNQ(Q, NTL, NTS, blocks_per_cu) RUNQM(Q, NTL, NTS, 0, blocks_per_cu)
#define RUNQM(Q, NTL, NTS, QM, blocks_per_cu) { \
    const uint32_t strips = w / (128 * Q); const uint32_t groups = (256u * blocks_per_cu * 4) / strips; int flip = 0; \
    const uint32_t grid = (strips * groups + 3) / 4; \
    float us = time_us([&] { const int s = (flip ^= 1); hipLaunchKernelGGL((k_multi<Q, NTL, NTS, QM>), dim3(grid), dim3(256), 0, 0, y[s], u[s], v[s], m[s], d[s], w, h, groups); }); \
    printf("Q=%d quads/lane qmajor=%d nt(l,s)=%d,%d blocks/CU=%d: %7.1f us  %6.0f GB/s (%.1f%% of 8 TB/s)\n", Q, QM, NTL, NTS, blocks_per_cu, us, px * 13.5 / us / 1e3, px * 13.5 / us / 1e3 / 80.0); }
  for (int rep = 0; rep < 2; rep++) {
    RUNQM(2, 0, 1, 1, 8) RUNQ(1, 0, 1, 8) RUNQ(2, 0, 1, 8) RUNQ(2, 0, 1, 4) RUNQ(4, 0, 1, 4) RUNQ(4, 0, 1, 8) RUNQ(2, 1, 1, 8) RUNQ(3, 0, 1, 8)
    RUN(2, 0, 0, 0, 8) RUN(2, 0, 1, 0, 8) RUN(2, 1, 1, 0, 8)
    RUN(4, 0, 0, 0, 8) RUN(4, 0, 1, 0, 8) RUN(4, 1, 1, 0, 8) RUN(4, 1, 0, 0, 8)
    RUN(4, 0, 1, 1, 8) RUN(4, 1, 1, 1, 8)
    RUN(4, 0, 1, 0, 4) RUN(4, 1, 1, 0, 4) RUN(4, 1, 1, 1, 4)
  }
  return 0;
}
