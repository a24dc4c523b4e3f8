// Round-3 factor study: what separates apply_quad_kernel (8K map C: 82 us first burst, 86 us sustained) from its bare access
// pattern (tools/ubench6: 77 us sustained)?  The pattern kernel of ubench6 (shipping loads, shipping mapping) plus, one
// factor at a time and all together:
//   DATA  random bytes in the input planes instead of memset constants (and therefore random output bytes)
//   VALU  K dependent packed FMAs per pixel-pair row on the loaded data (the real kernel: ~79 VALU per row)
//   LDS   G gathers per pixel-pair row from an 8 KB LDS table at data-dependent indices (the real kernel: 12)
//   PRO   a prologue that stages 19 KB of tables from global memory into LDS behind a barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct Pat {
  const uint8_t *y, *u, *v, *m;
  uint8_t* d;
  const float* tab;
  uint32_t w, h, groups, n_iter;
};

template <int K, int G, int PRO>
__global__ __launch_bounds__(256) void k_pat(const Pat p) {
  __shared__ float s_tab[(G || PRO) ? 4864 : 1];  // 19 KB
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t wave = blockIdx.x * 4 + wv;
  const uint32_t w = p.w, qh = p.h / 2, strips = w / 256, groups = p.groups;
  if constexpr (G || PRO) {
    for (uint32_t i = threadIdx.x; i < 4864; i += 256) s_tab[i] = p.tab[i];
    __syncthreads();
  }
  if (wave >= strips * groups) return;
  const uint32_t g0 = wave / strips, sx = wave - g0 * strips;
  const uint32_t x0 = sx * 256;
  for (uint32_t i = 0; i < p.n_iter; i++) {
    const uint32_t qy = g0 + i * groups;
    if (qy >= qh) break;
    const uint32_t row = qy * 2;
    uint32_t yv[2][2], cu[2], cv[2];
    u2 mm[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const uint32_t xc = x0 + q * 128 + lane * 2;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        yv[q][r] = *(const uint16_t*)(p.y + (size_t)(row + r) * w + xc);
        mm[q][r] = *(const u2*)(p.m + ((size_t)(row + r) * w + xc) * 4);
      }
      cu[q] = p.u[(size_t)qy * (w / 2) + xc / 2];
      cv[q] = p.v[(size_t)qy * (w / 2) + xc / 2];
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const int r = a / 2, q = a % 2;
      const uint32_t xc = x0 + q * 128 + lane * 2;
      u4 val = {yv[q][r] ^ cu[q], mm[q][r].x ^ cv[q], mm[q][r].y, yv[q][r] + q};
      if constexpr (G > 0) {
        uint32_t acc = 0;
#pragma unroll
        for (int g = 0; g < G; g++) {
          const uint32_t idx = ((g & 1 ? mm[q][r].y : mm[q][r].x) >> (8 * (g >> 1) & 31)) & 0x7ffu;  // 11-bit data-dependent index
          acc ^= __float_as_uint(s_tab[idx + (g & 3) * 512]);
        }
        val.w ^= acc;
      }
      if constexpr (K > 0) {
        f2 t = {__uint_as_float((val.x & 0x007fffffu) | 0x3f800000u), __uint_as_float((val.y & 0x007fffffu) | 0x3f800000u)};
        const f2 c1 = {1.0000001f, 0.9999999f}, c2 = {1e-7f, -1e-7f};
#pragma unroll
        for (int k = 0; k < K; k++) t = __builtin_elementwise_fma(t, c1, c2);
        val.z ^= __float_as_uint(t.x) ^ __float_as_uint(t.y);
      }
      __builtin_nontemporal_store(val, (u4*)(p.d + ((size_t)(row + r) * w + xc) * 8));
    }
  }
}

__global__ void k_fill(uint32_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t x = (uint32_t)i * 2654435761u ^ seed;
    x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12; x *= 0x297a2d39u; x ^= x >> 15;
    p[i] = x;
  }
}

template <typename F>
void time_us(F f, int n, int reps, float* out) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); f(); (void)hipDeviceSynchronize();
  for (int r = 0; r < reps; r++) {
    (void)hipEventRecord(a);
    for (int i = 0; i < n; i++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    out[r] = ms * 1e3f / n;
  }
}

int main() {
  const uint32_t w = 7680, h = 4320;
  const size_t px = (size_t)w * h;
  const int N = 30, REPS = 5;
  uint8_t *y[2], *u[2], *v[2], *m[2], *d[2];
  float* tab;
  CK(hipMalloc(&tab, 4864 * 4)); CK(hipMemset(tab, 0x3c, 4864 * 4));
  for (int s = 0; s < 2; s++) {
    CK(hipMalloc(&y[s], px + 4096)); CK(hipMalloc(&u[s], px / 4 + 4096)); CK(hipMalloc(&v[s], px / 4 + 4096)); CK(hipMalloc(&m[s], px * 4 + 4096)); CK(hipMalloc(&d[s], px * 8));
  }
  auto fill = [&](bool rnd) {
    for (int s = 0; s < 2; s++) {
      if (rnd) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)y[s], px / 4, 11u + s);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)u[s], px / 16, 22u + s);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)v[s], px / 16, 33u + s);
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (uint32_t*)m[s], px, 44u + s);
      } else {
        (void)hipMemset(y[s], 1, px); (void)hipMemset(u[s], 2, px / 4); (void)hipMemset(v[s], 3, px / 4); (void)hipMemset(m[s], 4, px * 4);
      }
    }
    (void)hipDeviceSynchronize();
  };
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  auto run = [&](auto kern, const char* name, const char* data) {
    Pat p; memset(&p, 0, sizeof p);
    const uint32_t strips = w / 256, qh = h / 2;
    p.w = w; p.h = h; p.tab = tab;
    p.groups = ((uint32_t)cus * 8 * 4) / strips;
    p.n_iter = (qh + p.groups - 1) / p.groups;
    const uint32_t grid = (strips * p.groups + 3) / 4;
    int flip = 0;
    float t[REPS];
    time_us([&] { const int s = (flip ^= 1); p.y = y[s]; p.u = u[s]; p.v = v[s]; p.m = m[s]; p.d = d[s];
                  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p); }, N, REPS, t);
    printf("%-34s data=%-6s us per launch, five regions of %d launches in order:", name, data, N);
    for (int r = 0; r < REPS; r++) printf(" %6.1f", t[r]);
    printf("\n");
    fflush(stdout);
  };
  for (int rnd = 0; rnd < 2; rnd++) {
    fill(rnd != 0);
    const char* dn = rnd ? "random" : "memset";
    run(k_pat<0, 0, 0>, "pattern", dn);
    run(k_pat<0, 0, 1>, "pattern + prologue", dn);
    run(k_pat<40, 0, 0>, "pattern + 40 pk_fma/row", dn);
    run(k_pat<80, 0, 0>, "pattern + 80 pk_fma/row", dn);
    run(k_pat<0, 12, 0>, "pattern + 12 LDS gathers/row", dn);
    run(k_pat<80, 12, 1>, "pattern + 80 fma + 12 LDS + prologue", dn);
    run(k_pat<120, 12, 1>, "pattern + 120 fma + 12 LDS + prologue", dn);
    run(k_pat<0, 0, 0>, "pattern (again)", dn);
  }
  return 0;
}
