// Round-3, second pattern study (tools/ubench6 continued): map A (Y400, scale 4) at 8K with COLD inputs.
// With two rotating buffer sets the 51 MB of map-A inputs per set stay in the 256 MB infinity cache and the loads cost nothing
// (pattern 47 us = the store stream alone); with six sets they come from HBM and the shipping kernel takes 77 us.  Is that the
// access pattern (narrow reads trickling into a saturated write stream), and does it go away when the reads are wide, or
// issued in bulk up front?
//   LM 0  the shipping loads (per quad: two 16-bit luma loads, two chroma bytes, four tap bytes), one work item ahead
//   LM 1  wide loads per row step, redistributed through the wave's LDS slice (one 8-byte load = both luma rows ...)
//   LM 3  bulk prefetch: a wave first loads ALL the inputs of its row steps (everything it reads during the launch, ~7 KB)
//         into its LDS slice with the wide loads, then computes and stores from LDS
//   LM 2  no loads
//   LM 4  touch-ahead: the wave first READS all the inputs of its row steps with the wide loads and throws them away (they
//         land in L2 / the infinity cache as one read burst at the start of the launch), then runs the shipping loop
//   LM 5  linear touch-ahead: the same idea, but the touch phase is a plain grid-wide sweep -- wave k reads the k-th
//         1/nwaves-th of every input plane with 16-byte loads (1 KiB contiguous per instruction), whoever needs it later
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
constexpr int kMaxIter = 8;

struct Pat {
  const uint8_t *y, *u, *v, *m;
  uint8_t* d;
  uint32_t w, h, groups, n_iter;
};

template <int LM>
__global__ __launch_bounds__(256) void k8(const Pat p) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][(LM == 3 ? kMaxIter : 1) * 1024];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t wave = blockIdx.x * 4 + wv;
  const uint32_t w = p.w, qh = p.h / 2, strips = w / 256, groups = p.groups, mw = w / 4, mh = p.h / 4;
  if (wave >= strips * groups) return;
  const uint32_t g0 = wave / strips, sx = wave - g0 * strips, x0 = sx * 256;
  uint8_t* st = s_stage[wv];
  const uint32_t half = lane >> 5, l32 = lane & 31;
  auto wide_fetch = [&](uint32_t qy, uint8_t* dst) {
    const uint32_t row = qy * 2;
    const u2 ly = *(const u2*)(p.y + (size_t)(row + half) * w + x0 + l32 * 8);
    const uint8_t* cp = half ? p.v : p.u;
    const uint32_t lc = *(const uint32_t*)(cp + (size_t)qy * (w / 2) + x0 / 2 + l32 * 4);
    const uint32_t yl = row / 4, yu = min(yl + 1, mh - 1);
    uint32_t lt = 0;
    if (l32 < 17) lt = *(const uint32_t*)(p.m + (size_t)(half ? yu : yl) * mw + x0 / 4 + l32 * 4);
    *(u2*)(dst + lane * 8) = ly;
    *(uint32_t*)(dst + 512 + lane * 4) = lc;
    *(uint32_t*)(dst + 768 + lane * 4) = lt;
  };
  if constexpr (LM == 3) {
    for (uint32_t i = 0; i < p.n_iter; i++) {
      const uint32_t qy = g0 + i * groups;
      if (qy < qh) wide_fetch(qy, st + i * 1024);
    }
  }
  uint32_t touch = 0;
  if constexpr (LM == 4) {
#pragma unroll 8
    for (uint32_t i = 0; i < p.n_iter; i++) {
      const uint32_t qy = min(g0 + i * groups, qh - 1), row = qy * 2;
      const u2 ly = *(const u2*)(p.y + (size_t)(row + half) * w + x0 + l32 * 8);
      const uint8_t* cp = half ? p.v : p.u;
      const uint32_t lc = *(const uint32_t*)(cp + (size_t)qy * (w / 2) + x0 / 2 + l32 * 4);
      const uint32_t yl = row / 4, yu = min(yl + 1, mh - 1);
      uint32_t lt = 0;
      if (l32 < 17) lt = *(const uint32_t*)(p.m + (size_t)(half ? yu : yl) * mw + x0 / 4 + l32 * 4);
      touch ^= ly.x ^ ly.y ^ lc ^ lt;
    }
    if (touch == 0x9e3779b9u) p.d[0] = 1;  // never true for the test data; keeps the loads
  }
  if constexpr (LM == 5) {
    const uint32_t nw = strips * groups;
    auto sweep = [&](const uint8_t* base, size_t bytes) {
      const size_t per = ((bytes / nw) + 1023) & ~(size_t)1023;  // bytes per wave, whole KiB
      u4 acc = {0, 0, 0, 0};
      for (size_t o = (size_t)wave * per; o < (size_t)(wave + 1) * per && o + lane * 16 + 16 <= bytes + 4096; o += 1024) {
        const u4 a = *(const u4*)(base + o + lane * 16);
        acc.x ^= a.x; acc.y ^= a.y; acc.z ^= a.z; acc.w ^= a.w;
      }
      return acc.x ^ acc.y ^ acc.z ^ acc.w;
    };
    touch = sweep(p.y, (size_t)w * p.h) ^ sweep(p.u, (size_t)w * p.h / 4) ^ sweep(p.v, (size_t)w * p.h / 4) ^ sweep(p.m, (size_t)mw * mh);
    if (touch == 0x9e3779b9u) p.d[0] = 1;
  }
  for (uint32_t i = 0; i < p.n_iter; i++) {
    const uint32_t qy = g0 + i * groups;
    if (qy >= qh) break;
    const uint32_t row = qy * 2;
    uint32_t yv[2][2], cu[2], cv[2], tp[2][4];
    if constexpr (LM == 0 || LM == 4 || LM == 5) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint32_t xc = x0 + q * 128 + lane * 2;
#pragma unroll
        for (int r = 0; r < 2; r++) yv[q][r] = *(const uint16_t*)(p.y + (size_t)(row + r) * w + xc);
        cu[q] = p.u[(size_t)qy * (w / 2) + xc / 2];
        cv[q] = p.v[(size_t)qy * (w / 2) + xc / 2];
        const uint32_t yl = row / 4, yu = min(yl + 1, mh - 1), xl = xc / 4, xu = min(xl + 1, mw - 1);
        tp[q][0] = p.m[(size_t)yl * mw + xl];
        tp[q][1] = p.m[(size_t)yu * mw + xl];
        tp[q][2] = p.m[(size_t)yl * mw + xu];
        tp[q][3] = p.m[(size_t)yu * mw + xu];
      }
    } else if constexpr (LM == 1 || LM == 3) {
      const uint8_t* s = st + (LM == 3 ? i * 1024 : 0);
      if constexpr (LM == 1) wide_fetch(qy, st);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        yv[q][0] = *(const uint16_t*)(s + q * 128 + lane * 2);
        yv[q][1] = *(const uint16_t*)(s + 256 + q * 128 + lane * 2);
        cu[q] = s[512 + q * 64 + lane];
        cv[q] = s[640 + q * 64 + lane];
        const uint32_t xl = q * 32 + lane / 2;
        tp[q][0] = s[768 + xl];
        tp[q][1] = s[896 + xl];
        tp[q][2] = s[768 + xl + 1];
        tp[q][3] = s[896 + xl + 1];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; q++) { yv[q][0] = yv[q][1] = lane + i; cu[q] = cv[q] = row; for (int k = 0; k < 4; k++) tp[q][k] = lane ^ k; }
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const int r = a / 2, q = a % 2;
      const uint32_t xc = x0 + q * 128 + lane * 2;
      const u4 val = {yv[q][r] ^ cu[q], tp[q][0] ^ cv[q], tp[q][1] + (tp[q][2] << 8), yv[q][r] + tp[q][3]};
      __builtin_nontemporal_store(val, (u4*)(p.d + ((size_t)(row + r) * w + xc) * 8));
    }
  }
}

template <typename F>
void time_us(F f, int n, int reps, float* out) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 12; i++) f();
  (void)hipDeviceSynchronize();
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
  const int N = 200, REPS = 3, NS = 6;
  uint8_t *y[NS], *u[NS], *v[NS], *m[NS], *d[NS];
  for (int s = 0; s < NS; s++) {
    CK(hipMalloc(&y[s], px + 4096)); CK(hipMalloc(&u[s], px / 4 + 4096)); CK(hipMalloc(&v[s], px / 4 + 4096)); CK(hipMalloc(&m[s], px / 16 + 4096)); CK(hipMalloc(&d[s], px * 8));
    CK(hipMemset(y[s], 1, px)); CK(hipMemset(u[s], 2, px / 4)); CK(hipMemset(v[s], 3, px / 4)); CK(hipMemset(m[s], 4, px / 16));
  }
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  // clock ramp: the part idles at 1.4 GHz and needs ~0.5 s of load to reach 2.4 GHz
  auto run = [&](auto kern, const char* name, int nsets, int bpc) {
    Pat p; memset(&p, 0, sizeof p);
    const uint32_t strips = w / 256, qh = h / 2;
    p.w = w; p.h = h;
    p.groups = ((uint32_t)cus * bpc * 4) / strips;
    p.n_iter = (qh + p.groups - 1) / p.groups;
    if (p.n_iter > (uint32_t)kMaxIter) { printf("%s: n_iter %u too large\n", name, p.n_iter); return; }
    const uint32_t grid = (strips * p.groups + 3) / 4;
    int k = 0;
    float t[REPS];
    time_us([&] { const int s = (k++) % nsets; p.y = y[s]; p.u = u[s]; p.v = v[s]; p.m = m[s]; p.d = d[s];
                  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, p); }, N, REPS, t);
    const double bytes = px * 9.5625;
    printf("%-28s sets=%d blocks/CU=%d n_iter=%u: %6.1f %6.1f %6.1f us  (%.3f of 8 TB/s)\n", name, nsets, bpc, p.n_iter, t[0], t[1], t[2], bytes / t[2] / 8e6);
    fflush(stdout);
  };
  for (int warm = 0; warm < 40; warm++) { Pat p; memset(&p, 0, sizeof p); p.w = w; p.h = h; p.groups = 273; p.n_iter = 8; p.y = y[0]; p.u = u[0]; p.v = v[0]; p.m = m[0]; p.d = d[0];
    for (int j = 0; j < 100; j++) hipLaunchKernelGGL(k8<2>, dim3(2048), dim3(256), 0, 0, p); (void)hipDeviceSynchronize(); }
  for (int rep = 0; rep < 2; rep++) {
    for (int nsets : {2, 6}) {
      run(k8<0>, "shipping loads", nsets, 8);
      run(k8<1>, "wide loads via LDS", nsets, 8);
      run(k8<4>, "touch-ahead + shipping loads", nsets, 8);
      run(k8<5>, "linear touch-ahead + shipping", nsets, 8);
      run(k8<2>, "stores only", nsets, 8);
    }
  }
  return 0;
}
