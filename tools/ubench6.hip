// Round-3 access-pattern study for applyGainMap at 8K (tools/ubench4 continued).  Same bytes and the same
// wave -> (256-pixel column strip, row group) ownership as apply_quad_kernel, trivial arithmetic.
//   MAP   0 = map A (Y400, scale 4: four tap bytes per quad), 2 = map C (RGBA8888, scale 1: 8 bytes per pixel pair)
//   LM    0 = the shipping loads (per quad: two 16-bit luma loads, two chroma bytes, map bytes)
//         1 = WIDE loads staged through the wave's private LDS slice: one 8-byte load fetches both luma rows of the
//             strip (lanes 0-31 row 0, lanes 32-63 row 1), one 4-byte load both chroma rows (lanes 0-31 U, 32-63 V),
//             map A: one 4-byte load both tap rows; the lanes then pick their bytes out of LDS
//         2 = no loads at all (the store stream alone)
//   mapping (run time): 0 shipping (wave -> row group qy0 = wave / strips, rows qy0 + i * groups)
//                       1 contiguous rows per wave (rows qy0 * n .. qy0 * n + n - 1)
//                       2 XCD bands: workgroup b runs on XCD b % 8; XCD x owns the row groups [x * G / 8, (x + 1) * G / 8)
//   over  (run time): grid = over x the resident waves (row groups scale with it; trip count shrinks)
//   exact (run time): 1 = a wave stops at the last row instead of recomputing the clamped row
// Also: plain stream kernels (k_mix) for the on-box ceiling of the same read : write mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void st_nt(void* p, u4 v) { __builtin_nontemporal_store(v, (u4*)p); }

struct Pat {
  const uint8_t *y, *u, *v, *m;
  uint8_t* d;
  uint32_t w, h, groups, mapping, exact, n_iter, blocks_per_xcd;
};

template <int MAP, int LM>
__global__ __launch_bounds__(256) void k_pat(const Pat p) {
  __shared__ __attribute__((aligned(16))) uint8_t s_stage[4][1024];
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  uint32_t blk = blockIdx.x;
  if (p.mapping == 2) blk = (blk & 7) * p.blocks_per_xcd + (blk >> 3);  // XCD x gets a contiguous range of virtual blocks
  const uint32_t wave = blk * 4 + wv;
  const uint32_t w = p.w, qh = p.h / 2, strips = w / 256, groups = p.groups;
  if (wave >= strips * groups) return;
  const uint32_t g0 = wave / strips, sx = wave - g0 * strips;
  const uint32_t x0 = sx * 256;
  uint8_t* st = s_stage[wv];
  const uint32_t mw = MAP == 0 ? w / 4 : w;
  for (uint32_t i = 0; i < p.n_iter; i++) {
    uint32_t qy = p.mapping == 1 ? g0 * p.n_iter + i : g0 + i * groups;
    if (qy >= qh) { if (p.exact) break; qy = qh - 1; }
    const uint32_t row = qy * 2;
    uint32_t yv[2][2], cu[2], cv[2], tp[2][4];
    u2 mm[2][2];
    if constexpr (LM == 0) {
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const uint32_t xc = x0 + q * 128 + lane * 2;
#pragma unroll
        for (int r = 0; r < 2; r++) {
          yv[q][r] = *(const uint16_t*)(p.y + (size_t)(row + r) * w + xc);
          if constexpr (MAP == 2) mm[q][r] = *(const u2*)(p.m + ((size_t)(row + r) * w + xc) * 4);
        }
        cu[q] = p.u[(size_t)qy * (w / 2) + xc / 2];
        cv[q] = p.v[(size_t)qy * (w / 2) + xc / 2];
        if constexpr (MAP == 0) {
          const uint32_t yl = row / 4, yu = min(yl + 1, p.h / 4 - 1), xl = xc / 4, xu = min(xl + 1, mw - 1);
          tp[q][0] = p.m[(size_t)yl * mw + xl];
          tp[q][1] = p.m[(size_t)yu * mw + xl];
          tp[q][2] = p.m[(size_t)yl * mw + xu];
          tp[q][3] = p.m[(size_t)yu * mw + xu];
        }
      }
    } else if constexpr (LM == 1) {
      const uint32_t half = lane >> 5, l32 = lane & 31;
      const u2 ly = *(const u2*)(p.y + (size_t)(row + half) * w + x0 + l32 * 8);
      const uint8_t* cp = half ? p.v : p.u;
      const uint32_t lc = *(const uint32_t*)(cp + (size_t)qy * (w / 2) + x0 / 2 + l32 * 4);
      uint32_t lt = 0;
      if constexpr (MAP == 0) {
        const uint32_t yl = row / 4, yu = min(yl + 1, p.h / 4 - 1);
        if (l32 < 17) lt = *(const uint32_t*)(p.m + (size_t)(half ? yu : yl) * mw + x0 / 4 + l32 * 4);  // 65 bytes per row (the buffer has slack)
      } else {
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int r = 0; r < 2; r++) mm[q][r] = *(const u2*)(p.m + ((size_t)(row + r) * w + x0 + q * 128 + lane * 2) * 4);
      }
      *(u2*)(st + lane * 8) = ly;               // row 0 at 0..255, row 1 at 256..511
      *(uint32_t*)(st + 512 + lane * 4) = lc;   // U at 512..639, V at 640..767
      if constexpr (MAP == 0) *(uint32_t*)(st + 768 + lane * 4) = lt;  // taps: row yl at 768.., row yu at 896..
#pragma unroll
      for (int q = 0; q < 2; q++) {
        yv[q][0] = *(const uint16_t*)(st + q * 128 + lane * 2);
        yv[q][1] = *(const uint16_t*)(st + 256 + q * 128 + lane * 2);
        cu[q] = st[512 + q * 64 + lane];
        cv[q] = st[640 + q * 64 + lane];
        if constexpr (MAP == 0) {
          const uint32_t xl = q * 32 + lane / 2;
          tp[q][0] = st[768 + xl];
          tp[q][1] = st[896 + xl];
          tp[q][2] = st[768 + xl + 1];
          tp[q][3] = st[896 + xl + 1];
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < 2; q++) { yv[q][0] = yv[q][1] = lane + i; cu[q] = cv[q] = row; }
      if constexpr (MAP == 0) { for (int q = 0; q < 2; q++) for (int k = 0; k < 4; k++) tp[q][k] = lane ^ k; }
      else { for (int q = 0; q < 2; q++) for (int r = 0; r < 2; r++) mm[q][r] = (u2){lane, row}; }
    }
#pragma unroll
    for (int a = 0; a < 4; a++) {
      const int r = a / 2, q = a % 2;
      const uint32_t xc = x0 + q * 128 + lane * 2;
      u4 val;
      if constexpr (MAP == 0) val = (u4){yv[q][r] ^ cu[q], tp[q][0] ^ cv[q], tp[q][1] + (tp[q][2] << 8), yv[q][r] + tp[q][3]};
      else val = (u4){yv[q][r] ^ cu[q], mm[q][r].x ^ cv[q], mm[q][r].y, yv[q][r] + q};
      st_nt(p.d + ((size_t)(row + r) * w + xc) * 8, val);
    }
  }
}

// plain streams: NL 16-byte loads and NS 16-byte stores per lane per iteration, 1 KiB contiguous per wave access
template <int NL, int NS, int NTL, int NTS>
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
void time_us(F f, int n, int reps, float* best, float* med) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  f(); f(); (void)hipDeviceSynchronize();
  float t[16];
  for (int r = 0; r < reps; r++) {
    (void)hipEventRecord(a);
    for (int i = 0; i < n; i++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    t[r] = ms * 1e3f / n;
  }
  for (int i = 0; i < reps; i++) for (int j = i + 1; j < reps; j++) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
  *best = t[0]; *med = t[reps / 2];
}

int main(int argc, char** argv) {
  const uint32_t w = 7680, h = 4320;
  const size_t px = (size_t)w * h;
  const int N = getenv("UB_N") ? atoi(getenv("UB_N")) : 30, REPS = 5;
  uint8_t *y[2], *u[2], *v[2], *m[2], *d[2];
  for (int s = 0; s < 2; s++) {
    CK(hipMalloc(&y[s], px + 4096)); CK(hipMalloc(&u[s], px / 4 + 4096)); CK(hipMalloc(&v[s], px / 4 + 4096)); CK(hipMalloc(&m[s], px * 4 + 4096)); CK(hipMalloc(&d[s], px * 8));
    CK(hipMemset(y[s], 1, px)); CK(hipMemset(u[s], 2, px / 4)); CK(hipMemset(v[s], 3, px / 4)); CK(hipMemset(m[s], 4, px * 4));
  }
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  auto run = [&](auto kern, const char* name, int MAP, uint32_t mapping, uint32_t over, uint32_t exact) {
    const uint32_t strips = w / 256, qh = h / 2;
    Pat p; memset(&p, 0, sizeof p);
    p.w = w; p.h = h; p.mapping = mapping; p.exact = exact;
    uint32_t groups = ((uint32_t)cus * 8 * 4 * over) / strips;
    if (groups > qh) groups = qh;
    p.groups = groups;
    p.n_iter = (qh + groups - 1) / groups;
    const uint32_t grid = (strips * groups + 3) / 4;
    p.blocks_per_xcd = (grid + 7) / 8;
    const uint32_t launch_grid = mapping == 2 ? p.blocks_per_xcd * 8 : grid;
    int flip = 0;
    float best, med;
    time_us([&] { const int s = (flip ^= 1); p.y = y[s]; p.u = u[s]; p.v = v[s]; p.m = m[s]; p.d = d[s];
                  hipLaunchKernelGGL(kern, dim3(launch_grid), dim3(256), 0, 0, p); }, N, REPS, &best, &med);
    const double bytes = px * (MAP == 0 ? 9.5625 : 13.5);
    printf("%-26s map %c mapping=%u over=%u exact=%u groups=%4u n_iter=%2u: best %6.1f us  median %6.1f us  %5.0f GB/s (%.3f of 8 TB/s at median)\n", name,
           MAP == 0 ? 'A' : 'C', mapping, over, exact, groups, p.n_iter, best, med, bytes / med / 1e3, bytes / med / 8e6);
    fflush(stdout);
  };
  // ---- stream ceilings on this box --------------------------------------------------------------------------------
  const size_t cap = (size_t)1200 << 20;
  u4 *src, *dst;
  CK(hipMalloc(&src, cap)); CK(hipMalloc(&dst, cap)); CK(hipMemset(src, 1, cap)); CK(hipMemset(dst, 0, cap));
#define MIX(NL, NS, NTL, NTS, blocks) { \
    const double total = (double)px * 13.5; const uint32_t iters = (uint32_t)(total / ((NL + NS) * 1024.0)); int flip = 0; float best, med; \
    time_us([&] { const size_t off = (flip ^= 1) ? 0 : cap / 2 / 16; hipLaunchKernelGGL((k_mix<NL, NS, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, src + off, dst + off, iters); }, N, REPS, &best, &med); \
    printf("stream loads:stores %2d:%2d nt(l,s)=%d,%d blocks %5d: best %6.1f us  median %6.1f us  %5.0f GB/s (%.3f of 8 TB/s at median)\n", NL, NS, NTL, NTS, blocks, best, med, (double)iters * (NL + NS) * 1024.0 / med / 1e3, (double)iters * (NL + NS) * 1024.0 / med / 8e6); fflush(stdout); }
  for (int rep = 0; rep < 2; rep++) {
    MIX(1, 1, 0, 0, 2048) MIX(1, 1, 1, 1, 2048) MIX(1, 1, 1, 1, 8192) MIX(4, 4, 1, 1, 2048)
    MIX(2, 3, 1, 1, 2048) MIX(2, 3, 1, 1, 8192) MIX(2, 3, 0, 1, 2048) MIX(2, 3, 0, 1, 8192)
    MIX(4, 0, 1, 0, 2048) MIX(0, 4, 0, 1, 2048) MIX(0, 4, 0, 0, 2048) MIX(0, 4, 0, 1, 8192) MIX(0, 4, 0, 0, 8192)
    MIX(1, 4, 0, 1, 2048) MIX(1, 4, 1, 1, 8192)
  }
  // ---- the kernel's pattern -------------------------------------------------------------------------------------------
  for (int rep = 0; rep < 2; rep++) {
    for (int MAPI = 0; MAPI < 2; MAPI++) {
#define BOTH(LM, name, mapping, over, exact) { if (MAPI == 0) run(k_pat<0, LM>, name, 0, mapping, over, exact); else run(k_pat<2, LM>, name, 2, mapping, over, exact); }
      BOTH(0, "shipping loads", 0, 1, 0)
      BOTH(0, "shipping loads", 0, 1, 1)
      BOTH(0, "shipping loads", 0, 2, 1)
      BOTH(0, "shipping loads", 0, 4, 1)
      BOTH(0, "shipping loads", 1, 1, 1)
      BOTH(0, "shipping loads", 2, 1, 1)
      BOTH(1, "wide loads via LDS", 0, 1, 0)
      BOTH(1, "wide loads via LDS", 0, 1, 1)
      BOTH(1, "wide loads via LDS", 0, 2, 1)
      BOTH(1, "wide loads via LDS", 0, 4, 1)
      BOTH(1, "wide loads via LDS", 1, 1, 1)
      BOTH(1, "wide loads via LDS", 2, 1, 1)
      BOTH(2, "stores only", 0, 1, 1)
      BOTH(2, "stores only", 0, 4, 1)
      BOTH(2, "stores only", 1, 1, 1)
      BOTH(2, "stores only", 2, 1, 1)
    }
  }
  return 0;
}
