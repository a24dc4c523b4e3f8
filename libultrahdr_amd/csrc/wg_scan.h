// Prefix sums over a workgroup: a wave-level scan by lane shuffles, then the waves' totals through LDS -- two barriers, where the
// Hillis-Steele form over LDS that these kernels had through round 5 takes 2 * log2(threads) of them (sixteen for 256 threads, twenty for
// 1024; the single-workgroup scans between the entropy passes consist of little else).  Integer sums: the result is the same bit for bit.
#ifndef UHDR_WG_SCAN_H
#define UHDR_WG_SCAN_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uhdr {

template <typename T>
__device__ __forceinline__ T wave_incl_scan_t(T v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const T y = __shfl_up(v, d, 64);  // (every lane takes part: a lane that sits out cannot be read from)
    if (lane >= (uint32_t)d) v += y;
  }
  return v;
}

// v[c] <- inclusive prefix sum of v[c] over the NT threads of the workgroup (thread order), for C independent sums at once; total[c] <- the
// workgroup's sum.  s_w: C * (NT / 64) elements of LDS, free again on return.
template <int NT, int C, typename T>
__device__ __forceinline__ void wg_incl_scan(T (&v)[C], T* s_w, T (&total)[C]) {
  static_assert(NT % 64 == 0 && NT <= 1024, "whole waves");
  constexpr int NW = NT / 64;
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
  for (int c = 0; c < C; c++) {
    v[c] = wave_incl_scan_t(v[c], lane);
    if (lane == 63u) s_w[c * NW + wv] = v[c];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < C; c++) {
    T before = 0, all = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) {
      const T t = s_w[c * NW + k];
      before += (uint32_t)k < wv ? t : (T)0;
      all += t;
    }
    v[c] += before;
    total[c] = all;
  }
  __syncthreads();
}

}  // namespace uhdr
#endif
