// Global -> LDS staging of tables by a whole workgroup.
// A plain "for (i = tid; i < n; i += nthreads) lds[i] = src[i]" loop compiles to ONE dword load, a wait for it, one LDS store per
// iteration (the compiler keeps the order of a load and the store that depends on it): a 128-thread workgroup that stages 48 KB of
// Huffman tables spent 96 dependent memory latencies -- tens of microseconds -- before its first symbol (round 6: found when a
// write pass without its stores was no faster).  Here a thread keeps UNROLL 16-byte loads in flight.
#ifndef UHDR_LDS_COPY_H
#define UHDR_LDS_COPY_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uhdr {

// nwords % 4 == 0; src and dst 16-byte aligned
template <int UNROLL = 8>
__device__ __forceinline__ void copy_words_to_lds(uint32_t* dst, const uint32_t* __restrict__ src, uint32_t nwords, uint32_t tid, uint32_t nthreads) {
  const uint4* s4 = (const uint4*)src;
  uint4* d4 = (uint4*)dst;
  const uint32_t nv = nwords >> 2;
  for (uint32_t i = tid; i < nv; i += UNROLL * nthreads) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t j = i + (uint32_t)u * nthreads;
      v[u] = s4[j < nv ? j : i];  // (a clamped index instead of a branch: the loads stay back to back)
    }
    // Unconditional stores (beyond the end: element i once more, the same value from the same thread).  With "if (j < nv) d4[j] = v[u]" the
    // compiler sinks every load into the branch of its store and waits for it there -- one latency per element again (seen in the ISA of
    // hyp_straggler_kernel, round 6; a compiler barrier between the two loops does not stop it).
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t j = i + (uint32_t)u * nthreads;
      d4[j < nv ? j : i] = v[u];
    }
  }
}

// Any word count, any alignment: 16-byte loads when both sides allow it, else dword loads -- UNROLL of them in flight either way.
template <int UNROLL = 8>
__device__ __forceinline__ void copy_to_lds(void* dst_, const void* __restrict__ src_, uint32_t nwords, uint32_t tid, uint32_t nthreads) {
  uint32_t* dst = (uint32_t*)dst_;
  const uint32_t* src = (const uint32_t*)src_;
  if ((((uintptr_t)src | (uintptr_t)dst) & 15u) == 0) {
    const uint32_t body = nwords & ~3u;
    copy_words_to_lds<UNROLL>(dst, src, body, tid, nthreads);
    if (body + tid < nwords) dst[body + tid] = src[body + tid];  // (at most three words)
    return;
  }
  for (uint32_t i = tid; i < nwords; i += UNROLL * nthreads) {
    uint32_t v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t j = i + (uint32_t)u * nthreads;
      v[u] = src[j < nwords ? j : i];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t j = i + (uint32_t)u * nthreads;
      dst[j < nwords ? j : i] = v[u];
    }
  }
}

}  // namespace uhdr
#endif
