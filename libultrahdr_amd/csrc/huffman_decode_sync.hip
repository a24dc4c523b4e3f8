// Baseline Huffman decoding of a scan WITHOUT restart markers on gfx950 -- every file the reference writes
// (/root/reference/lib/src/jpegencoderhelper.cpp:187-201 never sets restart_interval), i.e. what libjpeg's
// decode_mcu_huff (jdhuff.c) walks bit-serially behind JpegDecoderHelper::decompressImage
// (lib/src/jpegdecoderhelper.cpp:169-535).  One entropy-coded segment has no byte-aligned entry points, but Huffman
// codes re-synchronise: a decoder started at an arbitrary bit soon falls in step with the true one.  The classic
// self-synchronising parallel decode (Klein & Wiseman 2003; Weissenberger & Schmidt 2018 / 2021 for JPEG) in five steps,
// all on the device:
//   1. unstuff: drop the 0x00 after every 0xFF (count per 4 KiB chunk -> scan -> compact), so that positions are plain
//      bit indices;
//   2. the clean stream is cut into subsequences of S bits, one lane each.  A lane's decoder state is
//      (bit position, block within the MCU, zig-zag index).  Round 0: every lane starts at the first bit of its
//      subsequence in state (0, 0) -- right for lane 0 only -- decodes to the first symbol boundary at or beyond the end
//      of its subsequence and publishes that END state;
//   3. rounds 1..: lane i restarts from lane i-1's published end state; if its new end state equals the one it published
//      before, everything downstream of it is unchanged.  Only lanes whose predecessor changed run; a fixed point is
//      reached after (longest unsynchronised chain) rounds -- typically 2-4 -- and is, by induction from lane 0, the
//      true decode.  Each lane also counts the blocks it completed;
//   4. exclusive scan of the block counts = the index of the block a lane starts in; the lanes decode once more, now
//      storing coefficients (natural order, into zero-initialised JBLOCK arrays) and the DC DIFFERENCE of every block
//      in scan order;
//   5. the DC predictions (a running sum per component over the scan order, dummy blocks of edge MCUs included) are one
//      workgroup-wide scan per component.
// Malformed streams (undefined code, run past the end of a block) only count as errors in step 4, where every lane is
// on the true path; steps 2-3 treat them as ordinary (deterministic) state transitions of a decoder that is lost.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lds_copy.h"
#include "wg_scan.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kChunk = 4096;   // bytes per workgroup of the unstuff passes (256 threads x 16 bytes)
constexpr int kSyncBlock = 64; // one wave per workgroup: a wave's lanes decode neighbouring subsequences

// RSTn sequence check without knowing a marker's ordinal where it is found: sum over the markers of (n + 1) x weight(position),
// two independent weights; equal sums on both sides <=> equal numbers at every position, up to a 2^-64 coincidence
__device__ __forceinline__ uint32_t rst_weight1(uint32_t at) { return (at * 2654435761u + 0x9e3779b9u) | 1u; }
__device__ __forceinline__ uint32_t rst_weight2(uint32_t at) { return ((at ^ (at >> 15)) * 0x85ebca6bu + 0xc2b2ae35u) | 1u; }
__device__ __forceinline__ bool is_stuffed(const uint8_t* __restrict__ d, uint32_t i) { return i > 0 && d[i] == 0 && d[i - 1] == 0xffu; }
// RST: also both bytes of every RSTn marker (0xFF 0xD0..0xD7; inside entropy-coded data a 0xFF that is followed by anything
// but 0x00 is a marker, T.81 B.1.1.5).  The interval that follows starts at the marker's place in the clean stream.
__device__ __forceinline__ bool is_rst_first(const uint8_t* __restrict__ d, uint32_t i, uint32_t n) {
  return d[i] == 0xffu && i + 1 < n && (d[i + 1] & 0xf8u) == 0xd0u;
}
template <bool RST>
__device__ __forceinline__ bool is_dropped(const uint8_t* __restrict__ d, uint32_t i, uint32_t n) {
  if (is_stuffed(d, i)) return true;
  if (!RST) return false;
  return is_rst_first(d, i, n) || (i > 0 && (d[i] & 0xf8u) == 0xd0u && d[i - 1] == 0xffu);
}

// Round 5: 16 bytes per thread as ONE 16-byte load (+ the byte in front of it) and a 16-bit drop mask instead of sixteen byte
// loads with a neighbour look-up each.  dropmask16: bit k set = byte base + k is dropped.
template <bool RST>
__device__ __forceinline__ uint32_t dropmask16(const uint8_t* __restrict__ data, uint32_t base, uint32_t n) {
  uint32_t m = 0;
  if (base + 17u <= n && base > 0 && ((uintptr_t)(data + base) & 15u) == 0) {
    const uint4 v = *(const uint4*)(data + base);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t prev = data[base - 1];
    const uint32_t next = data[base + 16u];
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      const uint32_t cur = (w[k >> 2] >> (8u * (k & 3u))) & 0xffu;
      const uint32_t nx = k < 15 ? (w[(k + 1u) >> 2] >> (8u * ((k + 1u) & 3u))) & 0xffu : next;
      bool drop = cur == 0u && prev == 0xffu;                                        // a stuffed zero
      if (RST) drop = drop || (cur == 0xffu && (nx & 0xf8u) == 0xd0u) || ((cur & 0xf8u) == 0xd0u && prev == 0xffu);  // both bytes of an RSTn marker
      m |= drop ? (1u << k) : 0u;
      prev = cur;
    }
  } else {
    for (uint32_t k = 0; k < 16 && base + k < n; k++) m |= is_dropped<RST>(data, base + k, n) ? (1u << k) : 0u;
  }
  return m;
}
// Round 6, measured and dropped: folding the three single-workgroup steps of this file (this scan, the chain's tile-entry walk, the DC
// partials' scan) into the kernels that feed them by the "last workgroup to finish does it" pattern.  The agent-scope release every
// workgroup has to make before it takes its ticket (per-XCD L2s are not coherent: buffer_wbl2) costs far more than the 5 us launch it
// saves once a kernel has hundreds of workgroups: unstuff_count 6 -> 19 us, dc_partial2 7 -> 18 / 10 -> 36 us, chain tiles 5 -> 10 /
// 7 -> 21 us, the encoder's stuff_count (8100 workgroups) 5 -> 144 us (profiles/r06_last_workgroup_merge_no.txt).
template <bool RST>
__global__ __launch_bounds__(256) void unstuff_count_kernel(const uint8_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ counts, const HuffInitFill fill) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  // the decoder's buffers get their initial state here (grid-stride; nothing in front of this kernel reads them, everything behind it is
  // stream-ordered): three fill launches of 2-5 us each -- and the launch boundaries between them -- used to open every decode
  {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x, nt = gridDim.x * 256u;
    const uint4 z = make_uint4(0, 0, 0, 0), f = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    for (uint32_t v = t; v < fill.zero_vec; v += nt) fill.zero_ptr[v] = z;
    for (uint32_t v = t; v < fill.ff_vec; v += nt) fill.ff_ptr[v] = f;
  }
  __syncthreads();
  const uint32_t base = blockIdx.x * kChunk + threadIdx.x * 16;
  const uint32_t c = base < n ? (uint32_t)__builtin_popcount(dropmask16<RST>(data, base, n)) : 0u;
  if (c) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt;
}
// single workgroup: counts[nchunks] -> exclusive prefix sums in place, total -> *total
__global__ __launch_bounds__(1024) void sync_scan_kernel(uint32_t* __restrict__ counts, int n, uint32_t* __restrict__ total) {
  __shared__ uint32_t s_w[16];
  const int tid = (int)threadIdx.x;
  const int per = (n + 1023) / 1024, lo = min(tid * per, n), hi = min(lo + per, n);
  // eight independent loads at a time (clamped index, no branch around a load: the compiler would wait for each inside its branch)
  uint32_t sum = 0;
  for (int i0 = lo; i0 < hi; i0 += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = counts[min(i0 + j, hi - 1)];
#pragma unroll
    for (int j = 0; j < 8; j++) sum += (i0 + j < hi) ? v[j] : 0u;
  }
  uint32_t sc[1] = {sum}, all[1];
  wg_incl_scan<1024, 1>(sc, s_w, all);
  uint32_t run = sc[0] - sum;
  for (int i0 = lo; i0 < hi; i0 += 8) {
    uint32_t v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = counts[min(i0 + j, hi - 1)];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (i0 + j < hi) counts[i0 + j] = run;
      run += v[j];
    }
  }
  if (tid == 1023 && total) *total = all[0];
}
// Exclusive scan of n words over many workgroups (the single-workgroup kernel above walks its elements with a stride
// between lanes: one cache line per lane and instruction, all on one CU -- 80 us for 50 K elements).  Step 1: a workgroup
// scans its kScanTile consecutive elements (coalesced loads, eight per thread) and publishes their sum; step 2: every
// consumer adds the sums of the tiles before it (at most a few dozen): tile_offset below.
constexpr int kScanThreads = 256, kScanPer = 8, kScanTile = kScanThreads * kScanPer;
__global__ __launch_bounds__(kScanThreads) void scan_tiles_kernel(uint32_t* __restrict__ v, int n, uint32_t* __restrict__ tile_sum) {
  __shared__ uint32_t s_val[kScanTile];
  __shared__ uint32_t s_w[kScanThreads / 64];
  const int tid = (int)threadIdx.x, base = (int)blockIdx.x * kScanTile;
  uint32_t ld[kScanPer];
#pragma unroll
  for (int j = 0; j < kScanPer; j++) ld[j] = v[min(base + j * kScanThreads + tid, n - 1)];  // (all eight in flight; base < n for every workgroup)
#pragma unroll
  for (int j = 0; j < kScanPer; j++) s_val[j * kScanThreads + tid] = base + j * kScanThreads + tid < n ? ld[j] : 0u;
  __syncthreads();
  uint32_t x[kScanPer], sum = 0;
#pragma unroll
  for (int j = 0; j < kScanPer; j++) { x[j] = s_val[tid * kScanPer + j]; sum += x[j]; }
  uint32_t sc[1] = {sum}, all[1];
  wg_incl_scan<kScanThreads, 1>(sc, s_w, all);  // (its first barrier: every x[] has been read)
  uint32_t run = sc[0] - sum;
#pragma unroll
  for (int j = 0; j < kScanPer; j++) { s_val[tid * kScanPer + j] = run; run += x[j]; }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kScanPer; j++) {
    const int i = base + j * kScanThreads + tid;
    if (i < n) v[i] = s_val[j * kScanThreads + tid];
  }
  if (tid == kScanThreads - 1) tile_sum[blockIdx.x] = all[0];
}
// The block counts' scan is left per tile (scan_tiles_kernel: kScanTile subsequences each, tile sums in scan_tmp); a write-pass workgroup
// -- 64 to 256 subsequences, never across a tile boundary -- adds the sums of the tiles in front of its own (round 6: was a launch of
// its own, scan_add_kernel).  Wave-uniform: scalar loads.
__device__ __forceinline__ uint32_t tile_offset(const HuffSyncArgs& a) {
  const uint32_t tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x) / (uint32_t)kScanTile));
  uint32_t off = 0;
  for (uint32_t t = 0; t < tile; t++) off += a.scan_tmp[t];
  return off;
}
// RST: rst_map gets bit (clean byte index) set where an interval starts (zero-initialised by the caller); rst_partial[chunk]
// = {markers found, their two sequence sums} -- per chunk, because thousands of atomics on three global words are serialised
// (3239 markers: 88 us for this kernel instead of 15)
// chunk_base: the chunks' drop counts -- scanned (exclusive) by sync_scan_kernel when `total` is null; else RAW, and every workgroup adds up the
// counts in front of its chunk itself (round 6: one launch less at the head of every decode; up to kUnstuffSelfPrefix chunks -- the reads grow
// with the square of the count) and the last one leaves the stream's total in *total.
constexpr int kUnstuffSelfPrefix = 4096;
template <bool RST>
__global__ __launch_bounds__(256) void unstuff_compact_kernel(const uint8_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ chunk_base,
                                                              uint8_t* __restrict__ clean, uint32_t* __restrict__ rst_map, uint32_t* __restrict__ rst_partial,
                                                              uint32_t* __restrict__ total) {
  __shared__ uint32_t s_w[4];
  __shared__ uint32_t s_rst[3];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kChunk + tid * 16;
  if (RST && tid < 3) s_rst[tid] = 0;
  const uint32_t mask = base < n ? dropmask16<RST>(data, base, n) : 0u;
  const uint32_t c = (uint32_t)__builtin_popcount(mask);
  uint32_t before = 0;
  if (total) {
    uint32_t part[1] = {0}, all[1];
    for (uint32_t i = tid; i < blockIdx.x; i += 256u) part[0] += chunk_base[i];
    wg_incl_scan<256, 1>(part, s_w, all);
    before = all[0];
  } else {
    before = chunk_base[blockIdx.x];
  }
  uint32_t sc[1] = {c}, mine[1];
  wg_incl_scan<256, 1>(sc, s_w, mine);
  if (total && blockIdx.x == gridDim.x - 1 && tid == 0) *total = before + mine[0];
  uint32_t dropped = before + sc[0] - c;  // dropped bytes before this thread's first byte
  if (mask == 0 && base + 16u <= n && ((uintptr_t)(data + base) & 15u) == 0) {
    // nothing dropped in these 16 bytes (15 threads in 16 of a stuffed stream): four dword stores at the shifted, in general
    // unaligned, destination (global memory takes unaligned dword accesses) instead of sixteen byte stores
    const uint4 v = *(const uint4*)(data + base);
    uint8_t* d = clean + (base - dropped);
    __builtin_memcpy(d, &v.x, 4);
    __builtin_memcpy(d + 4, &v.y, 4);
    __builtin_memcpy(d + 8, &v.z, 4);
    __builtin_memcpy(d + 12, &v.w, 4);
  } else {
    for (uint32_t i = base; i < base + 16 && i < n; i++) {
      if (RST && is_rst_first(data, i, n)) {
        const uint32_t at = i - dropped;  // where the next interval's first byte lands
        atomicOr(rst_map + (at >> 5), 1u << (at & 31u));
        // the marker's number (RST0..RST7 in turn), folded into two position-weighted sums; the write pass forms the same
        // sums from the numbers the interval ends SHOULD have (rst_weight above)
        const uint32_t m = (data[i + 1] & 7u) + 1u;
        atomicAdd(&s_rst[0], 1u);
        atomicAdd(&s_rst[1], rst_weight1(at) * m);
        atomicAdd(&s_rst[2], rst_weight2(at) * m);
      }
      if ((mask >> (i - base)) & 1u) dropped++;
      else clean[i - dropped] = data[i];
    }
  }
  if (RST) {
    __syncthreads();
    if (tid < 3) rst_partial[blockIdx.x * 3 + tid] = s_rst[tid];
  }
}

// ---- the decoder --------------------------------------------------------------------------------------------------
// A wave stages the 64 subsequences of its lanes (+ 16 bytes of the next one) in LDS with coalesced 16-byte loads; every
// lane then reads ITS bytes from LDS, 32 bits at a time -- a decode step never waits for global memory.  A lane's chunk
// is padded by one word so that the lanes' simultaneous reads fall into different banks.
struct Staged {
  const uint32_t* w;   // the wave's LDS region, as words
  uint32_t cshift;     // log2(bytes per subsequence)
  __device__ __forceinline__ uint32_t word(uint32_t byte_off) const {  // byte_off: multiple of 4, relative to the region
    // chunk * (words per chunk + 1 pad word) + word inside the chunk == byte_off / 4 + chunk.  Raw (little-endian) word: the
    // reader swaps it when it APPENDS it, so that nothing waits for the load at the place where it is issued
    return w[(byte_off >> 2) + (byte_off >> cshift)];
  }
};
struct Bits {  // MSB-first reader over a staged region; region_bit = global bit index of the region's first bit
  Staged st;
  uint32_t region_bit, next;  // next: byte offset (relative, multiple of 4) of the next word to append
  uint32_t pre;               // that word, already loaded: the LDS read of a refill is never on a symbol's critical path
  uint64_t acc;
  int n;
  __device__ __forceinline__ void seek(uint32_t bit) {
    const uint32_t rel = bit - region_bit;
    next = (rel >> 5) << 2;
    acc = 0;
    n = 0;
    pre = st.word(next);
    fill();
    fill();
    n -= (int)(rel & 31u);
  }
  // after fill(): n > 32, enough for one whole symbol (code <= 16 bits + magnitude <= 15 bits)
  __device__ __forceinline__ void fill() {
    if (n <= 32) {
      acc = (acc << 32) | __builtin_bswap32(pre);
      next += 4;
      n += 32;
      pre = st.word(next);
    }
  }
  __device__ __forceinline__ uint32_t pos() const { return region_bit + next * 8u - (uint32_t)n; }
  __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)(acc >> (n - k)) & ((1u << k) - 1u); }
  __device__ __forceinline__ void skip(int k) { n -= k; }
};
// stage bytes [first, first + 64 * cb + 16) of the clean stream (zeros past its end) into the wave's LDS region
__device__ __forceinline__ void stage_wave(const uint8_t* __restrict__ clean, uint32_t nclean, uint32_t first, uint32_t cshift, uint32_t* lds,
                                           uint32_t lane, uint32_t nchunks = 64u, uint32_t nthreads = 64u) {
  const uint32_t cb = 1u << cshift, total = nchunks * cb + 16u, pitch = (cb >> 2) + 1u;
  for (uint32_t off = lane * 16u; off < total; off += nthreads * 16u) {
    const uint32_t g = first + off;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g + 16u <= nclean && ((uintptr_t)(clean + g) & 15u) == 0) {
      v = *(const uint4*)(clean + g);
    } else {
      uint32_t t[4] = {0, 0, 0, 0};
      for (uint32_t k = 0; k < 16; k++)
        if (g + k < nclean) t[k >> 2] |= (uint32_t)clean[g + k] << (8 * (k & 3));
      v = make_uint4(t[0], t[1], t[2], t[3]);
    }
    const uint32_t chunk = off >> cshift, in = off & (cb - 1u);
    uint32_t* d = lds + chunk * pitch + (in >> 2);
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

// One decode step: a Huffman code (two-level table: 9 bits, then 7 more for the long codes) and its magnitude bits.
// Written without data-dependent branches -- the 64 lanes of a wave are in 64 different places of 64 different blocks,
// and every branch would be paid by all of them.  k == 0 is the DC step (symbol = size category), k > 0 an AC step
// (symbol = run << 4 | size).  Returns the coefficient value and where it goes; k is advanced (64 = block finished).
struct Step {
  int value;
  uint32_t zzpos;
  bool store, is_dc, bad;
};
__device__ __forceinline__ Step decode_step(Bits& r, const HuffFastTable* tabs, uint32_t comp, uint32_t& k) {
  const bool is_dc = k == 0;
  const HuffFastTable& t = tabs[(comp ? 2u : 0u) + (is_dc ? 0u : 1u)];
  const uint32_t w16 = r.peek(16);
  uint32_t e = t.l1[w16 >> 7];
  if (__builtin_amdgcn_ballot_w64((e & 0x8000u) != 0) != 0) {  // some lane holds a code longer than 9 bits
    const uint32_t e2 = t.l2[(e & 0x8000u) ? (e & 31u) : 0u][w16 & 127u];
    e = (e & 0x8000u) ? e2 : e;
  }
  Step o;
  o.bad = e == 0;
  const uint32_t len = o.bad ? 16u : (e >> 8) & 31u, rs = o.bad ? 0u : e & 255u;
  r.skip((int)len);
  uint32_t s = is_dc ? rs : rs & 15u;
  const uint32_t run = is_dc ? 0u : rs >> 4;
  if (s > 15u) { o.bad = true; s = 0; }  // a DC size category beyond 15
  const int raw = s ? (int)r.peek((int)s) : 0;
  r.skip((int)s);
  o.value = (s && raw < (1 << (s - 1))) ? raw - (1 << s) + 1 : raw;  // HUFF_EXTEND
  o.is_dc = is_dc;
  const bool marker = !is_dc && s == 0;  // EOB or ZRL
  const uint32_t k2 = k + run;
  const bool over = !marker && k2 > 63u;
  o.bad = o.bad || over;
  o.zzpos = k2 & 63u;
  o.store = !marker && !over && !is_dc;
  k = marker ? (run == 15u ? k + 16u : 64u) : (over ? 64u : k2 + 1u);
  return o;
}

__device__ __forceinline__ uint64_t pack_state(uint32_t p, uint32_t b, uint32_t k) { return (uint64_t)p | ((uint64_t)b << 32) | ((uint64_t)k << 40); }

// per-workgroup constants of the scan layout, in LDS
struct __attribute__((aligned(16))) ScanLds {
  HuffFastTable t[4];
  uint8_t zz[64];
  uint8_t comp[16], yo[16], xo[16];  // block inside the MCU -> component, row / column of the block inside the component's MCU tile
  int vs[4], hs[4], bw[4], bh[4];
  int16_t* coef[4];
};
// TRACK: the state-tracking form of the tables (see track_span) instead of the symbol form
template <bool TRACK = false>
__device__ __forceinline__ void load_scan_lds(const HuffSyncArgs& a, ScanLds& L) {
  copy_words_to_lds((uint32_t*)L.t, (const uint32_t*)(TRACK ? a.ttabs : a.ftabs), (uint32_t)(sizeof(HuffFastTable) * 4 / 4), threadIdx.x, blockDim.x);
  if (threadIdx.x < 64) L.zz[threadIdx.x] = a.zigzag[threadIdx.x];
  if (threadIdx.x < 16) {
    const int j = (int)threadIdx.x;
    int c = a.comp_of[j];
    if (c >= a.ncomp) c = 0;
    const int jj = j - a.first_blk[c];
    L.comp[j] = (uint8_t)c;
    L.yo[j] = (uint8_t)(jj >= 0 ? jj / a.hs[c] : 0);
    L.xo[j] = (uint8_t)(jj >= 0 ? jj % a.hs[c] : 0);
  }
  if (threadIdx.x < 3) {
    const int c = (int)threadIdx.x;
    L.vs[c] = a.vs[c]; L.hs[c] = a.hs[c]; L.bw[c] = a.bw[c]; L.bh[c] = a.bh[c]; L.coef[c] = a.coef[c];
  }
}

// Restart intervals in the self-synchronising scheme.  The unstuff pass removes the RSTn markers and flags the byte at
// which every interval starts; what remains between two intervals are the 0-7 one-bits that pad the last byte (T.81
// F.1.2.3; libjpeg's emit_restart pads with ones).  Rule, applied by every decoder -- lost or not -- when it has just
// finished the last block of what it takes for an MCU, at bit `pos`: if the next byte boundary is an interval start and all
// bits up to it are ones, go to that boundary.  On the true path this is exactly the end of an interval (an MCU that is
// still to come would start with a Huffman code, and no code consists of ones only); on a lost path it is one more
// deterministic transition -- and one that puts it in step.  Checked after the last block of an MCU only (intervals are whole MCUs): the
// test sits in a branch that 64 lanes in 64 different places enter almost every step if it is taken per block.
// Returns the number of bits skipped, kNoJump when the rule does not apply.
constexpr uint32_t kNoJump = 0xffffffffu;
__device__ __forceinline__ uint32_t restart_jump(const HuffSyncArgs& a, Bits& r, uint32_t pos) {
  const uint32_t byte = (pos + 7u) >> 3, pad = (0u - pos) & 7u;
  // the bits first (registers): three block ends in four are ruled out without touching the map in global memory
  r.fill();
  if (r.peek((int)pad) != (1u << pad) - 1u) return kNoJump;
  if (((a.rst_map[byte >> 5] >> (byte & 31u)) & 1u) == 0) return kNoJump;
  r.skip((int)pad);
  return pad;
}

// State tracking without symbols: the passes that only need to know WHERE the decoder is (bit position, block in the
// MCU, zig-zag index) read tables whose entries hold the bits a symbol consumes (code + magnitude bits) and the zig-zag
// advance (run + 1; 16 for ZRL; 64 = to the end of the block for EOB and for undefined codes; 1 for a DC symbol) --
// host: make_track_table.  Identical state transitions to decode_step, at about half the instructions per symbol.
__device__ __forceinline__ void track_span(const HuffSyncArgs& a, const Staged& st, uint32_t region_bit, const ScanLds& L, uint32_t& p, uint32_t& b,
                                           uint32_t& k, uint32_t end_bit, uint32_t& nblk) {
  Bits r;
  r.st = st;
  r.region_bit = region_bit;
  r.seek(p);
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  uint32_t cpack = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) cpack |= ((uint32_t)a.comp_of[j] & 3u) << (2 * j);
  constexpr uint32_t kWords = sizeof(HuffFastTable) / 2;
  const uint16_t* T = (const uint16_t*)L.t;
  uint32_t cbase = ((cpack >> (2u * b)) & 3u) ? 2u * kWords : 0u;  // the component's DC table; its AC table follows
  int left = (int)(end_bit - p);
  while (left > 0) {
    r.fill();
    const uint32_t w16 = r.peek(16);
    const uint32_t tb = cbase + (k ? kWords : 0u);
    uint32_t e = T[tb + (w16 >> 7)];
    if (__builtin_amdgcn_ballot_w64((e & 0x8000u) != 0) != 0) {
      const uint32_t e2 = T[tb + 512u + ((e & 0x8000u) ? (e & 31u) : 0u) * 128u + (w16 & 127u)];
      e = (e & 0x8000u) ? e2 : e;
    }
    const int adv = (int)(e & 31u);
    r.skip(adv);
    left -= adv;
    k += (e >> 5) & 127u;
    if (k >= 64u) {
      k = 0;
      b++;
      nblk++;
      if (b == bpm) {
        b = 0;
        if (a.rst_map) {  // restart intervals: see restart_jump
          const uint32_t pad = restart_jump(a, r, end_bit - (uint32_t)left);
          if (pad != kNoJump) left -= (int)pad;
        }
      }
      cbase = ((cpack >> (2u * b)) & 3u) ? 2u * kWords : 0u;
    }
  }
  p = r.pos();
}

// Round 6: the same walk, up to two symbols per step (pair tables: uhdr_types.h kHuffPairBits, host make_pair_table).  The
// tracking passes are VALU-issue bound -- ~50 instructions per step for a wave whose 64 lanes are in 64 different places, 6 hypotheses
// x 2 passes over every bit of a 4:2:0 scan -- so the number of steps is the cost.  A table entry covers the AC symbol that follows
// the first one when its code lies inside the same 10 index bits; the step takes it unless the first symbol ends the block (the
// next table would be another one) or reaches the end of the subsequence (the walk must stop at the FIRST boundary at or beyond
// end_bit): then exactly the single-symbol transition happens.  4K 4:2:0 q95: 0.59 steps per symbol, gain map 0.60.
// Every state this walk passes through is a state of track_span's walk, and the final state is the same.
struct __attribute__((aligned(16))) PairLds {  // == the layout of HuffSyncArgs::ptabs
  uint32_t p[4][kHuffPairWords];        // 16 KB
  uint16_t l2[kHuffL2Total * 128];      // 8 KB: the second level of the tracking form, for codes longer than the index (all four tables', packed)
};
static_assert(sizeof(PairLds) == kHuffPairBlobWords * 4, "PairLds is one copy of the host's blob");
__device__ __forceinline__ void load_pair_lds(const HuffSyncArgs& a, PairLds& L) {
  copy_words_to_lds((uint32_t*)&L, a.ptabs, (uint32_t)kHuffPairBlobWords, threadIdx.x, blockDim.x);
}
__device__ __forceinline__ void track_span_pair(const HuffSyncArgs& a, const Staged& st, uint32_t region_bit, const PairLds& L, uint32_t& p, uint32_t& b,
                                                uint32_t& k, uint32_t end_bit, uint32_t& nblk) {
  Bits r;
  r.st = st;
  r.region_bit = region_bit;
  r.seek(p);
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  uint32_t cpack = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) cpack |= ((uint32_t)a.comp_of[j] & 3u) << (2 * j);
  const uint32_t* P = &L.p[0][0];
  const uint16_t* S = &L.l2[0];
  uint32_t cls = ((cpack >> (2u * b)) & 3u) ? 2u : 0u;  // the component's DC table; its AC table follows
  int left = (int)(end_bit - p);
  while (left > 0) {
    r.fill();
    const uint32_t w16 = r.peek(16);
    const uint32_t ti = cls + (k ? 1u : 0u);
    uint32_t e = P[ti * (uint32_t)kHuffPairWords + (w16 >> (16 - kHuffPairBits))];
    if (__builtin_amdgcn_ballot_w64((e >> 31) != 0) != 0) {
      const uint32_t e2 = S[((e >> 31) ? (e & 63u) : 0u) * 128u + (w16 & 127u)];
      e = (e >> 31) ? e2 : e;  // tracking form: no second symbol
    }
    const uint32_t adv1 = e & 31u, kinc1 = (e >> 5) & 127u, adv2 = (e >> 12) & 31u, kinc2 = (e >> 17) & 127u;
    const bool two = adv2 != 0u && k + kinc1 < 64u && left > (int)adv1;
    const uint32_t adv = adv1 + (two ? adv2 : 0u);
    r.skip((int)adv);
    left -= (int)adv;
    k += kinc1 + (two ? kinc2 : 0u);
    if (k >= 64u) {
      k = 0;
      b++;
      nblk++;
      if (b == bpm) {
        b = 0;
        if (a.rst_map) {  // restart intervals: see restart_jump
          const uint32_t pad = restart_jump(a, r, end_bit - (uint32_t)left);
          if (pad != kNoJump) left -= (int)pad;
        }
      }
      cls = ((cpack >> (2u * b)) & 3u) ? 2u : 0u;
    }
  }
  p = r.pos();
}

// The same walk with notes for the write pass's pieces (HuffSyncArgs::pieces): the subsequence that starts at bit sub_start is crossed piece by
// piece; after piece q < pieces - 1 the state (the first symbol boundary at or beyond the cut) and the blocks completed inside the piece go to
// mid[q] / cnt[q].  Same states as one track_span_pair call over the whole span (a walk's boundaries do not depend on where it pauses); the
// price is that a wave's lanes wait for each other at every cut instead of only at the end.
__device__ __forceinline__ void track_span_pieces(const HuffSyncArgs& a, const Staged& st, uint32_t region_bit, const PairLds& L, uint32_t& p, uint32_t& b,
                                                  uint32_t& k, uint32_t sub_start, uint32_t end_bit, uint32_t& nblk, uint64_t* __restrict__ mid,
                                                  uint16_t* __restrict__ cnt) {
  const uint32_t Q = (uint32_t)a.pieces, piece = a.sub_bits / Q;
#pragma unroll
  for (uint32_t q = 0; q < 4u; q++) {  // (at most four pieces; unrolled: callers may keep mid / cnt in registers)
    if (q < Q) {
      const uint32_t cut = q + 1u < Q ? min(sub_start + (q + 1u) * piece, end_bit) : end_bit;
      uint32_t nb = 0;
      if (p < cut) track_span_pair(a, st, region_bit, L, p, b, k, cut, nb);
      nblk += nb;
      if (q + 1u < Q) {
        mid[q] = pack_state(p, b, k);
        cnt[q] = (uint16_t)nb;
      }
    }
  }
}

// Decodes from state (p, b, k) to the first symbol boundary at or beyond end_bit.  WRITE: store coefficients / DC
// differences for the blocks [blk, total_blocks) and flag malformed data; else just track the state.
template <bool WRITE>
__device__ __forceinline__ void run_span(const HuffSyncArgs& a, const Staged& st, uint32_t region_bit, const ScanLds& L, uint32_t& p, uint32_t& b,
                                         uint32_t& k, uint32_t end_bit, uint32_t& nblk, uint32_t blk) {
  Bits r;
  r.st = st;
  r.region_bit = region_bit;
  r.seek(p);
  bool bad = false;
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  // WRITE: the block's position, tracked incrementally (MCU column / row, block inside the MCU)
  int mx = 0, my = 0;
  int16_t* dst = nullptr;
  auto locate = [&]() {  // the JBLOCK of the current block, nullptr for a dummy block of an edge MCU
    const int c = L.comp[b];
    const int by = my * L.vs[c] + L.yo[b], bx = mx * L.hs[c] + L.xo[b];
    dst = (by < L.bh[c] && bx < L.bw[c]) ? L.coef[c] + ((size_t)by * L.bw[c] + bx) * 64 : nullptr;
  };
  if (WRITE) {
    const uint32_t m = blk / bpm;
    my = (int)(m / (uint32_t)a.mcus_per_row);
    mx = (int)(m - (uint32_t)my * (uint32_t)a.mcus_per_row);
    locate();
  }
  // block inside the MCU -> component, two bits each: wave-uniform, so the lookup is a shift instead of an LDS read
  uint32_t cpack = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) cpack |= ((uint32_t)a.comp_of[j] & 3u) << (2 * j);
  while (r.pos() < end_bit && (!WRITE || blk < a.total_blocks)) {
    r.fill();
    const Step o = decode_step(r, L.t, (cpack >> (2u * b)) & 3u, k);
    bad = bad || o.bad;
    if (WRITE) {
      if (o.is_dc) a.dcd[blk] = o.value;
      if (o.store && dst) dst[L.zz[o.zzpos]] = (int16_t)o.value;
    }
    if (k >= 64u) {
      k = 0;
      b++;
      nblk++;
      blk++;
      if (b == bpm) {
        b = 0;
        if (WRITE) {
          mx++;
          if (mx == a.mcus_per_row) { mx = 0; my++; }
        }
      }
      if (WRITE) locate();
    }
  }
  p = r.pos();
  if (WRITE && bad) atomicOr(a.flags + 1, 2u);
}

// ---- the write pass ------------------------------------------------------------------------------------------------------
// Tables in VALUE form (host: make_value_table), one 32-bit word per entry:
//   bits 0-4 bits consumed (code + magnitude), 5-11 zig-zag advance (as in the tracking form), 12-15 magnitude bits s,
//   16 malformed (undefined code / DC category beyond 15), 31 "longer code: sub-table in bits 0-4" (first level only).
// One lookup yields everything a symbol needs: the value is the last s of the consumed bits, HUFF_EXTENDed.
struct __attribute__((aligned(16))) WriteLds {
  uint32_t t[4][kHuffValWords];
  uint8_t zz[64];
  uint8_t comp[16], yo[16], xo[16];
  int vs[4], hs[4], bw[4], bh[4];
  int16_t* coef[4];
};
__device__ __forceinline__ void load_write_lds(const HuffSyncArgs& a, WriteLds& L) {
  copy_words_to_lds(&L.t[0][0], a.vtabs, 4u * kHuffValWords, threadIdx.x, blockDim.x);
  if (threadIdx.x < 64) L.zz[threadIdx.x] = a.zigzag[threadIdx.x];
  if (threadIdx.x < 16) {
    const int j = (int)threadIdx.x;
    int c = a.comp_of[j];
    if (c >= a.ncomp) c = 0;
    const int jj = j - a.first_blk[c];
    L.comp[j] = (uint8_t)c;
    L.yo[j] = (uint8_t)(jj >= 0 ? jj / a.hs[c] : 0);
    L.xo[j] = (uint8_t)(jj >= 0 ? jj % a.hs[c] : 0);
  }
  if (threadIdx.x < 3) {
    const int c = (int)threadIdx.x;
    L.vs[c] = a.vs[c]; L.hs[c] = a.hs[c]; L.bw[c] = a.bw[c]; L.bh[c] = a.bh[c]; L.coef[c] = a.coef[c];
  }
}
// Decodes from the TRUE state (p, b, k) to the first symbol boundary at or beyond end_bit, storing coefficients (natural
// order) and DC differences for the blocks [blk, total_blocks); flags malformed data.  Same transitions as track_span.
__device__ __forceinline__ void write_span(const HuffSyncArgs& a, const Staged& st, uint32_t region_bit, const WriteLds& L, uint32_t p, uint32_t b,
                                           uint32_t k, uint32_t end_bit, uint32_t& nblk, uint32_t blk) {
  Bits r;
  r.st = st;
  r.region_bit = region_bit;
  r.seek(p);
  bool bad = false, rst_bad = false;
  uint32_t seq1 = 0, seq2 = 0;
  uint32_t until = a.rst_map ? a.rst_blocks - blk % a.rst_blocks : 0u;  // blocks to the next interval end (a countdown: no division per MCU)
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  uint32_t cpack = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) cpack |= ((uint32_t)a.comp_of[j] & 3u) << (2 * j);
  const uint32_t m0 = blk / bpm;
  int my = (int)(m0 / (uint32_t)a.mcus_per_row), mx = (int)(m0 - (uint32_t)my * (uint32_t)a.mcus_per_row);
  int16_t* dst = nullptr;
  auto locate = [&]() {  // the JBLOCK of the current block, nullptr for a dummy block of an edge MCU
    const int c = L.comp[b];
    const int by = my * L.vs[c] + L.yo[b], bx = mx * L.hs[c] + L.xo[b];
    dst = (by < L.bh[c] && bx < L.bw[c]) ? L.coef[c] + ((size_t)by * L.bw[c] + bx) * 64 : nullptr;
  };
  locate();
  const uint32_t* T = &L.t[0][0];
  uint32_t cbase = ((cpack >> (2u * b)) & 3u) ? 2u * kHuffValWords : 0u;
  int left = (int)(end_bit - p);
  while (left > 0 && blk < a.total_blocks) {
    r.fill();
    const uint32_t w16 = r.peek(16);
    const uint32_t tb = cbase + (k ? (uint32_t)kHuffValWords : 0u);
    uint32_t e = T[tb + (w16 >> 7)];
    if (__builtin_amdgcn_ballot_w64((e >> 31) != 0) != 0) {
      const uint32_t e2 = T[tb + 512u + ((e >> 31) ? (e & 31u) : 0u) * 128u + (w16 & 127u)];
      e = (e >> 31) ? e2 : e;
    }
    const uint32_t adv = e & 31u, kinc = (e >> 5) & 127u, sz = (e >> 12) & 15u;
    const uint32_t ext = (1u << sz) - 1u;
    const uint32_t raw = r.peek((int)adv) & ext;  // adv >= 1: every code is at least one bit long
    const int value = (int)raw - (((raw << 1) > ext) ? 0 : (int)ext);  // HUFF_EXTEND; 0 when there are no magnitude bits
    const uint32_t zzpos = k + kinc - 1u;  // AC symbol with a value: k + run
    const bool is_dc = k == 0;
    const bool over = !is_dc && sz != 0 && zzpos > 63u;
    bad = bad || ((e >> 16) & 1u) != 0 || over;
    if (is_dc) a.dcd[blk] = value;
    if (!is_dc && sz != 0 && !over && dst) dst[L.zz[zzpos & 63u]] = (int16_t)value;
    r.skip((int)adv);
    left -= (int)adv;
    k += kinc;
    if (k >= 64u) {
      k = 0;
      b++;
      nblk++;
      blk++;
      until--;
      if (b == bpm) {
        b = 0;
        mx++;
        if (mx == a.mcus_per_row) { mx = 0; my++; }
        if (a.rst_map) {
          // the true path: an interval must end exactly where the frame header says (every rst_blocks blocks), nowhere else
          const uint32_t pos0 = end_bit - (uint32_t)left;
          const uint32_t pad = restart_jump(a, r, pos0);
          const bool due = until == 0 && blk < a.total_blocks;
          if (until == 0) until = a.rst_blocks;
          if ((pad != kNoJump) != due) rst_bad = true;
          if (pad != kNoJump) {
            left -= (int)pad;
            const uint32_t at = (pos0 + 7u) >> 3, m = ((blk / a.rst_blocks - 1u) & 7u) + 1u;  // the number this marker must have had
            seq1 += rst_weight1(at) * m;
            seq2 += rst_weight2(at) * m;
          }
        }
      }
      cbase = ((cpack >> (2u * b)) & 3u) ? 2u * kHuffValWords : 0u;
      locate();
    }
  }
  if (bad) atomicOr(a.flags + 1, 2u);
  if (rst_bad) atomicOr(a.flags + 1, 4u);
  if (seq1 | seq2) {  // the caller compares with the sums of the markers that were found: every one an interval end, numbered in turn
    atomicAdd(a.flags + 0, seq1);
    atomicAdd(a.flags + 7, seq2);
  }
}

// rounds of step 2 / 3.  state[cur] is read, state[cur ^ 1] written; changed[] likewise.  flags[4 + r % 3] counts the
// changes of round r; a round clears the counter of the round after it.
__global__ __launch_bounds__(kSyncBlock) void sync_round_kernel(const HuffSyncArgs a, int round) {
  extern __shared__ uint32_t s_stage[];
  __shared__ ScanLds L;
  if (blockIdx.x == 0 && threadIdx.x == 0) a.flags[4 + (round + 1) % 3] = 0;  // always: also a round that returns early
  if (round > 0 && a.flags[4 + (round - 1) % 3] == 0) return;  // the previous round changed nothing: fixed point reached
  const uint32_t i = blockIdx.x * kSyncBlock + threadIdx.x;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const int cur = round & 1;
  const uint64_t* sin = a.state[cur];
  uint64_t* sout = a.state[cur ^ 1];
  const uint8_t* cin = a.changed[cur];
  uint8_t* cout = a.changed[cur ^ 1];
  // does any lane of this wave have work?  (round > 0: only lanes whose predecessor's end state changed)
  bool active = i < nsub && (round == 0 || (i > 0 && cin[i - 1]));
  if (__builtin_amdgcn_ballot_w64(active) == 0) {
    if (i < nsub) { sout[i] = sin[i]; cout[i] = 0; }
    return;
  }
  load_scan_lds<true>(a, L);
  const uint32_t cshift = 31u - (uint32_t)__builtin_clz(a.sub_bits >> 3);
  const uint32_t first_byte = blockIdx.x * kSyncBlock * (a.sub_bits >> 3);
  stage_wave(a.clean, nclean, first_byte, cshift, s_stage, threadIdx.x);
  __syncthreads();
  if (i >= nsub) return;
  if (!active) {  // my start state is what it was: so is my end state
    sout[i] = sin[i];
    cout[i] = 0;
    return;
  }
  uint32_t p, b, k;
  if (round == 0) {
    p = i * a.sub_bits; b = 0; k = 0;
  } else {
    const uint64_t s = sin[i - 1];
    p = (uint32_t)s; b = (uint32_t)(s >> 32) & 0xffu; k = (uint32_t)(s >> 40) & 0xffu;
  }
  const uint32_t end_bit = min((i + 1) * a.sub_bits, nbits);
  uint32_t nblk = 0;
  const Staged st = {s_stage, cshift};
  if (p < end_bit) track_span(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk);
  const uint64_t e = pack_state(p, b, k);
  const bool ch = round == 0 || e != sin[i];
  sout[i] = e;
  cout[i] = ch ? 1 : 0;
  a.nblk[i] = nblk;
  if (ch) atomicAdd(a.flags + 4 + round % 3, 1u);
}

// blockDim.x / 64 waves share one copy of the value tables; every wave stages its own 64 subsequences
__global__ __launch_bounds__(256) void sync_write_kernel(const HuffSyncArgs a, int final_buf) {
  extern __shared__ uint32_t s_stage_all[];
  __shared__ WriteLds L;
  load_write_lds(a, L);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const uint32_t cshift = 31u - (uint32_t)__builtin_clz(a.sub_bits >> 3);
  const uint32_t wv = threadIdx.x >> 6;
  uint32_t* s_stage = s_stage_all + wv * ((64u * ((a.sub_bits >> 3) + 4u) + 64u) >> 2);
  const uint32_t first_byte = (blockIdx.x * blockDim.x + wv * 64u) * (a.sub_bits >> 3);
  stage_wave(a.clean, nclean, first_byte, cshift, s_stage, threadIdx.x & 63u);
  __syncthreads();
  if (i >= nsub) return;
  uint32_t p = 0, b = 0, k = 0;
  if (i > 0) {
    const uint64_t s = a.state[final_buf][i - 1];
    p = (uint32_t)s; b = (uint32_t)(s >> 32) & 0xffu; k = (uint32_t)(s >> 40) & 0xffu;
  }
  const uint32_t end_bit = min((i + 1) * a.sub_bits, nbits);
  uint32_t nblk = 0;
  const Staged st = {s_stage, cshift};
  const uint32_t blk0 = a.nblk[i] + tile_offset(a);  // nblk[] holds the per-tile exclusive scan by now
  if (p < end_bit) write_span(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk, blk0);
  if (i == nsub - 1) {
    // the scan must hold exactly total_blocks blocks: fewer = truncated data, more cannot be seen (the loop stops there)
    if (blk0 + nblk < a.total_blocks) atomicOr(a.flags + 1, 8u);
  }
}

// ---- the write pass, form 2 (round 5; marker-less scans) ---------------------------------------------------------------
// Form 1 above pays, on nearly every symbol step of a wave (64 lanes in 64 different places: some lane ends a block on 9 steps
// in 10), for the block's JBLOCK address (component, MCU row / column, edge test: eight LDS reads and a few multiplications), a
// zig-zag lookup and a separate DC-difference store.  Form 2 stores where the decoder already is: block t of the scan at
// coef_scan + 64 t, coefficient at its ZIG-ZAG index, the DC difference at [0] like any other value -- one predicated
// 2-byte store per symbol, a pointer bump per block.  coef_place_kernel then reads the scratch once (coalesced), applies the
// DC prediction (the running sums of dc_partial / dc_scan_partials over [0]) and writes every JBLOCK whole, in natural order --
// so the component arrays need no zero fill any more, the scratch (scan order) gets it instead.
// Round 6: up to two symbols per step here as well (pair form of the value tables, host make_pair_value_table): this pass has one
// lane per subsequence -- 290 waves for a 4K gain map, 790 for the base image, on 1024 SIMDs -- so it runs at the latency of one
// lane's symbol chain, and 0.6 steps per symbol are 0.6 of its time.
struct __attribute__((aligned(16))) Write2Lds {  // == the layout of HuffSyncArgs::pvtabs
  uint2 p[4][kHuffPairWords];             // 32 KB: {first symbol, second symbol or 0}, value form
  uint16_t l2[kHuffL2Total * 128];        // 8 KB: the sub-tables of the value form, packed (codes longer than the index): bits | advance << 5 |
                                          // magnitude bits << 12; 0 = a malformed code (16 bits, to the end of the block / one DC step)
};
__device__ __forceinline__ void write_span2(const HuffSyncArgs& a, const Staged& st, uint32_t region_bit, const Write2Lds& L, uint32_t p, uint32_t b, uint32_t k,
                                            uint32_t end_bit, uint32_t& nblk, uint32_t blk) {
  Bits r;
  r.st = st;
  r.region_bit = region_bit;
  r.seek(p);
  bool bad = false;
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  uint32_t cpack = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) cpack |= ((uint32_t)a.comp_of[j] & 3u) << (2 * j);
  const uint2* P = &L.p[0][0];
  const uint16_t* S = &L.l2[0];
  uint32_t cls = ((cpack >> (2u * b)) & 3u) ? 2u : 0u;
  int16_t* dst = a.coef_scan + (size_t)blk * 64;
  int left = (int)(end_bit - p);
  while (left > 0 && blk < a.total_blocks) {
    r.fill();
    const uint32_t w16 = r.peek(16);
    const uint32_t ti = cls + (k ? 1u : 0u);
    const uint2 e = P[ti * (uint32_t)kHuffPairWords + (w16 >> (16 - kHuffPairBits))];
    uint32_t e1 = e.x, e2 = e.y;
    if (__builtin_amdgcn_ballot_w64((e1 >> 31) != 0) != 0) {  // a code longer than the index: its sub-table, one symbol
      uint32_t s2 = S[((e1 >> 31) ? (e1 & 63u) : 0u) * 128u + (w16 & 127u)];
      s2 = s2 ? s2 : (16u | ((k ? 64u : 1u) << 5) | (1u << 16));  // malformed: make_value_table's entry for an undefined code
      e1 = (e1 >> 31) ? s2 : e1;
    }
    const uint32_t adv1 = e1 & 31u, kinc1 = (e1 >> 5) & 127u, sz1 = (e1 >> 12) & 15u;
    const bool two = e2 != 0u && k + kinc1 < 64u && left > (int)adv1;
    e2 = two ? e2 : 0u;
    const uint32_t adv2 = e2 & 31u, kinc2 = (e2 >> 5) & 127u, sz2 = (e2 >> 12) & 15u;
    const uint32_t adv = adv1 + adv2;
    const uint32_t bits = r.peek((int)adv);  // adv >= 1: every code is at least one bit long
    const uint32_t ext1 = (1u << sz1) - 1u, ext2 = (1u << sz2) - 1u;
    const uint32_t raw1 = (bits >> adv2) & ext1, raw2 = bits & ext2;
    const int v1 = (int)raw1 - (((raw1 << 1) > ext1) ? 0 : (int)ext1);  // HUFF_EXTEND; 0 when there are no magnitude bits
    const int v2 = (int)raw2 - (((raw2 << 1) > ext2) ? 0 : (int)ext2);
    const uint32_t zz1 = k + kinc1 - 1u, zz2 = zz1 + kinc2;  // DC: 0; AC symbol with a value: k + run
    const bool over1 = sz1 != 0 && zz1 > 63u, over2 = sz2 != 0 && zz2 > 63u;  // (a DC symbol has zzpos 0)
    bad = bad || ((e1 >> 16) & 1u) != 0 || over1 || over2;
    if (sz1 != 0 && !over1) dst[zz1] = (int16_t)v1;
    if (sz2 != 0 && !over2) dst[zz2] = (int16_t)v2;
    r.skip((int)adv);
    left -= (int)adv;
    k += kinc1 + kinc2;
    if (k >= 64u) {
      k = 0;
      b++;
      nblk++;
      blk++;
      dst += 64;
      if (b == bpm) b = 0;
      cls = ((cpack >> (2u * b)) & 3u) ? 2u : 0u;
    }
  }
  if (bad) atomicOr(a.flags + 1, 2u);
}
__global__ __launch_bounds__(256) void sync_write2_kernel(const HuffSyncArgs a, int final_buf) {
  extern __shared__ __attribute__((aligned(16))) uint32_t s_dyn[];
  // the tables sit in the dynamic segment as well (static + dynamic LDS beyond 64 KB needs the opt-in either way)
  Write2Lds& L = *(Write2Lds*)s_dyn;
  uint32_t* s_stage_all = s_dyn + sizeof(Write2Lds) / 4;
  static_assert(sizeof(Write2Lds) == kHuffPairValBlobWords * 4, "Write2Lds is one copy of the host's blob");
  copy_words_to_lds(s_dyn, a.pvtabs, (uint32_t)kHuffPairValBlobWords, threadIdx.x, blockDim.x);
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const uint32_t cshift = 31u - (uint32_t)__builtin_clz(a.sub_bits >> 3);
  const uint32_t wv = threadIdx.x >> 6;
  uint32_t* s_stage = s_stage_all + wv * ((64u * ((a.sub_bits >> 3) + 4u) + 64u) >> 2);
  const uint32_t first_byte = (blockIdx.x * blockDim.x + wv * 64u) * (a.sub_bits >> 3);
  stage_wave(a.clean, nclean, first_byte, cshift, s_stage, threadIdx.x & 63u);
  __syncthreads();
  if (i >= nsub) return;
  uint32_t p = 0, b = 0, k = 0;
  if (i > 0) {
    const uint64_t s = a.state[final_buf][i - 1];
    p = (uint32_t)s; b = (uint32_t)(s >> 32) & 0xffu; k = (uint32_t)(s >> 40) & 0xffu;
  }
  const uint32_t end_bit = min((i + 1) * a.sub_bits, nbits);
  uint32_t nblk = 0;
  const Staged st = {s_stage, cshift};
  const uint32_t blk0 = a.nblk[i] + tile_offset(a);  // nblk[] holds the per-tile exclusive scan by now
  if (p < end_bit) write_span2(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk, blk0);
  if (i == nsub - 1) {
    if (blk0 + nblk < a.total_blocks) atomicOr(a.flags + 1, 8u);  // truncated data
  }
}

// ---- hypothesis decode: the fixed point in a fixed number of passes ------------------------------------------------------
// The rounds above converge slowly on interleaved scans: a lane that starts in the wrong BLOCK OF THE MCU reads luma
// blocks with the chroma tables (or the reverse) and can only fall in step with the true decoder by luck, so the correct
// state advances about one subsequence per round (4:2:0: ~18 rounds for a 4K frame).  Here every subsequence is decoded
// once per possible block position h (H = blocks per MCU hypotheses, one wave each, sharing the staged bytes):
//   pass 0   slot h of subsequence i: start at its first bit as block h, zig-zag 0 -> state at the end of i;
//   pass 1   every path goes on through i+1, i+2, ... (at most `levels` subsequences) until its state at the end of
//            subsequence j equals a FRESH slot g of j -- from there on it IS that path.  Until then it occupies slot
//            l*H + h of j (l = j - i).  map[j-1][slot] = the slot it is in at the end of j; cnt[j-1][slot] = blocks it
//            completed inside j;
//   chain    the true path is slot 0 of subsequence 0; g(i+1) = map[i][g(i)] -- a scan over function composition.  It
//            yields every subsequence's true start state (state[0][i] = hyp_state[i][g(i)]) and block count;
//   then the write pass and the DC scan of the round scheme, unchanged.
// A path that does not merge within `levels` subsequences leaves map = 0xff; if the true path hits one, flags[2] is set
// and the caller runs the rounds instead.

__global__ __launch_bounds__(1024) void hyp_pass0_kernel(const HuffSyncArgs a) {
  extern __shared__ uint32_t s_stage[];
  __shared__ PairLds L;
  load_pair_lds(a, L);
  const uint32_t lane = threadIdx.x & 63u, h = threadIdx.x >> 6;
  const uint32_t i = blockIdx.x * 64u + lane;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const uint32_t cshift = 31u - (uint32_t)__builtin_clz(a.sub_bits >> 3);
  const uint32_t first_byte = blockIdx.x * 64u * (a.sub_bits >> 3);
  stage_wave(a.clean, nclean, first_byte, cshift, s_stage, threadIdx.x, 64u, blockDim.x);
  if (a.zero_vec) {
    // the write pass's scan-order scratch: zeros (25-50 MB at 4K).  BEHIND the last global load of this kernel: the walk below reads LDS only, so
    // the stores drain while it runs.  In front of the staging loads (where this loop stood until round 6) every wave waited for its 56 KB of
    // stores before its first symbol -- memory operations of a wave complete in order.
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < a.zero_vec; v += gridDim.x * blockDim.x) a.zero_ptr[v] = z;
  }
  __syncthreads();
  if (i >= nsub) return;
  uint32_t p = i * a.sub_bits, b = h, k = 0, nblk = 0;
  const uint32_t end_bit = min((i + 1) * a.sub_bits, nbits);
  const Staged st = {s_stage, cshift};
  if (a.pieces > 1 && blockIdx.x == 0 && h == 0) {
    // the wave that holds the stream's first subsequence -- the true path's -- walks in pieces (the WHOLE wave: a branch for that one lane would
    // make the wave walk twice, and this pass lasts as long as its slowest wave); lane 0's notes go where the chain's walk puts all the others
    uint64_t m[3] = {0, 0, 0};
    uint16_t c16[3] = {0, 0, 0};
    track_span_pieces(a, st, first_byte * 8u, L, p, b, k, i * a.sub_bits, end_bit, nblk, m, c16);
    if (i == 0) {
      uint32_t sum = 0;
#pragma unroll
      for (int q = 0; q < 3; q++)
        if (q + 1 < a.pieces) { a.pend[q] = m[q]; a.pcnt[q] = c16[q]; sum += c16[q]; }
      a.pcnt[a.pieces - 1] = nblk - sum;
    }
  } else {
    track_span_pair(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk);
  }
  a.hyp_state[(size_t)i * kHuffHypSlots + h] = pack_state(p, b, k);
  if (i == 0 && h == 0) {
    a.nblk[0] = nblk;
    a.flags[kHuffFlagStragglers] = 0;  // every attempt starts with pass 0: the list pass 1 hands to hyp_straggler_kernel is empty
  }
}

__global__ __launch_bounds__(1024) void hyp_pass1_kernel(const HuffSyncArgs a) {
  extern __shared__ uint32_t s_stage[];
  __shared__ ScanLds L;
  load_scan_lds<true>(a, L);
  const uint32_t lane = threadIdx.x & 63u, h = threadIdx.x >> 6, H = (uint32_t)a.hyp_h;
  const uint32_t i = blockIdx.x * 64u + lane;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const uint32_t cshift = 31u - (uint32_t)__builtin_clz(a.sub_bits >> 3);
  const uint32_t first_byte = (blockIdx.x * 64u + 1u) * (a.sub_bits >> 3);  // the window starts one subsequence further on
  stage_wave(a.clean, nclean, first_byte, cshift, s_stage, threadIdx.x, 64u + (uint32_t)a.hyp_levels, blockDim.x);
  __syncthreads();
  if (i + 1 >= nsub) return;
  const uint64_t s0 = a.hyp_state[(size_t)i * kHuffHypSlots + h];
  uint32_t p = (uint32_t)s0, b = (uint32_t)(s0 >> 32) & 0xffu, k = (uint32_t)(s0 >> 40) & 0xffu;
  uint32_t slot = h;
  const Staged st = {s_stage, cshift};
  for (uint32_t l = 1; l <= (uint32_t)a.hyp_levels; l++) {
    const uint32_t j = i + l;
    if (j >= nsub) break;
    const uint32_t end_bit = min((j + 1) * a.sub_bits, nbits);
    uint32_t nblk = 0;
    if (p < end_bit) track_span(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk);
    const uint64_t e = pack_state(p, b, k);
    const uint64_t* fresh = a.hyp_state + (size_t)j * kHuffHypSlots;
    uint32_t g = 0xffu;
    for (uint32_t t = 0; t < H; t++)
      if (fresh[t] == e && g == 0xffu) g = t;
    const size_t at = (size_t)(j - 1) * kHuffHypSlots + slot;
    a.hyp_cnt[at] = (uint16_t)nblk;
    if (g != 0xffu) {
      a.hyp_map[at] = (uint8_t)g;
      if (a.hyp_hist) atomicAdd(a.flags + 9 + min(l, 6u), 1u);
      break;
    }
    if (l == (uint32_t)a.hyp_levels) {  // map stays 0xff: not merged within the budget
      atomicAdd(a.flags + 3, 1u);
      break;
    }
    const uint32_t nslot = l * H + h;
    a.hyp_state[(size_t)j * kHuffHypSlots + nslot] = e;
    a.hyp_map[at] = (uint8_t)nslot;
    slot = nslot;
  }
}

// pass 1, round 4 form: the workgroup's paths advance level by level in lockstep (one barrier per level), so that a path can
// also fall in step with a path that is IN FLIGHT -- started later, inside this workgroup's 64 subsequences, and already
// some levels old -- not only with the fresh paths of the subsequence it has just finished.  Two decoders on the same stream
// converge on each other long before either meets a decoder that was started a moment ago: a straggler hands over to the
// path one subsequence younger as soon as the two agree, and that one has more levels left than it.  The chain composes
// such links like any other (map[j - 1][slot] = the in-flight slot at the end of j).  A slot m * H + t of subsequence j holds a
// live state iff its owner's previous link continued into it (hyp_map[j - 1][(m - 1) * H + t] == m * H + t; the map is reset
// to 0xff for every attempt), written one or more barriers ago by a lane of this workgroup -- owners in the NEXT workgroup
// (start index beyond this one's last subsequence) are not looked at.  Youngest first: the younger the target, the more
// levels it has left.
// Round 5: (a) the level loop is COMPACTED.  After level 1 some 8 % of a 4:2:0 workgroup's 384 paths are alive -- about five per
// wave, so that nearly every wave still ran level 2 at the full cost of its slowest lane (the second level took 50 us of the
// kernel's 148, the first 70).  Survivors now go into an LDS list and level l >= 2 is run by the first ceil(n / 64) waves, one
// list entry per thread; the barriers (and with them the in-flight merging) stay.  (b) the candidates' states are loaded in
// batches, not one dependent global load per candidate.  (c) after hyp_main_levels levels the list goes to hyp_straggler_kernel.
__device__ __forceinline__ uint32_t hyp_find_slot(const uint64_t* __restrict__ row, const uint8_t* __restrict__ prev, uint64_t e, uint32_t H, uint32_t l,
                                                  uint32_t m_lo) {
  uint32_t g = 0xffu;
  for (uint32_t t0 = 0; t0 < H; t0 += 4) {  // fresh slots: four independent loads in flight
    uint64_t v[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) v[q] = row[min(t0 + q, H - 1u)];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++)
      if (t0 + q < H && v[q] == e && g == 0xffu) g = t0 + q;
  }
  for (uint32_t m = m_lo; m < l && g == 0xffu; m++)  // in flight, youngest first
    for (uint32_t t0 = 0; t0 < H; t0 += 4) {
      uint64_t v[4];
      uint32_t o[4];
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t sl = m * H + min(t0 + q, H - 1u);
        o[q] = prev[sl - H];
        v[q] = row[sl];
      }
#pragma unroll
      for (uint32_t q = 0; q < 4; q++) {
        const uint32_t sl = m * H + t0 + q;
        if (t0 + q < H && g == 0xffu && o[q] == sl && v[q] == e) g = sl;
      }
    }
  return g;
}
__global__ __launch_bounds__(1024) void hyp_pass1q_kernel(const HuffSyncArgs a) {
  extern __shared__ uint32_t s_stage[];
  __shared__ PairLds L;
  __shared__ uint32_t s_n;
  load_pair_lds(a, L);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, H = (uint32_t)a.hyp_h;
  const uint32_t i0 = blockIdx.x * 64u, i_last = i0 + 63u;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const uint32_t cshift = 31u - (uint32_t)__builtin_clz(a.sub_bits >> 3);
  const uint32_t first_byte = (i0 + 1u) * (a.sub_bits >> 3);  // the window starts one subsequence further on
  const uint32_t nchunks = 64u + (uint32_t)a.hyp_levels;
  // the survivor list behind the staged bytes: blockDim.x states (2 x 4 bytes), then blockDim.x ids (lane | hypothesis << 6)
  const uint32_t stage_words = (nchunks * ((a.sub_bits >> 3) + 4u) + 16u + 7u) >> 2;
  uint32_t* s_lo = s_stage + ((stage_words + 1u) & ~1u);  // (two words per state: the dynamic segment is only known to be 4-byte aligned)
  uint32_t* s_hi = s_lo + blockDim.x;
  uint16_t* s_id = (uint16_t*)(s_hi + blockDim.x);
  if (tid == 0) s_n = 0;
  stage_wave(a.clean, nclean, first_byte, cshift, s_stage, tid, nchunks, blockDim.x);
  __syncthreads();
  const Staged st = {s_stage, cshift};
  // lockstep levels: all of them, or the first hyp_main_levels -- what is still alive then goes to hyp_straggler_kernel
  const uint32_t l_main = (a.hyp_main_levels > 0 && a.hyp_main_levels < a.hyp_levels) ? (uint32_t)a.hyp_main_levels : (uint32_t)a.hyp_levels;
  uint32_t n_in = blockDim.x;  // level 1: every thread has its own path (lane, hypothesis = wave)
  uint32_t l = 1;
  for (; l <= l_main; l++) {
    bool alive = tid < n_in;
    uint32_t il = lane, h = tid >> 6;
    uint64_t s0 = 0;
    if (l > 1) {
      if (alive) {
        const uint32_t id = s_id[tid];
        il = id & 63u;
        h = id >> 6;
        s0 = (uint64_t)s_lo[tid] | ((uint64_t)s_hi[tid] << 32);
      }
      __syncthreads();  // every entry of the list is in registers ...
      if (tid == 0) s_n = 0;
      __syncthreads();  // ... before the survivors of this level are appended from its start
    }
    const uint32_t i = i0 + il, j = i + l;
    if (j >= nsub) alive = false;
    if (alive) {
      const uint32_t slot = (l - 1u) * H + h;
      if (l == 1) s0 = a.hyp_state[(size_t)i * kHuffHypSlots + h];
      uint32_t p = (uint32_t)s0, b = (uint32_t)(s0 >> 32) & 0xffu, k = (uint32_t)(s0 >> 40) & 0xffu;
      const uint32_t end_bit = min((j + 1) * a.sub_bits, nbits);
      uint32_t nblk = 0;
      const size_t at = (size_t)(j - 1) * kHuffHypSlots + slot;
      if (a.pieces > 1) {
        const size_t mo = at * (size_t)(a.pieces - 1);
        track_span_pieces(a, st, first_byte * 8u, L, p, b, k, j * a.sub_bits, end_bit, nblk, a.mid_state + mo, a.mid_cnt + mo);
      } else if (p < end_bit) {
        track_span_pair(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk);
      }
      const uint64_t e = pack_state(p, b, k);
      const uint64_t* row = a.hyp_state + (size_t)j * kHuffHypSlots;
      const uint8_t* prev = a.hyp_map + (size_t)(j - 1) * kHuffHypSlots;
      const uint32_t m_lo = j > i_last + 1u ? j - i_last : 1u;  // owners started at j - m <= i_last
      const uint32_t g = hyp_find_slot(row, prev, e, H, l, m_lo);
      a.hyp_cnt[at] = (uint16_t)nblk;
      if (g != 0xffu) {
        a.hyp_map[at] = (uint8_t)g;
        if (a.hyp_hist) atomicAdd(a.flags + 9 + min(l, 6u), 1u);
      } else if (l == (uint32_t)a.hyp_levels) {  // map stays 0xff: not merged within the budget
        atomicAdd(a.flags + 3, 1u);
      } else {
        const uint32_t nslot = l * H + h;
        a.hyp_state[(size_t)j * kHuffHypSlots + nslot] = e;
        a.hyp_map[at] = (uint8_t)nslot;
        const uint32_t w = atomicAdd(&s_n, 1u);  // survives: onto the list (its order does not matter)
        s_lo[w] = (uint32_t)e;
        s_hi[w] = (uint32_t)(e >> 32);
        s_id[w] = (uint16_t)(il | (h << 6));
      }
    }
    __threadfence_block();
    __syncthreads();
    n_in = s_n;
    if (n_in == 0) break;
  }
  if (n_in != 0 && l_main < (uint32_t)a.hyp_levels) {  // hand-off: (start subsequence, hypothesis | levels done << 8)
    __shared__ uint32_t s_base;
    if (tid == 0) s_base = atomicAdd(a.flags + kHuffFlagStragglers, n_in);
    __syncthreads();
    if (tid < n_in) {
      const uint32_t at = s_base + tid, id = s_id[tid];
      if (at < a.strag_cap) {
        a.strag_list[2 * at] = i0 + (id & 63u);
        a.strag_list[2 * at + 1] = (id >> 6) | (l_main << 8);
      }
    }
  }
}

// ---- passes 0 and 1 in ONE launch, for one lockstep level (hyp_main_levels == 1: sparse scans -- the gain map) (round 6) ----------------------
// A workgroup walks 64 subsequences' fresh paths (pass 0) and, from registers, the next subsequence of the first 63 of them (level 1 of pass 1):
// the fresh end states that level 1 compares with are this workgroup's own (LDS), the bytes are staged once, the tables once, and the launch
// boundary between the two passes -- on the critical chain of the decode -- is gone.  Workgroups therefore advance by 63 subsequences: the
// 64th fresh walk of one is the first of the next (1.6 % of the walks twice, same result).  More than one lockstep level would need fresh states of
// the NEXT workgroup's subsequences inside this kernel; those scans keep the two launches.  What is still alive after level 1 goes to the
// straggler waves exactly as from hyp_pass1q_kernel.  flags[kHuffFlagStragglers] is zero when this kernel starts (the decode's initial fill / start_over).
constexpr uint32_t kFusedStride = 63;
__global__ __launch_bounds__(1024) void hyp_pass01_kernel(const HuffSyncArgs a) {
  extern __shared__ uint32_t s_stage[];
  __shared__ PairLds L;
  __shared__ uint32_t s_n, s_base;
  load_pair_lds(a, L);
  const uint32_t tid = threadIdx.x, lane = tid & 63u, h = tid >> 6, H = (uint32_t)a.hyp_h;
  const uint32_t i0 = blockIdx.x * kFusedStride, i = i0 + lane;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const uint32_t cshift = 31u - (uint32_t)__builtin_clz(a.sub_bits >> 3);
  const uint32_t first_byte = i0 * (a.sub_bits >> 3);
  // behind the staged bytes: the fresh end states of the 64 subsequences (two words each), then the survivors' ids (lane | hypothesis << 6)
  const uint32_t stage_words = ((64u * ((a.sub_bits >> 3) + 4u) + 16u + 7u) >> 2) + 1u & ~1u;
  uint32_t* s_fresh_lo = s_stage + stage_words;
  uint32_t* s_fresh_hi = s_fresh_lo + 64u * H;
  uint16_t* s_id = (uint16_t*)(s_fresh_hi + 64u * H);
  if (tid == 0) s_n = 0;
  stage_wave(a.clean, nclean, first_byte, cshift, s_stage, tid, 64u, blockDim.x);
  if (a.zero_vec) {  // (behind the last global load, as in hyp_pass0_kernel)
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (uint32_t v = blockIdx.x * blockDim.x + tid; v < a.zero_vec; v += gridDim.x * blockDim.x) a.zero_ptr[v] = z;
  }
  __syncthreads();
  const Staged st = {s_stage, cshift};
  const bool have = i < nsub;
  uint32_t p = i * a.sub_bits, b = h, k = 0;
  if (have) {  // ---- pass 0: the fresh path (subsequence i, hypothesis h)
    uint32_t nblk = 0;
    const uint32_t end_bit = min((i + 1) * a.sub_bits, nbits);
    if (a.pieces > 1 && blockIdx.x == 0 && h == 0) {  // (see hyp_pass0_kernel: the wave of the stream's first subsequence notes its pieces)
      uint64_t m[3] = {0, 0, 0};
      uint16_t c16[3] = {0, 0, 0};
      track_span_pieces(a, st, first_byte * 8u, L, p, b, k, i * a.sub_bits, end_bit, nblk, m, c16);
      if (i == 0) {
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < 3; q++)
          if (q + 1 < a.pieces) { a.pend[q] = m[q]; a.pcnt[q] = c16[q]; sum += c16[q]; }
        a.pcnt[a.pieces - 1] = nblk - sum;
      }
    } else {
      track_span_pair(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk);
    }
    a.hyp_state[(size_t)i * kHuffHypSlots + h] = pack_state(p, b, k);  // (the 64th walk: also written, with the same value, by the next workgroup)
    if (i == 0 && h == 0) a.nblk[0] = nblk;
  }
  const uint64_t e0 = pack_state(p, b, k);
  s_fresh_lo[lane * H + h] = (uint32_t)e0;
  s_fresh_hi[lane * H + h] = (uint32_t)(e0 >> 32);
  __syncthreads();
  // ---- level 1: the same path through subsequence j = i + 1; merge targets: the fresh paths of j
  const uint32_t j = i + 1u;
  if (have && lane < kFusedStride && j < nsub) {
    const uint32_t end_bit = min((j + 1) * a.sub_bits, nbits);
    uint32_t nblk = 0;
    const size_t at = (size_t)i * kHuffHypSlots + h;  // link (j - 1, slot h)
    if (a.pieces > 1) {
      const size_t mo = at * (size_t)(a.pieces - 1);
      track_span_pieces(a, st, first_byte * 8u, L, p, b, k, j * a.sub_bits, end_bit, nblk, a.mid_state + mo, a.mid_cnt + mo);
    } else if (p < end_bit) {
      track_span_pair(a, st, first_byte * 8u, L, p, b, k, end_bit, nblk);
    }
    const uint64_t e = pack_state(p, b, k);
    uint32_t g = 0xffu;
    for (uint32_t t = 0; t < H; t++) {
      const uint64_t f = (uint64_t)s_fresh_lo[(lane + 1u) * H + t] | ((uint64_t)s_fresh_hi[(lane + 1u) * H + t] << 32);
      if (f == e && g == 0xffu) g = t;
    }
    a.hyp_cnt[at] = (uint16_t)nblk;
    if (g != 0xffu) {
      a.hyp_map[at] = (uint8_t)g;
      if (a.hyp_hist) atomicAdd(a.flags + 10, 1u);
    } else if (a.hyp_levels == 1) {  // map stays 0xff: not merged within the budget
      atomicAdd(a.flags + 3, 1u);
    } else {
      const uint32_t nslot = H + h;
      a.hyp_state[(size_t)j * kHuffHypSlots + nslot] = e;
      a.hyp_map[at] = (uint8_t)nslot;
      const uint32_t w = atomicAdd(&s_n, 1u);
      s_id[w] = (uint16_t)(lane | (h << 6));
    }
  }
  __syncthreads();
  const uint32_t n_in = s_n;
  if (n_in != 0) {  // hand-off: (start subsequence, hypothesis | levels done << 8)
    if (tid == 0) s_base = atomicAdd(a.flags + kHuffFlagStragglers, n_in);
    __syncthreads();
    if (tid < n_in) {
      const uint32_t at = s_base + tid, id = s_id[tid];
      if (at < a.strag_cap) {
        a.strag_list[2 * at] = i0 + (id & 63u);
        a.strag_list[2 * at + 1] = (id >> 6) | (1u << 8);
      }
    }
  }
}

// ---- pass 1, the stragglers (round 5): one WAVE per path -------------------------------------------------------------------
// A lane follows a path at ~460 cycles per symbol (two dependent LDS reads and ~25 dependent VALU instructions), so pass 1 in
// lockstep lasts as long as its unluckiest path: levels x (symbols per subsequence) x 460 cycles.  Few paths get that far (4K
// 4:2:0 q95: 8 % are alive after level 1, 0.5 % after level 2), so each of them gets a wave: lane o looks up the symbol that
// WOULD start at bit p + o of the stream in all four tracking tables (two packed registers: luma DC | AC, chroma DC | AC) --
// four LDS gathers for 64 bit positions at once -- and one scalar chain hops from boundary to boundary with v_readlane:
// off += bits(entry[off]), k += advance, block / MCU bookkeeping in SGPRs.  ~40 cycles per symbol plus ~300 per 64-bit window.
// Same state transitions as track_span (zeros past the end of the stream included); restart files do not come here.
// Merge targets: the fresh slots of the subsequence just finished and the in-flight slots the lockstep levels wrote (their
// owners' links are complete before this kernel starts); slots written by other stragglers are never looked at.
constexpr int kStragWaves = 4;               // waves per workgroup (they share one copy of the tracking tables)
constexpr int kStragStageWords = 132 + 4;    // a 4096-bit subsequence + the overhang of its last symbol, as big-endian words

__global__ __launch_bounds__(64 * kStragWaves) void hyp_straggler_kernel(const HuffSyncArgs a) {
  __shared__ ScanLds L;
  __shared__ uint32_t s_bytes[kStragWaves][kStragStageWords];
  load_scan_lds<true>(a, L);
  __syncthreads();
  // everything the scalar chain depends on is made wave-uniform explicitly (v_readfirstlane): the compiler cannot know that a
  // value loaded from memory or derived from threadIdx.x >> 6 is the same in all 64 lanes, and would keep the walker's state in
  // VGPRs behind exec-mask branches otherwise
  auto uni = [](uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); };
  const uint32_t lane = threadIdx.x & 63u, wv = uni(threadIdx.x >> 6);
  const uint32_t wave = blockIdx.x * kStragWaves + wv, nwaves = gridDim.x * kStragWaves;
  uint32_t count = uni(a.flags[kHuffFlagStragglers]);
  if (count > a.strag_cap) count = a.strag_cap;
  const uint32_t nclean = uni(a.nbytes - *a.nstuffed), nbits = nclean * 8u;
  const uint32_t nsub = (nbits + a.sub_bits - 1) / a.sub_bits;
  const uint32_t H = (uint32_t)a.hyp_h, bpm = (uint32_t)a.blocks_per_mcu;
  const uint32_t l_main = (uint32_t)a.hyp_main_levels;
  const uint32_t l_cap = (uint32_t)min(a.hyp_levels, a.strag_levels);
  uint32_t cpack = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) cpack |= ((uint32_t)a.comp_of[j] & 3u) << (2 * j);
  constexpr uint32_t kWords = sizeof(HuffFastTable) / 2;
  const uint16_t* T = (const uint16_t*)L.t;
  const uint32_t* clean32 = (const uint32_t*)a.clean;
  uint32_t* stage = s_bytes[wv];
  // One level's global reads do not depend on the path's state: the words of subsequence j (the path enters it at a bit p >= j * sub_bits and
  // leaves it less than a symbol beyond its end) and the merge candidates of j.  Both are issued one level ahead -- the words -- or at the head of
  // the level -- the candidates -- and consumed after the scalar chain, which used to wait ~2 us for each of them per level (round 6).
  const uint32_t nw = min((a.sub_bits >> 5) + 4u, (uint32_t)kStragStageWords);
  // (raw loads only, at a clamped index: anything done with the value here would put the wait for it here)
  const uint32_t last_word = nclean > 4u ? (nclean - 1u) >> 2 : 0u;
  auto load_words = [&](uint32_t jj, uint32_t (&v)[3]) {
    const uint32_t w0 = (jj * a.sub_bits) >> 5;
#pragma unroll
    for (uint32_t q = 0; q < 3; q++) v[q] = clean32[min(w0 + lane + 64u * q, last_word)];
  };
  auto stage_words = [&](uint32_t jj, const uint32_t (&v)[3]) {  // -> big-endian, zeros past the end of the stream
    const uint32_t w0 = (jj * a.sub_bits) >> 5;
#pragma unroll
    for (uint32_t q = 0; q < 3; q++) {
      const uint32_t d = lane + 64u * q, byte0 = (w0 + d) * 4u;
      uint32_t x = __builtin_bswap32(v[q]);
      if (byte0 + 4u > nclean) x = byte0 < nclean ? x & (0xffffffffu << (8u * (byte0 + 4u - nclean))) : 0u;
      if (d < nw) stage[d] = x;
    }
  };
  for (uint32_t idx = wave; idx < count; idx += nwaves) {
    const uint32_t i = uni(a.strag_list[2 * idx]), hw = uni(a.strag_list[2 * idx + 1]);
    const uint32_t h = hw & 0xffu, l0 = hw >> 8;
    uint32_t slot = l0 * H + h;
    const uint64_t s0 = a.hyp_state[(size_t)(i + l0) * kHuffHypSlots + slot];
    uint32_t nx[3];
    load_words(i + l0 + 1u, nx);
    uint32_t p = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)s0);
    uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(s0 >> 32) & 0xffu));
    uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(s0 >> 40) & 0xffu));
    for (uint32_t l = l0 + 1; l <= l_cap; l++) {
      const uint32_t j = i + l;
      if (j >= nsub) break;
      const uint32_t end_bit = min((j + 1) * a.sub_bits, nbits);
      uint32_t nblk = 0;
      const uint32_t w0 = (j * a.sub_bits) >> 5;
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // the previous level's reads of the stage are done
      stage_words(j, nx);
      // lane c looks at slot c: fresh (c < H), or in flight from a lockstep level m = c / H <= l_main, m < l, whose owner's link continued into it
      const uint64_t* row = a.hyp_state + (size_t)j * kHuffHypSlots;
      const uint8_t* prev = a.hyp_map + (size_t)(j - 1) * kHuffHypSlots;
      const bool fresh = lane < H, flight = !fresh && lane < kHuffHypSlots && lane < l * H && lane / H <= l_main;
      const uint32_t cl = min(lane, (uint32_t)kHuffHypSlots - 1u);
      const uint64_t cand = row[cl];
      const uint32_t owner = prev[max(cl, H) - H];
      load_words(j + 1u, nx);
      if (p < end_bit) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        // the write pass's pieces (HuffSyncArgs::pieces): the chain pauses at every interior cut and lane 0 notes the state and the blocks so far
        const uint32_t Q = (uint32_t)a.pieces, piece_bits = a.sub_bits / Q;
        const size_t mo = ((size_t)(j - 1) * kHuffHypSlots + slot) * (size_t)(Q - 1u);
        uint32_t blk_noted = 0;
        for (uint32_t pq = 0; pq < Q; pq++) {
        const uint32_t cut = pq + 1u < Q ? min(j * a.sub_bits + (pq + 1u) * piece_bits, end_bit) : end_bit;
        while (p < cut) {
          const uint32_t q = p + lane - (w0 << 5);  // the lane's bit, relative to the stage
          const uint32_t wi = min(q >> 5, (uint32_t)kStragStageWords - 2u);
          const uint64_t two = ((uint64_t)stage[wi] << 32) | stage[wi + 1];
          const uint32_t w16 = (uint32_t)(two >> (48u - (q & 31u))) & 0xffffu;
          uint32_t ent[4];
#pragma unroll
          for (uint32_t t = 0; t < 4; t++) {
            const uint32_t tb = t * kWords;
            uint32_t e = T[tb + (w16 >> 7)];
            const uint32_t e2 = T[tb + 512u + ((e & 0x8000u) ? (e & 31u) : 0u) * 128u + (w16 & 127u)];
            ent[t] = (e & 0x8000u) ? e2 : e;
          }
          const uint32_t lim = min(64u, cut - p);
          uint32_t off = 0;
          // The chain: per block one DC step, then AC steps until the zig-zag index passes 63 -- the inner loop is the whole cost
          // (24 of 25 symbols of a busy block): v_readlane, two field extractions, two additions, two compares.  The component's
          // entry registers are picked once per block, not per symbol.
          while (off < lim) {
            const bool is_chroma = ((cpack >> (2u * b)) & 3u) != 0;
            const int dc = (int)(is_chroma ? ent[2] : ent[0]), ac = (int)(is_chroma ? ent[3] : ent[1]);
            if (k == 0) {
              const uint32_t e = (uint32_t)__builtin_amdgcn_readlane(dc, (int)off);
              off += e & 31u;
              k = (e >> 5) & 127u;  // 1 for every DC symbol
            }
            while ((int32_t)((k - 64u) & (off - lim)) < 0) {  // k < 64 && off < lim (both small): one compare
              const uint32_t e = (uint32_t)__builtin_amdgcn_readlane(ac, (int)off);
              off += e & 31u;
              k += (e >> 5) & 127u;
            }
            if (k >= 64u) {
              k = 0;
              b++;
              nblk++;
              if (b == bpm) b = 0;
            }
          }
          p += off;
        }
        if (pq + 1u < Q && lane == 0) {
          a.mid_state[mo + pq] = pack_state(p, b, k);
          a.mid_cnt[mo + pq] = (uint16_t)(nblk - blk_noted);
        }
        blk_noted = nblk;
        }
      } else if (a.pieces > 1 && lane == 0) {  // already beyond the subsequence: every cut sees the same state, no block completes
        const size_t mo = ((size_t)(j - 1) * kHuffHypSlots + slot) * (size_t)(a.pieces - 1);
        for (int pq = 0; pq + 1 < a.pieces; pq++) {
          a.mid_state[mo + pq] = pack_state(p, b, k);
          a.mid_cnt[mo + pq] = 0;
        }
      }
      const uint64_t e = pack_state(p, b, k);
      const bool match = (fresh || (flight && owner == lane)) && cand == e;
      const uint64_t mm = __builtin_amdgcn_ballot_w64(match);
      const size_t at = (size_t)(j - 1) * kHuffHypSlots + slot;
      if (lane == 0) a.hyp_cnt[at] = (uint16_t)nblk;
      if (mm != 0) {
        if (lane == 0) {
          a.hyp_map[at] = (uint8_t)__builtin_ctzll(mm);
          if (a.hyp_hist) atomicAdd(a.flags + 22 + min(l - 2u, 9u), 1u);  // debug: the stragglers' merges per level 2 .. 10, 11+ in flags[22..31]
        }
        break;
      }
      if (l == l_cap) {  // map stays 0xff: not merged within the budget
        if (lane == 0) atomicAdd(a.flags + 3, 1u);
        break;
      }
      const uint32_t nslot = l * H + h;
      if (lane == 0) {
        a.hyp_state[(size_t)j * kHuffHypSlots + nslot] = e;
        a.hyp_map[at] = (uint8_t)nslot;
      }
      slot = nslot;
    }
  }
}

// chain (round 4 form).  256 links per tile, function composition by CHASING instead of a scan over whole maps:
//   tiles   the tile's 256 link maps (48 bytes each) are staged in LDS; thread (group g, slot s) follows slot s through the 16
//           links of group g -- 16 dependent byte reads -- which gives the 16 group maps; 48 threads follow their slot through the
//           16 group maps (the tile's total map, and the slot every group is entered in); then (g, s) walks its 16 links once
//           more from there and writes, link by link, "slot at the tile's entry -> slot in front of this link" (the prefix maps);
//   entry   one workgroup does the same over the tile maps, 256 tiles per pass, but only for the true path (slot 0 of
//           subsequence 0): the slot the true path enters every tile in;
//   walk    tile entry -> the thread's entry (one byte of its prefix map) -> its link: state[0][i] = true end state of
//           subsequence i, nblk[i + 1] = blocks completed inside subsequence i + 1.
// 3 x 16 dependent LDS reads per workgroup where the scan form (round 2: eight steps over 256 whole maps, 12 word reads + 48
// dependent byte lookups + 12 writes per thread and step) took 18 us per workgroup, and 39 us with its 32-way bank conflicts (four
// consecutive 48-byte rows per thread: banks (48 t + ...) mod 32).  LDS rows have a pitch of 13 words, so the rows of the
// groups a wave spans start in different banks; lanes of one group read one row (broadcast within a word).
constexpr int kChainThreads = 256, kChainTile = kChainThreads;     // the walk: one link per thread
constexpr int kChainGroup = 16, kChainGroups = kChainTile / kChainGroup;
constexpr int kChainWg = kChainGroups * kHuffHypSlots;            // 768 threads: (group, slot)
constexpr int kMapPitch = kHuffHypSlots + 4;
static_assert(kHuffHypSlots % 16 == 0 && ((kMapPitch / 4) & 1) == 1, "map rows: whole 16-byte words in memory, an odd word pitch in LDS");
static_assert(kChainTile * kHuffHypSlots / 16 == kChainWg, "one 16-byte piece of the tile per thread");

// rows [first, first + 256) of a [rows][48] byte array -> s_rows (identity maps beyond n)
__device__ __forceinline__ void chain_stage_rows(uint8_t (*s_rows)[kMapPitch], int tid, const uint8_t* __restrict__ rows, int first, int n) {
  const int row = tid / 3, part = tid - row * 3;
  uint4 v;
  if (first + row < n) {
    v = *(const uint4*)(rows + (size_t)(first + row) * kHuffHypSlots + part * 16);
  } else {
    const uint32_t b0 = (uint32_t)(16 * part) * 0x01010101u + 0x03020100u;
    v = make_uint4(b0, b0 + 0x04040404u, b0 + 0x08080808u, b0 + 0x0c0c0c0cu);
  }
  uint32_t* d = (uint32_t*)(s_rows[row] + part * 16);
  d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
}
__device__ __forceinline__ uint32_t chain_step(const uint8_t* row, uint32_t cur) { return cur == 0xffu ? 0xffu : (uint32_t)row[cur]; }

// tile_entry[t] = the slot the true path is in where tile t begins (0xff: lost before that): one workgroup's walk over the tile maps, 256
// tiles per pass.  s_tiles / s_gmap: LDS of the caller (the tiles kernel's own arrays, free by then).
__device__ __forceinline__ void chain_entry_walk(const HuffSyncArgs& a, const uint8_t* __restrict__ tile_map, uint8_t* __restrict__ tile_entry,
                                                 uint8_t (*s_tiles)[kMapPitch], uint8_t (*s_gmap)[kMapPitch], uint32_t* s_gentry /* kChainGroups + 1 */) {
  const int tid = (int)threadIdx.x;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const int nlinks = (int)((nbits + a.sub_bits - 1) / a.sub_bits) - 1;
  const int ntiles = (nlinks + kChainTile - 1) / kChainTile;
  const int grp = tid / kHuffHypSlots, s = tid - grp * kHuffHypSlots;
  uint32_t carry = 0;  // the true path is slot 0 of subsequence 0
  for (int c0 = 0; c0 < ntiles; c0 += kChainTile) {
    chain_stage_rows(s_tiles, tid, tile_map, c0, ntiles);
    __syncthreads();
    {
      uint32_t cur = (uint32_t)s;
#pragma unroll
      for (int k = 0; k < kChainGroup; k++) cur = chain_step(s_tiles[grp * kChainGroup + k], cur);
      s_gmap[grp][s] = (uint8_t)cur;
    }
    __syncthreads();
    if (tid == 0) {
      uint32_t c = carry;
      for (int g = 0; g < kChainGroups; g++) {
        s_gentry[g] = c;
        c = chain_step(s_gmap[g], c);
      }
      s_gentry[kChainGroups] = c;
    }
    __syncthreads();
    if (tid < kChainGroups) {
      uint32_t cur = s_gentry[tid];
      for (int k = 0; k < kChainGroup; k++) {
        const int tl = c0 + tid * kChainGroup + k;
        if (tl < ntiles) tile_entry[tl] = (uint8_t)cur;
        cur = chain_step(s_tiles[tid * kChainGroup + k], cur);
      }
    }
    carry = s_gentry[kChainGroups];
    __syncthreads();  // the rows are rewritten by the next pass
  }
}
// tiles: the prefix maps of every link and the tile's total map
__global__ __launch_bounds__(kChainWg) void hyp_chain_tiles_kernel(const HuffSyncArgs a, uint8_t* __restrict__ prefix /* [links][slots] */,
                                                                   uint8_t* __restrict__ tile_map /* [tiles][slots] */) {
  __shared__ __attribute__((aligned(16))) uint8_t s_links[kChainTile][kMapPitch];  // 13 KB
  __shared__ __attribute__((aligned(16))) uint8_t s_gmap[kChainGroups][kMapPitch];
  __shared__ __attribute__((aligned(16))) uint8_t s_gpre[kChainGroups][kMapPitch];
  const int tid = (int)threadIdx.x;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const int nsub = (int)((nbits + a.sub_bits - 1) / a.sub_bits);
  const int nlinks = nsub - 1;
  const int base = (int)blockIdx.x * kChainTile;
  if (base < nlinks) {  // (the grid is sized for the stuffed stream)
    chain_stage_rows(s_links, tid, a.hyp_map, base, nlinks);
    __syncthreads();
    const int grp = tid / kHuffHypSlots, s = tid - grp * kHuffHypSlots;
    {
      uint32_t cur = (uint32_t)s;
#pragma unroll
      for (int k = 0; k < kChainGroup; k++) cur = chain_step(s_links[grp * kChainGroup + k], cur);
      s_gmap[grp][s] = (uint8_t)cur;
    }
    __syncthreads();
    if (tid < kHuffHypSlots) {
      uint32_t c = (uint32_t)tid;
#pragma unroll
      for (int g = 0; g < kChainGroups; g++) {
        s_gpre[g][tid] = (uint8_t)c;
        c = chain_step(s_gmap[g], c);
      }
      tile_map[(size_t)blockIdx.x * kHuffHypSlots + tid] = (uint8_t)c;
    }
    __syncthreads();
    uint32_t cur = s_gpre[grp][s];
#pragma unroll
    for (int k = 0; k < kChainGroup; k++) {
      const int link = grp * kChainGroup + k, i = base + link;
      if (i < nlinks) prefix[(size_t)i * kHuffHypSlots + s] = (uint8_t)cur;
      cur = chain_step(s_links[link], cur);
    }
  }
}
// one workgroup: the walk over the tile maps
__global__ __launch_bounds__(kChainWg) void hyp_chain_entry_kernel(const HuffSyncArgs a, const uint8_t* __restrict__ tile_map, uint8_t* __restrict__ tile_entry) {
  __shared__ __attribute__((aligned(16))) uint8_t s_tiles[kChainTile][kMapPitch];
  __shared__ __attribute__((aligned(16))) uint8_t s_gmap[kChainGroups][kMapPitch];
  __shared__ uint32_t s_gentry[kChainGroups + 1];
  chain_entry_walk(a, tile_map, tile_entry, s_tiles, s_gmap, s_gentry);
}
__global__ __launch_bounds__(kChainThreads) void hyp_chain_walk_kernel(const HuffSyncArgs a, const uint8_t* __restrict__ prefix,
                                                                       const uint8_t* __restrict__ tile_entry) {
  const int tid = (int)threadIdx.x;
  const uint32_t nclean = a.nbytes - *a.nstuffed, nbits = nclean * 8u;
  const int nsub = (int)((nbits + a.sub_bits - 1) / a.sub_bits);
  const int nlinks = nsub - 1;
  const int i = (int)blockIdx.x * kChainTile + tid;
  bool lost = false;
  if (i < nlinks) {
    uint32_t g = tile_entry[blockIdx.x];
    if (g != 0xffu) g = prefix[(size_t)i * kHuffHypSlots + g];
    if (g == 0xffu) {
      lost = true;
    } else {
      const size_t at = (size_t)i * kHuffHypSlots + g;
      const uint64_t end_i = a.hyp_state[at];
      const uint32_t cnt_next = a.hyp_cnt[at];
      a.state[0][i] = end_i;
      a.nblk[i + 1] = cnt_next;
      if (a.pieces > 1) {  // (see HuffSyncArgs::pieces) piece ends of subsequence i + 1 from the notes of the path that crossed it, and i's own end
        const uint32_t Q = (uint32_t)a.pieces;
        a.pend[(size_t)i * Q + (Q - 1u)] = end_i;
        const size_t mo = at * (size_t)(Q - 1u);
        uint32_t sum = 0;
        for (uint32_t q = 0; q + 1u < Q; q++) {
          const uint32_t cq = a.mid_cnt[mo + q];
          a.pend[(size_t)(i + 1) * Q + q] = a.mid_state[mo + q];
          a.pcnt[(size_t)(i + 1) * Q + q] = cq;
          sum += cq;
        }
        a.pcnt[(size_t)(i + 1) * Q + (Q - 1u)] = cnt_next - sum;
      }
      if (i == nlinks - 1 && a.hyp_map[at] == 0xffu) lost = true;  // the last link must resolve too
    }
  }
  const uint64_t m = __builtin_amdgcn_ballot_w64(lost);
  if (m != 0 && (uint32_t)(tid & 63) == (uint32_t)__builtin_ctzll(m)) atomicOr(a.flags + 2, 1u);
}

// step 5: DC prediction = running sum of the differences over the component's blocks in scan order, as a three-kernel
// scan over chunks of 1024 scan positions (per-chunk sums per component -> scan of the chunk sums -> rescan + store).
__device__ __forceinline__ void dc_block_scan(int v[3], int* s_sum /* 3 x 16 */, int) {
  int x[3] = {v[0], v[1], v[2]}, all[3];
  wg_incl_scan<1024, 3>(x, s_sum, all);
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2];  // inclusive
}
__global__ __launch_bounds__(1024) void dc_partial_kernel(const HuffSyncArgs a, int* __restrict__ partial) {
  __shared__ int s_sum[3 * 16];
  const int tid = (int)threadIdx.x;
  const uint32_t t = blockIdx.x * 1024u + (uint32_t)tid;
  int v[3] = {0, 0, 0};
  if (t < a.total_blocks) v[a.comp_of[t % (uint32_t)a.blocks_per_mcu]] = a.dcd[t];
  dc_block_scan(v, s_sum, tid);
  if (tid == 1023) { partial[blockIdx.x * 3] = v[0]; partial[blockIdx.x * 3 + 1] = v[1]; partial[blockIdx.x * 3 + 2] = v[2]; }
}
// form 2 of the write pass: the DC differences sit at [0] of every scan-order block.  Chunks of kPlaceChunk scan positions
// (1024 left a 4K 4:2:0 frame with 190 workgroups for 256 CUs, each spending most of its time in a ten-step scan).
constexpr int kPlaceChunk = 256;
__device__ __forceinline__ void dc_chunk_scan(int v[3], int* s_sum /* 3 x kPlaceChunk / 64 */, int) {
  int x[3] = {v[0], v[1], v[2]}, all[3];
  wg_incl_scan<kPlaceChunk, 3>(x, s_sum, all);
  v[0] = x[0]; v[1] = x[1]; v[2] = x[2];  // inclusive
}
__global__ __launch_bounds__(kPlaceChunk) void dc_partial2_kernel(const HuffSyncArgs a, int* __restrict__ partial) {
  __shared__ int s_sum[3 * (kPlaceChunk / 64)];
  const int tid = (int)threadIdx.x;
  const uint32_t t = blockIdx.x * (uint32_t)kPlaceChunk + (uint32_t)tid;
  int v[3] = {0, 0, 0};
  if (t < a.total_blocks) v[a.comp_of[t % (uint32_t)a.blocks_per_mcu]] = a.coef_scan[(size_t)t * 64];
  dc_chunk_scan(v, s_sum, tid);
  if (tid == kPlaceChunk - 1) { partial[blockIdx.x * 3] = v[0]; partial[blockIdx.x * 3 + 1] = v[1]; partial[blockIdx.x * 3 + 2] = v[2]; }
}
// ... and the kernel that finishes the decode: kPlaceChunk scan positions per workgroup.  DC prediction (a workgroup scan of the
// differences on top of the chunk's carry-in), then every wave moves 64 of the chunk's blocks, two per step: a lane reads the
// two coefficients of one natural-order pair from the block's zig-zag scratch row (one 128-byte line per block) and the wave
// stores 2 x 128 contiguous bytes.  Dummy blocks of edge MCUs are dropped here.
// partial: the chunks' per-component DC sums -- scanned (exclusive) by dc_scan_partials_kernel when self_prefix == 0; else raw, every workgroup adding
// up the chunks in front of its own (round 6: one launch less; up to kPlaceSelfPrefix chunks).
constexpr int kPlaceSelfPrefix = 4096;
__global__ __launch_bounds__(kPlaceChunk) void coef_place_kernel(const HuffSyncArgs a, const int* __restrict__ partial, int self_prefix) {
  __shared__ int s_sum[3 * (kPlaceChunk / 64)];
  __shared__ int16_t s_dc[kPlaceChunk];
  __shared__ uint32_t s_dst[kPlaceChunk];  // component << 30 | JBLOCK index inside its array; ~0: a dummy block of an edge MCU
  __shared__ uint8_t s_inv[64];            // natural index -> zig-zag position
  const int tid = (int)threadIdx.x;
  const uint32_t t0 = blockIdx.x * (uint32_t)kPlaceChunk, t = t0 + (uint32_t)tid;
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  if (tid < 64) s_inv[a.zigzag[tid]] = (uint8_t)tid;  // zigzag[]: zig-zag position -> natural index
  int v[3] = {0, 0, 0};
  int c = 0;
  uint32_t where = 0xffffffffu;
  if (t < a.total_blocks) {
    const uint32_t m = t / bpm, j = t - m * bpm;
    c = a.comp_of[j];
    v[c] = a.coef_scan[(size_t)t * 64];
    const int my = (int)(m / (uint32_t)a.mcus_per_row), mx = (int)(m - (uint32_t)my * (uint32_t)a.mcus_per_row);
    const int jj = (int)j - a.first_blk[c];
    const int by = my * a.vs[c] + jj / a.hs[c], bx = mx * a.hs[c] + jj % a.hs[c];
    if (by < a.bh[c] && bx < a.bw[c]) where = ((uint32_t)c << 30) | ((uint32_t)by * (uint32_t)a.bw[c] + (uint32_t)bx);
  }
  s_dst[tid] = where;
  int carry[3];
  if (self_prefix) {
    int part[3] = {0, 0, 0};
    for (uint32_t i = (uint32_t)tid; i < blockIdx.x; i += (uint32_t)kPlaceChunk) {
      part[0] += partial[i * 3];
      part[1] += partial[i * 3 + 1];
      part[2] += partial[i * 3 + 2];
    }
    wg_incl_scan<kPlaceChunk, 3>(part, s_sum, carry);
  } else {
    carry[0] = partial[blockIdx.x * 3];
    carry[1] = partial[blockIdx.x * 3 + 1];
    carry[2] = partial[blockIdx.x * 3 + 2];
  }
  dc_chunk_scan(v, s_sum, tid);
  s_dc[tid] = (int16_t)((c == 0 ? carry[0] : (c == 1 ? carry[1] : carry[2])) + v[c]);
  __syncthreads();
  const uint32_t lane = (uint32_t)tid & 63u, wv = (uint32_t)tid >> 6;
  const uint32_t half = lane >> 5, n0 = (lane & 31u) * 2u;  // the lane's natural-order pair inside its block
  const uint32_t z0 = s_inv[n0], z1 = s_inv[n0 + 1u];
  int16_t* const c0 = a.coef[0];
  int16_t* const c1 = a.coef[1];
  int16_t* const c2 = a.coef[2];
#pragma unroll 4
  for (uint32_t s = 0; s < 32u; s++) {
    const uint32_t loc = wv * 64u + s * 2u + half;
    const uint32_t ent = s_dst[loc];
    if (ent == 0xffffffffu) continue;
    const int16_t* src = a.coef_scan + (size_t)(t0 + loc) * 64;
    uint32_t lo = (uint16_t)src[z0];
    const uint32_t hi = (uint16_t)src[z1];
    if (n0 == 0) lo = (uint16_t)s_dc[loc];
    const uint32_t cc = ent >> 30;
    int16_t* base = cc == 0 ? c0 : (cc == 1 ? c1 : c2);
    uint32_t* dst = (uint32_t*)(base + (size_t)(ent & 0x3fffffffu) * 64);
    dst[lane & 31u] = lo | (hi << 16);
  }
}
__global__ __launch_bounds__(1024) void dc_scan_partials_kernel(int* __restrict__ partial, int nchunks) {  // exclusive, in place
  __shared__ int s_sum[3 * 16];
  const int tid = (int)threadIdx.x;
  const int per = (nchunks + 1023) / 1024, lo = min(tid * per, nchunks), hi = min(lo + per, nchunks);
  int v[3] = {0, 0, 0};
  for (int i = lo; i < hi; i++)
    for (int c = 0; c < 3; c++) v[c] += partial[i * 3 + c];
  int own[3] = {v[0], v[1], v[2]};
  dc_block_scan(v, s_sum, tid);
  int run[3] = {v[0] - own[0], v[1] - own[1], v[2] - own[2]};
  for (int i = lo; i < hi; i++)
    for (int c = 0; c < 3; c++) {
      const int x = partial[i * 3 + c];
      partial[i * 3 + c] = run[c];
      run[c] += x;
    }
}
__global__ __launch_bounds__(1024) void dc_apply_kernel(const HuffSyncArgs a, const int* __restrict__ partial) {
  __shared__ int s_sum[3 * 16];
  const int tid = (int)threadIdx.x;
  const uint32_t t = blockIdx.x * 1024u + (uint32_t)tid;
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  int v[3] = {0, 0, 0};
  int c = 0;
  uint32_t m = 0, j = 0;
  if (t < a.total_blocks) {
    m = t / bpm;
    j = t - m * bpm;
    c = a.comp_of[j];
    v[c] = a.dcd[t];
  }
  dc_block_scan(v, s_sum, tid);
  if (t >= a.total_blocks) return;
  const int dc = partial[blockIdx.x * 3 + c] + v[c];
  if (a.rst_map) {
    // restart intervals: the prediction starts again at zero every rst_blocks blocks.  The running sums are kept as they
    // are (dcd[t] <- the component's sum up to t); the last block of an interval also publishes all three sums, and
    // dc_restart_kernel subtracts what had accumulated before the block's own interval
    a.dcd[t] = dc;
    if ((t + 1u) % a.rst_blocks == 0 && t + 1u < a.total_blocks) {
      int* g = a.dc_seg + (size_t)((t + 1u) / a.rst_blocks) * 3;
      for (int cc = 0; cc < 3; cc++) g[cc] = partial[blockIdx.x * 3 + cc] + v[cc];
    }
    return;
  }
  const int my = (int)(m / (uint32_t)a.mcus_per_row), mx = (int)(m - (uint32_t)my * (uint32_t)a.mcus_per_row);
  const int jj = (int)j - a.first_blk[c];
  const int by = my * a.vs[c] + jj / a.hs[c], bx = mx * a.hs[c] + jj % a.hs[c];
  if (by < a.bh[c] && bx < a.bw[c]) a.coef[c][((size_t)by * a.bw[c] + bx) * 64] = (int16_t)dc;
}
__global__ __launch_bounds__(256) void dc_restart_kernel(const HuffSyncArgs a) {
  if (blockIdx.x == 0) {  // and, on the side: the unstuff pass's per-chunk marker statistics -> flags[9], [16], [17]
    __shared__ uint32_t s_acc[3];
    if (threadIdx.x < 3) s_acc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t v[3] = {0, 0, 0};
    for (uint32_t i = threadIdx.x; i < a.rst_chunks; i += 256u)
      for (int k = 0; k < 3; k++) v[k] += a.rst_partial[i * 3 + k];
    for (int k = 0; k < 3; k++)
      if (v[k]) atomicAdd(&s_acc[k], v[k]);
    __syncthreads();
    if (threadIdx.x == 0) { a.flags[9] = s_acc[0]; a.flags[16] = s_acc[1]; a.flags[17] = s_acc[2]; }
  }
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.total_blocks) return;
  const uint32_t bpm = (uint32_t)a.blocks_per_mcu;
  const uint32_t m = t / bpm, j = t - m * bpm, seg = t / a.rst_blocks;
  const int c = a.comp_of[j];
  const int dc = a.dcd[t] - (seg ? a.dc_seg[(size_t)seg * 3 + c] : 0);
  const int my = (int)(m / (uint32_t)a.mcus_per_row), mx = (int)(m - (uint32_t)my * (uint32_t)a.mcus_per_row);
  const int jj = (int)j - a.first_blk[c];
  const int by = my * a.vs[c] + jj / a.hs[c], bx = mx * a.hs[c] + jj % a.hs[c];
  if (by < a.bh[c] && bx < a.bw[c]) a.coef[c][((size_t)by * a.bw[c] + bx) * 64] = (int16_t)dc;
}

}  // namespace

// the write pass: up to four waves share the 41 KB of value tables (static LDS) next to their staged bytes (dynamic);
// beyond 64 KB in total the dynamic part needs the opt-in (gfx950 gives a workgroup up to 160 KB)
static void launch_write(const HuffSyncArgs& a, uint32_t nsub, int final_buf, hipStream_t s) {
  const size_t per_wave = (size_t)64 * ((a.sub_bits >> 3) + 4) + 64;
  const int waves = per_wave * 4 <= (20u << 10) ? 4 : (per_wave * 2 <= (20u << 10) ? 2 : 1);
  const size_t lds = per_wave * waves;
  const int threads = 64 * waves;
  const int grid = (int)((nsub + threads - 1) / threads);
  if (lds > (20u << 10)) (void)hipFuncSetAttribute((const void*)sync_write_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(sync_write_kernel, dim3(grid), dim3(threads), lds, s, a, final_buf);
}

int huff_sync_chunks(uint64_t nbytes) { return (int)((nbytes + kChunk - 1) / kChunk); }

namespace {
__global__ __launch_bounds__(256) void stray_marker_kernel(const uint8_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ flag) {
  const uint32_t base = (blockIdx.x * 256u + threadIdx.x) * 16u;
  if (base >= n) return;
  bool found = false;
  if (base + 17u <= n && ((uintptr_t)(data + base) & 15u) == 0) {
    const uint4 v = *(const uint4*)(data + base);
    const uint32_t w[5] = {v.x, v.y, v.z, v.w, (uint32_t)data[base + 16u]};
#pragma unroll
    for (uint32_t i = 0; i < 16; i++) {
      const uint32_t c0 = (w[i >> 2] >> (8u * (i & 3u))) & 0xffu, c1 = (w[(i + 1u) >> 2] >> (8u * ((i + 1u) & 3u))) & 0xffu;
      found = found || (c0 == 0xffu && c1 != 0u && c1 != 0xffu && (c1 & 0xf8u) != 0xd0u);
    }
  } else {
    for (uint32_t i = base; i < base + 16u && i + 1u < n; i++) {
      const uint32_t c0 = data[i], c1 = data[i + 1u];
      found = found || (c0 == 0xffu && c1 != 0u && c1 != 0xffu && (c1 & 0xf8u) != 0xd0u);
    }
  }
  if (found) *(volatile uint32_t*)flag = 1u;  // a plain store: the word may be pinned host memory (no PCIe atomics needed)
}
}  // namespace
hipError_t launch_stray_marker_check(const uint8_t* data, uint32_t nbytes, uint32_t* flag, hipStream_t s) {
  const int grid = (int)((nbytes + 4095u) / 4096u);
  hipLaunchKernelGGL(stray_marker_kernel, dim3(grid), dim3(256), 0, s, data, nbytes, flag);
  return hipGetLastError();
}

// Step 1 (unstuff): chunk_counts becomes the exclusive scan, *nstuffed_dev the number of dropped bytes.
// rst_map != nullptr: the stream has restart markers; they are dropped as well, rst_map (zero-initialised, one bit per byte)
// gets the interval starts and rst_partial[chunk * 3 ..] the markers' count and sequence sums.
hipError_t launch_huffman_unstuff(const uint8_t* data, uint32_t nbytes, uint32_t* chunk_counts, uint32_t* nstuffed_dev, uint8_t* clean, hipStream_t s,
                                  uint32_t* rst_map, uint32_t* rst_partial, const HuffInitFill* fill) {
  const int nchunks = huff_sync_chunks(nbytes);
  HuffInitFill f;
  memset(&f, 0, sizeof f);
  if (fill) f = *fill;
  if (rst_map) hipLaunchKernelGGL(unstuff_count_kernel<true>, dim3(nchunks), dim3(256), 0, s, data, nbytes, chunk_counts, f);
  else hipLaunchKernelGGL(unstuff_count_kernel<false>, dim3(nchunks), dim3(256), 0, s, data, nbytes, chunk_counts, f);
  uint32_t* total = nstuffed_dev;
  if (nchunks > kUnstuffSelfPrefix) {
    hipLaunchKernelGGL(sync_scan_kernel, dim3(1), dim3(1024), 0, s, chunk_counts, nchunks, nstuffed_dev);
    total = nullptr;
  }
  if (rst_map) hipLaunchKernelGGL(unstuff_compact_kernel<true>, dim3(nchunks), dim3(256), 0, s, data, nbytes, (const uint32_t*)chunk_counts, clean, rst_map, rst_partial, total);
  else hipLaunchKernelGGL(unstuff_compact_kernel<false>, dim3(nchunks), dim3(256), 0, s, data, nbytes, (const uint32_t*)chunk_counts, clean, rst_map, rst_partial, total);
  return hipGetLastError();
}
static void launch_write2(const HuffSyncArgs& a, uint32_t nsub, int final_buf, hipStream_t s) {
  const size_t per_wave = (size_t)64 * ((a.sub_bits >> 3) + 4) + 64;
  const int waves = per_wave * 4 <= (20u << 10) ? 4 : (per_wave * 2 <= (20u << 10) ? 2 : 1);
  const size_t lds = sizeof(Write2Lds) + per_wave * waves;  // 48 KB of pair tables + the staged bytes: beyond the 64 KB default, opt in
  const int threads = 64 * waves;
  const int grid = (int)((nsub + threads - 1) / threads);
  static size_t opted = 0;
  if (lds > opted) {
    (void)hipFuncSetAttribute((const void*)sync_write2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    opted = lds;
  }
  hipLaunchKernelGGL(sync_write2_kernel, dim3(grid), dim3(threads), lds, s, a, final_buf);
}
static void launch_place(const HuffSyncArgs& a, int* dc_partial, hipStream_t s) {
  const int nch = (int)((a.total_blocks + kPlaceChunk - 1) / kPlaceChunk);
  hipLaunchKernelGGL(dc_partial2_kernel, dim3(nch), dim3(kPlaceChunk), 0, s, a, dc_partial);
  const int self_prefix = nch <= kPlaceSelfPrefix ? 1 : 0;
  if (!self_prefix) hipLaunchKernelGGL(dc_scan_partials_kernel, dim3(1), dim3(1024), 0, s, dc_partial, nch);
  hipLaunchKernelGGL(coef_place_kernel, dim3(nch), dim3(kPlaceChunk), 0, s, a, (const int*)dc_partial, self_prefix);
}
int huff_place_chunk() { return kPlaceChunk; }
static void launch_dc(const HuffSyncArgs& a, int* dc_partial, hipStream_t s) {
  const int nch = (int)((a.total_blocks + 1023) / 1024);
  hipLaunchKernelGGL(dc_partial_kernel, dim3(nch), dim3(1024), 0, s, a, dc_partial);
  hipLaunchKernelGGL(dc_scan_partials_kernel, dim3(1), dim3(1024), 0, s, dc_partial, nch);
  hipLaunchKernelGGL(dc_apply_kernel, dim3(nch), dim3(1024), 0, s, a, (const int*)dc_partial);
  if (a.rst_map) hipLaunchKernelGGL(dc_restart_kernel, dim3((a.total_blocks + 255) / 256), dim3(256), 0, s, a);
}

uint32_t huff_sync_max_subsequences(uint64_t nbytes, uint32_t sub_bits) { return (uint32_t)((nbytes * 8 + sub_bits - 1) / sub_bits); }

// Steps 2-5 on the clean stream.  Its size lives on the device (nbytes - *nstuffed): the grids are sized for the stuffed
// size, surplus lanes exit.  max_rounds bounds step 3; flags[4 + max_rounds % 3] != 0 afterwards means "not yet at the
// fixed point" and the caller falls back to the serial decoder.  nblk[], dcd[] and flags[] must be zero-initialised;
// dc_partial: 3 * ceil(total_blocks / 1024) ints of scratch.
hipError_t launch_huffman_decode_sync(const HuffSyncArgs& a, int max_rounds, int* dc_partial, int* final_buf, hipStream_t s) {
  const uint32_t nsub = huff_sync_max_subsequences(a.nbytes, a.sub_bits);
  const int grid = (int)((nsub + kSyncBlock - 1) / kSyncBlock);
  const size_t lds = (size_t)kSyncBlock * ((a.sub_bits >> 3) + 4) + 64;  // 64 padded chunks + the 16-byte overhang
  for (int r = 0; r <= max_rounds; r++) hipLaunchKernelGGL(sync_round_kernel, dim3(grid), dim3(kSyncBlock), lds, s, a, r);
  // Every executed round writes a complete state buffer; a round that finds its predecessor unchanged returns at once.
  // At the fixed point the two buffers are equal, so either one is final.
  *final_buf = 0;
  {
    const int nt = (int)((nsub + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(nt), dim3(kScanThreads), 0, s, a.nblk, (int)nsub, a.scan_tmp);  // (the write pass adds the tile sums: tile_offset)
  }
  if (a.coef_scan && !a.rst_map) {
    launch_write2(a, nsub, *final_buf, s);
    launch_place(a, dc_partial, s);
  } else {
    launch_write(a, nsub, *final_buf, s);
    launch_dc(a, dc_partial, s);
  }
  return hipGetLastError();
}

size_t huff_hyp_chain_bytes(uint64_t nbytes, uint32_t sub_bits, size_t* tiles_offset) {
  const size_t nsub = huff_sync_max_subsequences(nbytes, sub_bits), ntiles = (nsub + kChainTile - 1) / kChainTile;
  const size_t pre = ntiles * kChainThreads * kHuffHypSlots;  // the prefix maps; then the tile maps; then one entry byte per tile
  if (tiles_offset) *tiles_offset = pre;
  return pre + ntiles * kHuffHypSlots + ((ntiles + 255) & ~(size_t)255);
}

// The hypothesis scheme (see above) in place of the rounds.  hyp_map must be filled with 0xff, hyp_cnt / nblk / dcd / flags
// with zeros; chain_prefix: huff_hyp_chain_bytes(nbytes, sub_bits) bytes of scratch, chain_tiles follows it.  flags[2] != 0 afterwards: the true path did not merge somewhere; run launch_huffman_decode_sync instead.
hipError_t launch_huffman_decode_hyp(const HuffSyncArgs& a, int* dc_partial, uint8_t* chain_prefix, uint8_t* chain_tiles, hipStream_t s) {
  const uint32_t nsub = huff_sync_max_subsequences(a.nbytes, a.sub_bits);
  const int grid = (int)((nsub + 63) / 64);
  const int threads = 64 * a.hyp_h;
  const size_t lds0 = (size_t)64 * ((a.sub_bits >> 3) + 4) + 64;
  // pass 1: 64 + levels staged chunks (+ the 16-byte overhang), then the survivor list: 10 bytes per thread
  const size_t lds1_stage = ((((size_t)(64 + a.hyp_levels) * ((a.sub_bits >> 3) + 4) + 16 + 7) >> 2) + 1 & ~(size_t)1) * 4;
  const size_t lds1 = lds1_stage + (size_t)threads * 10 + 16;
  {
    // static LDS of the pass-1 kernels (tables + scan layout) + this must fit the 64 KB a workgroup gets without opting in
    // (4096 bits x 15 levels: 63.6 KB); beyond that the opt-in (gfx950: up to 160 KB per workgroup)
    static size_t opted = 0;
    if (lds1 + sizeof(PairLds) + 64 > (64u << 10) && lds1 > opted) {
      const hipError_t e1 = hipFuncSetAttribute((const void*)hyp_pass1q_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
      if (e1 != hipSuccess) return e1;
      opted = lds1;
    }
  }
  // UHDR_HIP_HUFF_DEBUG: per-kernel times on stderr (events between the launches)
  const bool dbg = getenv("UHDR_HIP_HUFF_DEBUG") != nullptr;
  hipEvent_t ev[9];
  int nev = 0;
  auto mark = [&]() {
    if (dbg && nev < 9) {
      (void)hipEventCreate(&ev[nev]);
      (void)hipEventRecord(ev[nev], s);
      nev++;
    }
  };
  mark();
  static const bool qmerge = !(getenv("UHDR_HIP_HUFF_QMERGE") && atoi(getenv("UHDR_HIP_HUFF_QMERGE")) == 0);
  static const bool no_fused = getenv("UHDR_HIP_HUFF_NO_FUSED_PASSES") != nullptr;
  const size_t lds01 = (((((size_t)64 * ((a.sub_bits >> 3) + 4) + 16 + 7) >> 2) + 1) & ~(size_t)1) * 4 + (size_t)64 * a.hyp_h * 8 + (size_t)threads * 2 + 16;
  if (qmerge && !no_fused && a.hyp_main_levels == 1 && a.hyp_levels > 1 && !a.rst_map && lds01 + sizeof(PairLds) + 64 <= (64u << 10)) {
    // one lockstep level: passes 0 and 1 in one launch (hyp_pass01_kernel)
    hipLaunchKernelGGL(hyp_pass01_kernel, dim3((nsub + kFusedStride - 1) / kFusedStride), dim3(threads), lds01, s, a);
    mark();
  } else {
    hipLaunchKernelGGL(hyp_pass0_kernel, dim3(grid), dim3(threads), lds0, s, a);
    mark();
    if (qmerge) hipLaunchKernelGGL(hyp_pass1q_kernel, dim3(grid), dim3(threads), lds1, s, a);
    else hipLaunchKernelGGL(hyp_pass1_kernel, dim3(grid), dim3(threads), lds1, s, a);
  }
  {
    const hipError_t e1 = hipGetLastError();  // a launch that fails (LDS beyond the limit) must not read as a "lost" true path
    if (e1 != hipSuccess) return e1;
  }
  if (qmerge && a.hyp_main_levels > 0 && a.hyp_main_levels < a.hyp_levels) {
    // the paths still alive after the lockstep levels, one wave each; the grid is sized for a thick tail (a wave takes the
    // entries wave, wave + nwaves, ...), surplus waves leave at once
    static const int smul = [] { const char* e = getenv("UHDR_HIP_HUFF_STRAG_MUL"); const int v = e ? atoi(e) : 1; return v >= 1 && v <= 16 ? v : 1; }();
    int sgrid = (int)((nsub * (uint32_t)a.hyp_h / 16u + kStragWaves - 1) / kStragWaves) * smul;
    if (sgrid < 64) sgrid = 64;
    if (sgrid > 4096) sgrid = 4096;
    hipLaunchKernelGGL(hyp_straggler_kernel, dim3(sgrid), dim3(64 * kStragWaves), 0, s, a);
  }
  mark();
  const int ntiles = (int)((nsub + kChainTile - 1) / kChainTile);
  uint8_t* chain_entry = chain_tiles + (size_t)ntiles * kHuffHypSlots;
  hipLaunchKernelGGL(hyp_chain_tiles_kernel, dim3(ntiles), dim3(kChainWg), 0, s, a, chain_prefix, chain_tiles);
  hipLaunchKernelGGL(hyp_chain_entry_kernel, dim3(1), dim3(kChainWg), 0, s, a, (const uint8_t*)chain_tiles, chain_entry);
  hipLaunchKernelGGL(hyp_chain_walk_kernel, dim3(ntiles), dim3(kChainThreads), 0, s, a, (const uint8_t*)chain_prefix, (const uint8_t*)chain_entry);
  mark();
  // the write pass: on pieces of a subsequence when the tracking passes noted them (HuffSyncArgs::pieces), else on whole subsequences
  HuffSyncArgs w = a;
  uint32_t nw = nsub;
  if (a.pieces > 1 && a.coef_scan && !a.rst_map) {
    w.sub_bits = a.sub_bits / (uint32_t)a.pieces;
    w.state[0] = a.pend;
    w.nblk = a.pcnt;
    nw = nsub * (uint32_t)a.pieces;
  }
  {
    const int nt = (int)((nw + kScanTile - 1) / kScanTile);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(nt), dim3(kScanThreads), 0, s, w.nblk, (int)nw, w.scan_tmp);  // (the write pass adds the tile sums: tile_offset)
  }
  mark();
  if (a.coef_scan && !a.rst_map) {
    launch_write2(w, nw, 0, s);
    mark();
    launch_place(a, dc_partial, s);
  } else {
    launch_write(a, nsub, 0, s);
    mark();
    launch_dc(a, dc_partial, s);
  }
  mark();
  if (dbg) {
    (void)hipStreamSynchronize(s);
    static const char* names[] = {"pass0", "pass1 (+ stragglers)", "chain x3", "nblk scan", "write", "dc x3 / place", "-"};
    fprintf(stderr, "uhdr_hip: hypothesis decode kernels:");
    for (int i = 0; i + 1 < nev; i++) {
      float ms = 0;
      (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
      fprintf(stderr, " %s %.0f us,", names[i], ms * 1e3f);
    }
    fprintf(stderr, "\n");
    for (int i = 0; i < nev; i++) (void)hipEventDestroy(ev[i]);
  }
  return hipGetLastError();
}

}  // namespace uhdr
