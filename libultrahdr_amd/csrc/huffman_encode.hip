// Baseline Huffman entropy coding on gfx950 (the step after fdct_quant).
//
// In the reference this is libjpeg behind JpegEncoderHelper::compressImage
// (/root/reference/lib/src/jpegencoderhelper.cpp:131-244): one sequential pass over all MCUs with the Annex K
// tables, DC prediction chaining every block to its predecessor, no restart markers -- inherently serial.
// JPEG's own mechanism for cutting that chain is the restart interval (ITU-T T.81 B.2.4.4 / E.1.4): every
// `restart_interval` MCUs the bit stream is padded to a byte boundary, an RSTn marker is written and the DC
// predictors return to zero, so the intervals are independent.  That changes the bytes (a DRI segment and the RSTn
// markers) but not a single decoded coefficient -- the parity policy here is: for a given restart interval the
// entropy-coded segment equals, byte for byte, what libjpeg emits for the same coefficients with the same
// cinfo.restart_interval (the oracle's restatement of jchuff.c, itself pinned against the reference encoder at
// restart_interval 0 and through libjpeg's decoder at all others).
//
// Mapping: one wavefront = one restart interval, one lane = one 8x8 block (<= 64 blocks per interval).
//   1. the lane stages its block's 64 coefficients in LDS (rows of 33 words: conflict-free zig-zag reads) and the
//      interval's DC values are exchanged through LDS (prediction, libjpeg's dummy-block rule at the image edges);
//   2. a first walk over the block in zig-zag order adds up the code lengths (DC size category, run/size symbols,
//      ZRL, EOB); a wave prefix sum turns them into bit offsets inside the interval;
//   3. a second walk emits the bits into the interval's LDS bit buffer (MSB first; words are OR-ed in, the first and
//      last word of a lane are shared with its neighbours);
//   4. the buffer is padded with one bits to a byte boundary, byte-stuffed (0xFF -> 0xFF 0x00, positions from a wave
//      prefix sum of the 0xFF counts) and written to the interval's slot in device memory.
// Two launches of that kernel (size classes of the LDS bit buffer, see below), then a tiny kernel turns the interval
// sizes into offsets and a gather kernel copies the slots into the final stream with the RSTn markers in between.
#include "lds_copy.h"
#include "wg_scan.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kSegBlocks = 64;       // blocks per restart interval = lanes
constexpr int kWordsPerBlock = 52;   // 1664 bits >= the worst case of a baseline block (11 + 11 + 63 * (16 + 10) = 1660)
constexpr int kCoefRow = 33;         // LDS words per staged block (64 halfwords + 1 word of padding)

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, uint32_t lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)v, d, 64);
    if (lane >= (uint32_t)d) v += y;
  }
  return v;
}

// Round 5: the walk visits the NON-ZERO coefficients only.  A lane used to step through all 63 AC positions of its block (a
// zig-zag lookup and a coefficient read from LDS per step, twice: once for the lengths, once to emit) although a busy 4:2:0 q95
// block has ~25 of them non-zero and a gain-map block three.  The wave first builds every block's 64-bit occupancy map in
// zig-zag order cooperatively -- lane L reads zig-zag position L of block b, one v_cmp gives the whole map of block b as a wave
// mask (64 x ~6 instructions for 64 blocks) -- and the walk then hops from set bit to set bit (run = distance - 1).
// s_coef: the wave's staged blocks, kCoefRow words per block; zz_of_lane = natural index of zig-zag position `lane`.
__device__ __forceinline__ uint64_t zigzag_nonzero_map(const uint32_t* s_coef, uint32_t zz_of_lane, uint32_t lane) {
  uint64_t mine = 0;
  const uint32_t word = zz_of_lane >> 1, sh = (zz_of_lane & 1u) * 16u;
#pragma unroll 4
  for (uint32_t b = 0; b < (uint32_t)kSegBlocks; b++) {
    const uint32_t v = (s_coef[b * kCoefRow + word] >> sh) & 0xffffu;
    const uint64_t m = __builtin_amdgcn_ballot_w64(v != 0);
    if (lane == b) mine = m;
  }
  return mine;
}
// jchuff.c encode_one_block as a walk that reports every (code, length) pair to `put`; nz: the block's occupancy map (bit k =
// zig-zag position k holds a non-zero coefficient)
template <typename Put>
__device__ __forceinline__ void walk_block(const uint32_t* dct, const uint32_t* act, int dc_diff, bool real, const uint32_t* coef_row,
                                           const uint8_t* zz, uint64_t nz, uint32_t& out_of_range, Put put) {
  int temp = dc_diff, temp2 = dc_diff;
  if (temp < 0) { temp = -temp; temp2--; }
  uint32_t nbits = 32u - (uint32_t)__clz(temp);  // 0 for temp == 0
  out_of_range |= nbits > 11u;                   // jchuff.c: ERREXIT(JERR_BAD_DCT_COEF) beyond MAX_COEF_BITS + 1
  uint32_t e = dct[nbits & 15u];
  put(e & 0xffffu, e >> 16);
  if (nbits) put((uint32_t)temp2 & ((1u << nbits) - 1u), nbits);
  uint32_t prev = 0;  // zig-zag position of the last coefficient coded (0: the DC term)
  if (real) {
    uint64_t m = nz & ~1ull;
    while (m != 0) {
      const uint32_t k = (uint32_t)__builtin_ctzll(m);
      m &= m - 1ull;
      uint32_t r = k - prev - 1u;  // zeros since the last coefficient
      prev = k;
      const uint32_t idx = zz[k];
      const uint32_t wd = coef_row[idx >> 1];
      temp = (idx & 1u) ? ((int)wd >> 16) : (int)(int16_t)(wd & 0xffffu);
      while (r > 15) {
        e = act[0xf0];
        put(e & 0xffffu, e >> 16);
        r -= 16;
      }
      temp2 = temp;
      if (temp < 0) { temp = -temp; temp2--; }
      nbits = 32u - (uint32_t)__clz(temp);
      out_of_range |= nbits > 10u;  // MAX_COEF_BITS
      e = act[((r << 4) + nbits) & 255u];
      put(e & 0xffffu, e >> 16);
      put((uint32_t)temp2 & ((1u << (nbits & 31u)) - 1u), nbits);
    }
  }
  if (prev < 63u) {  // trailing zeros (a dummy block: all 63 AC terms): EOB
    e = act[0];
    put(e & 0xffffu, e >> 16);
  }
}

// Size classes: almost every interval of a real image needs a small fraction of the worst-case bit buffer, and LDS per
// wavefront decides how many intervals a CU encodes at once.  The first launch (WORDS = kWordsSmall: 512 bits per
// block on average, 4 KB) encodes every interval that fits and marks the others 0xFFFFFFFE; the second launch
// (RETRY, worst-case buffer) only picks those up.
constexpr int kWordsSmall = 16;
constexpr uint32_t kRetry = 0xFFFFFFFEu, kBadCoef = 0xFFFFFFFFu;

// Short intervals: a wavefront encodes G = 64 / (blocks per interval) intervals side by side (first launch only) -- at a
// restart interval of 2 MCUs (12 blocks at 4:2:0) five intervals share a wave instead of leaving 52 lanes idle.  The
// groups only differ in where their bits go: every group owns an equal slice of the LDS bit buffer, bit offsets come from
// one wave-wide scan minus the scan value in front of the group, and the pad / stuff / store steps run once per group.
template <int WORDS, bool RETRY>
__global__ __launch_bounds__(64) void huff_encode_kernel(const HuffArgs a) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[2 * (16 + 256)];
  __shared__ uint32_t s_coef[kSegBlocks * kCoefRow];
  __shared__ uint32_t s_bits[kSegBlocks * WORDS + 2];
  __shared__ int s_dc[kSegBlocks];
  __shared__ int s_real[kSegBlocks];
  __shared__ uint32_t s_gbits[kSegBlocks];  // bits of group g's interval
  __shared__ uint8_t s_zz[64];
  const uint32_t lane = threadIdx.x;
  copy_words_to_lds<4>(s_tab, a.tables, 2 * (16 + 256), lane, 64);
  s_zz[lane] = a.zigzag[lane];
  const int bpm = a.blocks_per_mcu;
  const int per = a.ri * bpm;                                // blocks per interval (<= 64, checked by the host)
  const int G = (RETRY || per > 32) ? 1 : 64 / per;          // intervals per wavefront
  const uint32_t group_words = (uint32_t)(kSegBlocks * WORDS) / (uint32_t)G;
  const int nwseg = (a.nseg + G - 1) / G;

  for (int wseg = (int)blockIdx.x; wseg < nwseg; wseg += (int)gridDim.x) {
    __syncthreads();  // the previous intervals' LDS contents are dead
    // ---- which interval and which block is this lane's? ---------------------------------------------------------
    const int g = (int)lane / per, l = (int)lane - g * per;  // group, block inside the interval
    const int seg = wseg * G + g;
    if constexpr (RETRY) {
      if (a.seg_bytes[wseg] != kRetry) continue;  // wave-uniform: G == 1, the wave's one interval is wseg
    }
    const int mcu_local = l / bpm, k_in_mcu = l - mcu_local * bpm;
    const int mcu = seg * a.ri + mcu_local;
    const bool active = g < G && seg < a.nseg && mcu_local < a.ri && mcu < a.total_mcus;
    int c = 0, kk = k_in_mcu;
    if (a.ncomp > 1) {
      while (c < a.ncomp - 1 && kk >= a.hs[c] * a.vs[c]) { kk -= a.hs[c] * a.vs[c]; c++; }
    }
    const int hs = a.ncomp > 1 ? a.hs[c] : 1, vs = a.ncomp > 1 ? a.vs[c] : 1;
    const int yi = kk / hs, xi = kk - yi * hs;
    const int my = mcu / a.mcus_per_row, mx = mcu - my * a.mcus_per_row;
    const int by = my * vs + yi, bx = mx * hs + xi;
    const bool real = active && by < a.bh[c] && bx < a.bw[c];
    uint32_t* crow = s_coef + lane * kCoefRow;
    if (real) {
      const uint4* src = (const uint4*)(a.coef[c] + ((size_t)by * a.bw[c] + bx) * 64);
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const uint4 q = src[i];
        crow[4 * i] = q.x; crow[4 * i + 1] = q.y; crow[4 * i + 2] = q.z; crow[4 * i + 3] = q.w;
      }
    }
    s_real[lane] = real ? 1 : 0;
    s_dc[lane] = real ? (int)(int16_t)(crow[0] & 0xffffu) : 0;
    __syncthreads();
    // dummy blocks (jctrans.c compress_output): DC of the previous block of the MCU; the first block of a component
    // inside an MCU is always real, so the search is bounded by the component's block count (and stays inside the MCU)
    int dcv = s_dc[lane];
    if (active && !real) {
      for (int d = 1; d <= 3; d++) {
        if (k_in_mcu >= d && s_real[lane - d]) { dcv = s_dc[lane - d]; break; }
      }
    }
    __syncthreads();
    s_dc[lane] = dcv;
    __syncthreads();
    // DC prediction: the previous block of the same component inside the interval (0 at its start)
    int pred = 0;
    if (active) {
      if (kk > 0) pred = s_dc[lane - 1];
      else if (mcu_local > 0) pred = s_dc[(int)lane - bpm + hs * vs - 1];
    }
    const uint32_t* dct = s_tab + (c ? (16 + 256) : 0);
    const uint32_t* act = dct + 16;
    const int diff = dcv - pred;
    const uint64_t nz = zigzag_nonzero_map(s_coef, s_zz[lane], lane);  // (rows of lanes without a real block hold stale words: never walked)

    // ---- pass 1: code lengths -> bit offsets inside the lane's interval ---------------------------------------------------
    uint32_t len = 0, oob = 0;
    if (active) walk_block(dct, act, diff, real, crow, s_zz, nz, oob, [&](uint32_t, uint32_t n) { len += n; });
    const uint32_t incl = wave_incl_scan(len, lane);
    // every lane takes part in both shuffles (a lane that sits out cannot be read from)
    const int first = g * per, last = min(first + per - 1, 63);
    const uint32_t v_before = (uint32_t)__shfl((int)incl, min(max(first - 1, 0), 63), 64);
    const uint32_t v_last = (uint32_t)__shfl((int)incl, last, 64);
    const uint32_t before = (first > 0 && first < 64) ? v_before : 0u;
    const uint32_t total_bits = g < G ? v_last - before : 0u;  // of this lane's interval
    const uint32_t off = incl - len - before;
    const uint32_t cap_bits = group_words * 32u;
    if (__builtin_amdgcn_ballot_w64(oob != 0) != 0) {  // coefficients outside the baseline range: report, do not write
      if (l == 0 && g < G && seg < a.nseg) a.seg_bytes[seg] = kBadCoef;
      continue;
    }
    const bool fits = total_bits <= cap_bits;  // else: next size class (cannot happen in the worst-case class for in-range coefficients)
    if (l == 0 && g < G) s_gbits[g] = fits ? total_bits : 0xFFFFFFFFu;
    for (uint32_t i = lane; i < (uint32_t)(kSegBlocks * WORDS) + 2u; i += 64) s_bits[i] = 0u;
    __syncthreads();

    // ---- pass 2: emit into the group's slice of the bit buffer ---------------------------------------------------------------
    const uint32_t gbase = (uint32_t)g * group_words;
    if (active && fits) {
      uint64_t acc = 0;
      uint32_t cnt = off & 31u, w = gbase + (off >> 5);
      walk_block(dct, act, diff, real, crow, s_zz, nz, oob, [&](uint32_t code, uint32_t n) {
        acc = (acc << n) | (uint64_t)code;
        cnt += n;
        if (cnt >= 32u) {
          cnt -= 32u;
          atomicOr(&s_bits[w++], (uint32_t)(acc >> cnt));
        }
      });
      if (cnt) atomicOr(&s_bits[w], (uint32_t)(acc << (32u - cnt)));
    }
    __syncthreads();
    // flush_bits: fill the last partial byte of every interval with ones
    if (l == 0 && g < G && fits && (total_bits & 7u)) {
      const uint32_t pad = 8u - (total_bits & 7u), pos = total_bits & 31u;
      atomicOr(&s_bits[gbase + (total_bits >> 5)], ((1u << pad) - 1u) << (32u - pos - pad));
    }
    __syncthreads();

    // ---- byte stuffing + store, one interval after the other ------------------------------------------------------------------
    for (int gg = 0; gg < G; gg++) {
      const int sg = wseg * G + gg;
      if (sg >= a.nseg) break;
      const uint32_t tb = s_gbits[gg];
      if (tb == 0xFFFFFFFFu) {
        if (lane == 0) a.seg_bytes[sg] = RETRY ? kBadCoef : kRetry;
        continue;
      }
      const uint32_t nbytes = (tb + 7u) >> 3;
      const uint32_t* bits = s_bits + (uint32_t)gg * group_words;
      uint8_t* dst = a.slots + (size_t)sg * a.slot_stride;
      uint32_t carry = 0;  // 0xFF bytes before this chunk
      for (uint32_t base = 0; base < nbytes; base += 256) {
        const uint32_t i0 = base + lane * 4;
        const uint32_t wd = i0 < nbytes ? bits[i0 >> 2] : 0u;
        const uint32_t nv = i0 < nbytes ? min(nbytes - i0, 4u) : 0u;
        uint32_t b[4], nff = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          b[k] = (wd >> (24 - 8 * k)) & 0xffu;
          if ((uint32_t)k < nv && b[k] == 0xffu) nff++;
        }
        const uint32_t incl_ff = wave_incl_scan(nff, lane);
        uint32_t pos = i0 + carry + (incl_ff - nff);
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if ((uint32_t)k < nv) {
            dst[pos++] = (uint8_t)b[k];
            if (b[k] == 0xffu) dst[pos++] = 0;
          }
        }
        carry += (uint32_t)__shfl((int)incl_ff, 63, 64);
      }
      if (lane == 0) a.seg_bytes[sg] = nbytes + carry;
    }
  }
}

// offsets[s] = sum over t < s of (seg_bytes[t] + 2)  (an RSTn marker follows every interval but the last);
// offsets[nseg] = total length; status = 1 when an interval reported out-of-range coefficients
__global__ __launch_bounds__(1024) void huff_offsets_kernel(const uint32_t* __restrict__ seg_bytes, int nseg, uint64_t* __restrict__ offsets,
                                                            uint32_t* __restrict__ status) {
  __shared__ uint64_t s_sum[1024];
  __shared__ uint32_t s_bad;
  const int tid = (int)threadIdx.x;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  const int per = (nseg + 1023) / 1024, lo = min(tid * per, nseg), hi = min(lo + per, nseg);
  uint64_t sum = 0;
  bool bad = false;
  for (int i = lo; i < hi; i++) {
    const uint32_t n = seg_bytes[i];
    bad |= n >= kRetry;
    sum += (uint64_t)n + 2u;
  }
  if (bad) atomicOr(&s_bad, 1u);
  s_sum[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {  // Hillis-Steele inclusive scan
    const uint64_t y = tid >= d ? s_sum[tid - d] : 0;
    __syncthreads();
    s_sum[tid] += y;
    __syncthreads();
  }
  uint64_t run = s_sum[tid] - sum;
  for (int i = lo; i < hi; i++) {
    offsets[i] = run;
    run += (uint64_t)seg_bytes[i] + 2u;
  }
  if (tid == 1023) offsets[nseg] = s_sum[1023] - 2u;  // no marker after the last interval
  if (tid == 0) *status = s_bad;
}

__global__ __launch_bounds__(256) void huff_gather_kernel(const uint8_t* __restrict__ slots, uint32_t slot_stride,
                                                          const uint32_t* __restrict__ seg_bytes, const uint64_t* __restrict__ offsets, int nseg,
                                                          uint8_t* __restrict__ out, uint64_t cap) {
  for (int seg = (int)blockIdx.x; seg < nseg; seg += (int)gridDim.x) {
    const uint32_t n = seg_bytes[seg];
    const uint64_t off = offsets[seg];
    if (n >= kRetry || off + n + 2u > cap + (seg == nseg - 1 ? 2u : 0u)) continue;  // the host reports the error
    const uint8_t* src = slots + (size_t)seg * slot_stride;
    for (uint32_t i = threadIdx.x; i < n; i += 256) out[off + i] = src[i];
    if (threadIdx.x == 0 && seg < nseg - 1) {
      out[off + n] = 0xff;
      out[off + n + 1] = (uint8_t)(0xd0 + (seg & 7));  // emit_restart: RST0..RST7 in rotation
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// Scans WITHOUT restart markers -- the stream the reference itself writes (jpegencoderhelper.cpp:187-201 never sets
// cinfo.restart_interval).  Nothing in ENCODING is serial: the DC predictor of a block is the DC value of a block whose
// coefficients are already in memory, the bit position of a block is a prefix sum of code lengths, and byte stuffing is a
// prefix sum of 0xFF counts.  The scan is cut into SEGMENTS of `ri` MCUs (not restart intervals: no marker, no alignment,
// no predictor reset), one wavefront each:
//   pass A  lengths: the wave walks its blocks once and stores the segment's bit count; lanes [0, blocks_per_mcu) hold the
//           MCU in FRONT of the segment (DC values only) so that the first MCU's predictors, libjpeg's dummy-block rule
//           included, come out of the same LDS exchange as all the others
//   scan    exclusive scan of the segment bit counts (one workgroup) -> every segment's first bit in the stream; the words
//           that two segments share are zeroed
//   pass B  emit: same walk, bits go to the LDS buffer at phase (first bit & 31) and from there to the unstuffed stream in
//           memory: interior words are plain stores, the first / last word of a segment is OR-ed in (its neighbour writes the
//           rest); the last segment appends flush_bits' one-padding
//   stuff   0xFF counts per 4 KiB chunk -> scan -> scatter with the stuffed zero bytes (jchuff.c emit_bits)
// The result equals libjpeg's entropy-coded segment byte for byte (tests: against the sequential CPU restatement of jchuff.c
// at restart_interval 0 and against the files the reference encoder writes).
constexpr int kStuffChunk = 4096;  // raw bytes per workgroup of the stuffing passes (256 threads x 16 bytes)

template <int WORDS, int PASS>  // PASS 0: lengths; 1: emit, small LDS buffer; 2: emit, worst-case buffer
__global__ __launch_bounds__(64) void huff_stream_kernel(const HuffArgs a, const HuffStream t) {
  __shared__ __attribute__((aligned(16))) uint32_t s_tab[2 * (16 + 256)];
  __shared__ uint32_t s_coef[kSegBlocks * kCoefRow];
  __shared__ uint32_t s_bits[PASS == 0 ? 1 : kSegBlocks * WORDS + 2];
  __shared__ int s_dc[kSegBlocks];
  __shared__ int s_real[kSegBlocks];
  __shared__ uint8_t s_zz[64];
  const uint32_t lane = threadIdx.x;
  if constexpr (PASS == 2) {
    if (t.meta[3] == 0) return;  // no segment of the large size class (round 6: this launch used to stage its tables and walk the segment list for nothing)
  }
  copy_words_to_lds<4>(s_tab, a.tables, 2 * (16 + 256), lane, 64);
  s_zz[lane] = a.zigzag[lane];
  const int bpm = a.blocks_per_mcu;
  constexpr uint32_t kCapBits = (uint32_t)(kSegBlocks * WORDS) * 32u;
  for (int seg = (int)blockIdx.x; seg < a.nseg; seg += (int)gridDim.x) {
    __syncthreads();
    uint64_t start = 0;
    uint32_t seg_total = 0;
    if constexpr (PASS != 0) {  // wave-uniform size-class choice
      seg_total = t.seg_bits[seg];
      start = t.seg_start[seg];
      const bool small = seg_total + 32u <= (uint32_t)(kSegBlocks * kWordsSmall) * 32u;
      if (seg_total >= kRetry || small != (PASS == 1)) continue;
    }
    const int mcu_local = (int)lane / bpm, k_in_mcu = (int)lane - mcu_local * bpm;  // MCU 0 of the wave = the one before the segment
    const int mcu = seg * a.ri + mcu_local - 1;
    const bool valid = mcu_local <= a.ri && mcu >= 0 && mcu < a.total_mcus;
    const bool emits = valid && mcu_local >= 1;
    int c = 0, kk = k_in_mcu;
    if (a.ncomp > 1) {
      while (c < a.ncomp - 1 && kk >= a.hs[c] * a.vs[c]) { kk -= a.hs[c] * a.vs[c]; c++; }
    }
    const int hs = a.ncomp > 1 ? a.hs[c] : 1, vs = a.ncomp > 1 ? a.vs[c] : 1;
    const int yi = kk / hs, xi = kk - yi * hs;
    const int mcu_c = valid ? mcu : 0;
    const int my = mcu_c / a.mcus_per_row, mx = mcu_c - my * a.mcus_per_row;
    const int by = my * vs + yi, bx = mx * hs + xi;
    const bool real = valid && by < a.bh[c] && bx < a.bw[c];
    uint32_t* crow = s_coef + lane * kCoefRow;
    if (real) {
      const uint4* src = (const uint4*)(a.coef[c] + ((size_t)by * a.bw[c] + bx) * 64);
      if (emits) {
#pragma unroll
        for (int i = 0; i < 8; i++) {
          const uint4 q = src[i];
          crow[4 * i] = q.x; crow[4 * i + 1] = q.y; crow[4 * i + 2] = q.z; crow[4 * i + 3] = q.w;
        }
      } else {
        crow[0] = *(const uint32_t*)src;  // the MCU in front of the segment: its DC values are all that matters
      }
    }
    s_real[lane] = real ? 1 : 0;
    s_dc[lane] = real ? (int)(int16_t)(crow[0] & 0xffffu) : 0;
    __syncthreads();
    int dcv = s_dc[lane];
    if (valid && !real) {  // dummy block (jctrans.c compress_output): DC of the previous block of the MCU
      for (int d = 1; d <= 3; d++) {
        if (k_in_mcu >= d && s_real[lane - d]) { dcv = s_dc[lane - d]; break; }
      }
    }
    __syncthreads();
    s_dc[lane] = dcv;
    __syncthreads();
    int pred = 0;
    if (emits) {
      if (kk > 0) pred = s_dc[lane - 1];
      else pred = s_dc[(int)lane - bpm + hs * vs - 1];  // last block of the component in the previous MCU (0 in front of MCU 0)
    }
    const uint32_t* dct = s_tab + (c ? (16 + 256) : 0);
    const uint32_t* act = dct + 16;
    const int diff = dcv - pred;
    const uint64_t nz = zigzag_nonzero_map(s_coef, s_zz[lane], lane);  // (only rows of emitting lanes with a real block are ever walked)
    uint32_t len = 0, oob = 0;
    if (emits) walk_block(dct, act, diff, real, crow, s_zz, nz, oob, [&](uint32_t, uint32_t n) { len += n; });
    const uint32_t incl = wave_incl_scan(len, lane);
    const uint32_t total_bits = (uint32_t)__shfl((int)incl, 63, 64);
    if constexpr (PASS == 0) {
      const bool bad = __builtin_amdgcn_ballot_w64(oob != 0) != 0;
      if (lane == 0) t.seg_bits[seg] = bad ? kBadCoef : total_bits;
    } else {
      const uint32_t phase = (uint32_t)(start & 31u);
      if (phase + total_bits > kCapBits) continue;  // cannot happen: the size class was chosen from the same count
      const uint32_t nwords = (phase + total_bits + 31u) >> 5;
      for (uint32_t i = lane; i < nwords + 1u; i += 64) s_bits[i] = 0u;
      __syncthreads();
      if (emits) {
        const uint32_t off = phase + incl - len;
        uint64_t acc = 0;
        uint32_t cnt = off & 31u, w = off >> 5;
        walk_block(dct, act, diff, real, crow, s_zz, nz, oob, [&](uint32_t code, uint32_t n) {
          acc = (acc << n) | (uint64_t)code;
          cnt += n;
          if (cnt >= 32u) {
            cnt -= 32u;
            atomicOr(&s_bits[w++], (uint32_t)(acc >> cnt));
          }
        });
        if (cnt) atomicOr(&s_bits[w], (uint32_t)(acc << (32u - cnt)));
      }
      __syncthreads();
      const uint64_t end_bit = start + total_bits;
      if (seg == a.nseg - 1 && lane == 0 && (end_bit & 7u)) {  // flush_bits: one-fill the last partial byte of the scan
        const uint32_t e = phase + total_bits, pad = 8u - (e & 7u), pos = e & 31u;
        s_bits[e >> 5] |= ((1u << pad) - 1u) << (32u - pos - pad);
      }
      __syncthreads();
      // words -> the unstuffed stream (bytes are MSB first: byte-swap the word)
      const uint64_t w0 = start >> 5;
      const uint32_t tail = (phase + total_bits) & 31u;
      for (uint32_t i = lane; i < nwords; i += 64) {
        if (w0 + i >= t.raw_words) break;  // capacity: the host reports the size the stream needs
        const uint32_t v = __builtin_bswap32(s_bits[i]);
        const bool shared = (i == 0 && phase != 0) || (i == nwords - 1 && tail != 0);
        if (shared) atomicOr(&t.raw[w0 + i], v);
        else t.raw[w0 + i] = v;
      }
    }
  }
}

// One 1024-thread workgroup scans n words in tiles of 8192: a tile is loaded COALESCED into LDS (eight words per thread), every
// thread adds up its eight consecutive words there, a ten-step block scan gives the threads' offsets, `sink(index, value, exclusive
// prefix)` is called for every element.  Round 5's form walked `per` consecutive elements per thread straight from global memory --
// one cache line per lane and instruction, each load waiting for the one before: 40 us for the 6075 chunk counts of a 4K map.
template <int NT = 1024, typename Sink>  // NT threads; tiles of 8 * NT words
__device__ __forceinline__ uint64_t wg_scan_tiles(const uint32_t* __restrict__ v, int n, uint32_t* s_val /* 8 * NT */, uint64_t* s_sum /* NT / 64 */, Sink sink) {
  const int tid = (int)threadIdx.x;
  uint64_t carry = 0;
  for (int base = 0; base < n; base += 8 * NT) {
    // all eight loads in flight (clamped index; "i < n ? v[i] : 0" compiles to a branch per load with the wait for it inside: eight memory
    // latencies per tile on the one workgroup everything behind this kernel waits for -- round 6, tools/lds_staging_check.py)
    uint32_t ld[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ld[j] = v[min(base + j * NT + tid, n - 1)];
#pragma unroll
    for (int j = 0; j < 8; j++) s_val[j * NT + tid] = base + j * NT + tid < n ? ld[j] : 0u;
    __syncthreads();
    uint32_t x[8];
    uint64_t sum = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { x[j] = s_val[tid * 8 + j]; sum += x[j]; }
    uint64_t sc[1] = {sum}, all[1];
    wg_incl_scan<NT, 1>(sc, s_sum, all);  // (ends with a barrier: s_val may be rewritten by the next tile)
    uint64_t run = carry + sc[0] - sum;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = base + tid * 8 + j;
      if (i < n) sink(i, x[j], run);
      run += x[j];
    }
    carry += all[0];
  }
  return carry;
}

// seg_start = exclusive scan of seg_bits (bit offsets), seg_start[nseg] = total; meta[0..1] = total bits, meta[2] = status
// (1: coefficients outside the baseline range); every word that two segments share (and the word behind the last bit) is zeroed
// 256 threads (round 6: a 1024-thread workgroup with 40 KB of LDS waited 15-35 us for a CU with that much room while the other scan's kernels
// filled the device; this one fits next to anything)
constexpr int kStreamScanThreads = 256;
__global__ __launch_bounds__(kStreamScanThreads) void huff_stream_scan_kernel(const uint32_t* __restrict__ seg_bits, int nseg, const HuffStream t) {
  __shared__ uint32_t s_val[8 * kStreamScanThreads];
  __shared__ uint64_t s_sum[kStreamScanThreads / 64];
  __shared__ uint32_t s_bad;
  const int tid = (int)threadIdx.x;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  bool bad = false;
  uint32_t big = 0;  // segments of the large size class (emit pass 2): usually none -- that pass then leaves at once (meta[3])
  // (a segment that must be redone -- kRetry / kBadCoef -- poisons the sums behind it; the status word makes the host discard them)
  const uint64_t total = wg_scan_tiles<kStreamScanThreads>(seg_bits, nseg, s_val, s_sum, [&](int i, uint32_t n, uint64_t run) {
    bad |= n >= kRetry;
    big += (n < kRetry && n + 32u > (uint32_t)(kSegBlocks * kWordsSmall) * 32u) ? 1u : 0u;
    t.seg_start[i] = run;
    if ((run >> 5) < t.raw_words) t.raw[run >> 5] = 0u;
  });
  if (bad) atomicOr(&s_bad, 1u);
  if (big) atomicAdd(&s_bad, big << 1);  // (bit 0: bad; the count above it)
  __syncthreads();
  if (tid == kStreamScanThreads - 1) {
    t.seg_start[nseg] = total;
    if ((total >> 5) < t.raw_words) t.raw[total >> 5] = 0u;
    t.meta[0] = (uint32_t)total;
    t.meta[1] = (uint32_t)(total >> 32);
  }
  if (tid == 0) {
    t.meta[2] = s_bad & 1u;
    t.meta[3] = s_bad >> 1;
  }
}

// 0xFF bytes per chunk of the unstuffed stream.  Round 6: sixteen bytes per thread (one 16-byte load; chunks of 4 KiB) instead of four,
// and grids sized for the stream that exists (the host passes the capacity; blocks beyond the last raw byte leave at once).
__device__ __forceinline__ uint32_t ff_mask16(const uint4& v, uint32_t nv) {  // bit k set: byte k (of the first nv) is 0xFF
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t m = 0;
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) m |= (k < nv && ((w[k >> 2] >> (8u * (k & 3u))) & 0xffu) == 0xffu) ? (1u << k) : 0u;
  return m;
}
__device__ __forceinline__ uint4 raw_load16(const HuffStream& t, uint64_t base) {  // raw[] is word-addressed and 16-byte aligned (hipMalloc)
  const uint64_t w = base >> 2;
  if (w + 4u <= t.raw_words) return *(const uint4*)(t.raw + w);
  uint32_t x[4] = {0, 0, 0, 0};
  for (uint32_t k = 0; k < 4; k++)
    if (w + k < t.raw_words) x[k] = t.raw[w + k];
  return make_uint4(x[0], x[1], x[2], x[3]);
}
__global__ __launch_bounds__(256) void huff_stuff_count_kernel(const HuffStream t, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_n;
  const uint64_t total_bits = (uint64_t)t.meta[0] | ((uint64_t)t.meta[1] << 32);
  const uint64_t nraw = (total_bits + 7u) >> 3;
  const uint64_t base = (uint64_t)blockIdx.x * kStuffChunk + threadIdx.x * 16u;
  if ((uint64_t)blockIdx.x * kStuffChunk >= nraw) {  // (workgroup-uniform) nothing of the stream in this chunk
    if (threadIdx.x == 0) counts[blockIdx.x] = 0;
    return;
  }
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  uint32_t n = 0;
  if (base < nraw) n = (uint32_t)__builtin_popcount(ff_mask16(raw_load16(t, base), (uint32_t)min((uint64_t)16, nraw - base)));
  n = wave_incl_scan(n, threadIdx.x & 63);
  if ((threadIdx.x & 63) == 63 && n) atomicAdd(&s_n, n);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_n;
}

// exclusive scan of the chunk counts in place (one workgroup); out_bytes = raw bytes + stuffed zeros.  (Folding this into the count
// kernel's last workgroup was measured and dropped: see huffman_decode_sync.hip, unstuff_count_kernel.)
__global__ __launch_bounds__(kStreamScanThreads) void huff_stuff_scan_kernel(uint32_t* __restrict__ counts, int nchunks, const HuffStream t, uint64_t* __restrict__ out_bytes) {
  __shared__ uint32_t s_val[8 * kStreamScanThreads];
  __shared__ uint64_t s_sum[kStreamScanThreads / 64];
  const uint64_t total_bits = (uint64_t)t.meta[0] | ((uint64_t)t.meta[1] << 32);
  const uint64_t nraw = (total_bits + 7u) >> 3;
  const int used = (int)min((uint64_t)nchunks, (nraw + kStuffChunk - 1) / kStuffChunk);  // chunks beyond the stream hold zeros and are never read
  // (< 2^32: the stream is bounded by the 32-bit capacity checked on the host)
  const uint64_t total = wg_scan_tiles<kStreamScanThreads>(counts, used, s_val, s_sum, [&](int i, uint32_t, uint64_t run) { counts[i] = (uint32_t)run; });
  if (threadIdx.x == kStreamScanThreads - 1) *out_bytes = nraw + total;
}

__global__ __launch_bounds__(256) void huff_stuff_scatter_kernel(const HuffStream t, const uint32_t* __restrict__ chunk_base, uint8_t* __restrict__ out, uint64_t cap) {
  __shared__ uint32_t s_wave[4];
  const uint64_t total_bits = (uint64_t)t.meta[0] | ((uint64_t)t.meta[1] << 32);
  const uint64_t nraw = (total_bits + 7u) >> 3;
  if ((uint64_t)blockIdx.x * kStuffChunk >= nraw) return;  // (wave-uniform)
  const uint64_t base = (uint64_t)blockIdx.x * kStuffChunk + threadIdx.x * 16u;
  uint4 v = make_uint4(0, 0, 0, 0);
  uint32_t nv = 0;
  if (base < nraw) {
    v = raw_load16(t, base);
    nv = (uint32_t)min((uint64_t)16, nraw - base);
  }
  const uint32_t mask = ff_mask16(v, nv), nff = (uint32_t)__builtin_popcount(mask);
  const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const uint32_t incl = wave_incl_scan(nff, lane);
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  uint32_t before = chunk_base[blockIdx.x] + incl - nff;
  for (uint32_t k = 0; k < wv; k++) before += s_wave[k];
  uint64_t pos = base + before;
  if (mask == 0 && nv == 16 && pos + 16 <= cap) {
    // no 0xFF among these sixteen bytes (15 threads in 16): four dword stores at the shifted, in general unaligned, destination
    uint8_t* d = out + pos;
    __builtin_memcpy(d, &v.x, 4);
    __builtin_memcpy(d + 4, &v.y, 4);
    __builtin_memcpy(d + 8, &v.z, 4);
    __builtin_memcpy(d + 12, &v.w, 4);
    return;
  }
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  for (uint32_t k = 0; k < nv; k++) {
    const uint32_t b = (w[k >> 2] >> (8u * (k & 3u))) & 0xffu;
    if (pos < cap) out[pos] = (uint8_t)b;
    pos++;
    if (b == 0xffu) {
      if (pos < cap) out[pos] = 0;
      pos++;
    }
  }
}

}  // namespace

uint32_t huff_slot_stride() { return (uint32_t)(kSegBlocks * kWordsPerBlock * 4 * 2); }  // every byte could be stuffed

hipError_t launch_huffman_encode(const HuffArgs& a, uint64_t* offsets, uint32_t* status, uint8_t* out, uint64_t cap, hipStream_t s) {
  int grid = a.nseg < 8192 ? a.nseg : 8192;
  const int per = a.ri * a.blocks_per_mcu, groups = per > 32 ? 1 : 64 / per;  // intervals per wavefront in the first launch
  const int nwseg = (a.nseg + groups - 1) / groups;
  hipLaunchKernelGGL((huff_encode_kernel<kWordsSmall, false>), dim3(nwseg < 8192 ? nwseg : 8192), dim3(64), 0, s, a);
  hipLaunchKernelGGL((huff_encode_kernel<kWordsPerBlock, true>), dim3(grid < 2048 ? grid : 2048), dim3(64), 0, s, a);
  hipLaunchKernelGGL(huff_offsets_kernel, dim3(1), dim3(1024), 0, s, a.seg_bytes, a.nseg, offsets, status);
  hipLaunchKernelGGL(huff_gather_kernel, dim3(grid), dim3(256), 0, s, a.slots, a.slot_stride, a.seg_bytes, offsets, a.nseg, out, cap);
  return hipGetLastError();
}


// MCUs per wavefront segment of the marker-less encoder: the wave's first blocks_per_mcu lanes carry the MCU in front
int huff_stream_segment_mcus(int blocks_per_mcu) { return kSegBlocks / blocks_per_mcu - 1; }
int huff_stuff_chunks(uint64_t raw_bytes) { return (int)((raw_bytes + kStuffChunk - 1) / kStuffChunk); }

// a.ri = huff_stream_segment_mcus(), a.nseg segments; t.raw holds t.raw_words words; chunk_counts: huff_stuff_chunks(raw capacity)
// words; out_bytes (device): stuffed size.  Everything is stream ordered; the host reads t.meta / out_bytes afterwards.
hipError_t launch_huffman_encode_stream(const HuffArgs& a, const HuffStream& t, uint32_t* chunk_counts, uint64_t* out_bytes, uint8_t* out, uint64_t cap,
                                        hipStream_t s) {
  const int grid = a.nseg < 16384 ? a.nseg : 16384;
  const int nchunks = huff_stuff_chunks(t.raw_words * 4u);
  hipLaunchKernelGGL((huff_stream_kernel<1, 0>), dim3(grid), dim3(64), 0, s, a, t);
  hipLaunchKernelGGL(huff_stream_scan_kernel, dim3(1), dim3(kStreamScanThreads), 0, s, (const uint32_t*)t.seg_bits, a.nseg, t);
  hipLaunchKernelGGL((huff_stream_kernel<kWordsSmall, 1>), dim3(grid), dim3(64), 0, s, a, t);
  hipLaunchKernelGGL((huff_stream_kernel<kWordsPerBlock, 2>), dim3(grid < 2048 ? grid : 2048), dim3(64), 0, s, a, t);
  hipLaunchKernelGGL(huff_stuff_count_kernel, dim3(nchunks), dim3(256), 0, s, t, chunk_counts);
  hipLaunchKernelGGL(huff_stuff_scan_kernel, dim3(1), dim3(kStreamScanThreads), 0, s, chunk_counts, nchunks, t, out_bytes);
  hipLaunchKernelGGL(huff_stuff_scatter_kernel, dim3(nchunks), dim3(256), 0, s, t, (const uint32_t*)chunk_counts, out, cap);
  return hipGetLastError();
}

}  // namespace uhdr
