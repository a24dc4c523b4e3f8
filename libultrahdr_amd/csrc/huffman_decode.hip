// Baseline Huffman DEcoding on gfx950, one restart interval per lane (decode direction).
//
// In the reference this is libjpeg behind JpegDecoderHelper::decompressImage
// (/root/reference/lib/src/jpegdecoderhelper.cpp:169-535 -> jdhuff.c decode_mcu_huff): bit-serial, every code's
// position depends on all codes before it.  A stream that carries restart markers (T.81 B.2.4.4; the streams
// uhdr_hip_huffman_encode_dev writes do, the reference's own files do not) is a sequence of independent intervals:
// byte aligned, DC predictors reset.  Three steps, all on the device:
//   1. find the RSTn markers: inside entropy-coded data 0xFF is followed by 0x00 (stuffing) or by a marker, so
//      "0xFF, 0xD0..0xD7" is unambiguous; chunk counts -> exclusive scan -> each marker's rank = its interval;
//   2. every lane decodes one interval on its own (jdhuff.c's arithmetic: 9-bit look-ahead table, canonical
//      maxcode / valoffset walk for longer codes, HUFF_EXTEND), writes the coefficients in natural order into the
//      zero-initialised JBLOCK arrays and drops the dummy blocks of edge MCUs;
//   3. malformed input (bad code, run past the end of a block, marker count or numbering that does not match the
//      restart interval) raises a status flag: UHDR_CODEC_INVALID_PARAM at the API.
// Tables are whatever the file's DHT segments hold (built on the host from BITS / HUFFVAL, T.81 Annex C / F.2.2.3).
// A stream without restart markers is one interval: correct, but decoded by a single lane.
#include "lds_copy.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kChunk = 4096;  // bytes per workgroup of the marker scan (256 threads x 16 bytes)

__device__ __forceinline__ bool is_rst(const uint8_t* __restrict__ d, uint32_t i, uint32_t n) {
  return d[i] == 0xffu && i + 1 < n && (d[i + 1] & 0xf8u) == 0xd0u;
}

__global__ __launch_bounds__(256) void huff_count_markers_kernel(const uint8_t* __restrict__ data, uint32_t n, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_cnt;
  if (threadIdx.x == 0) s_cnt = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * kChunk + threadIdx.x * 16;
  uint32_t c = 0;
  for (uint32_t i = base; i < base + 16 && i < n; i++) c += is_rst(data, i, n) ? 1u : 0u;
  if (c) atomicAdd(&s_cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = s_cnt;
}

// single workgroup: counts[nchunks] -> exclusive prefix sums (in place), total -> *total
__global__ __launch_bounds__(1024) void huff_scan_counts_kernel(uint32_t* __restrict__ counts, int nchunks, uint32_t* __restrict__ total) {
  __shared__ uint32_t s_sum[1024];
  const int tid = (int)threadIdx.x;
  const int per = (nchunks + 1023) / 1024, lo = min(tid * per, nchunks), hi = min(lo + per, nchunks);
  uint32_t sum = 0;
  for (int i = lo; i < hi; i++) sum += counts[i];
  s_sum[tid] = sum;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {
    const uint32_t y = tid >= d ? s_sum[tid - d] : 0u;
    __syncthreads();
    s_sum[tid] += y;
    __syncthreads();
  }
  uint32_t run = s_sum[tid] - sum;
  for (int i = lo; i < hi; i++) {
    const uint32_t c = counts[i];
    counts[i] = run;
    run += c;
  }
  if (tid == 1023) *total = s_sum[1023];
}

// interval k = bytes [starts[k], ends[k]); marker r (in stream order) closes interval r and must be RST(r mod 8)
__global__ __launch_bounds__(256) void huff_emit_intervals_kernel(const uint8_t* __restrict__ data, uint32_t n, const uint32_t* __restrict__ chunk_base,
                                                                  int nseg, uint32_t* __restrict__ starts, uint32_t* __restrict__ ends,
                                                                  uint32_t* __restrict__ status) {
  __shared__ uint32_t s_scan[256];
  const uint32_t tid = threadIdx.x, base = blockIdx.x * kChunk + tid * 16;
  uint32_t c = 0;
  for (uint32_t i = base; i < base + 16 && i < n; i++) c += is_rst(data, i, n) ? 1u : 0u;
  s_scan[tid] = c;
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const uint32_t y = tid >= d ? s_scan[tid - d] : 0u;
    __syncthreads();
    s_scan[tid] += y;
    __syncthreads();
  }
  uint32_t rank = chunk_base[blockIdx.x] + s_scan[tid] - c;
  for (uint32_t i = base; i < base + 16 && i < n; i++) {
    if (!is_rst(data, i, n)) continue;
    if (rank + 1 < (uint32_t)nseg) {
      ends[rank] = i;
      starts[rank + 1] = i + 2;
      if ((data[i + 1] & 7u) != (rank & 7u)) atomicOr(status, 4u);  // markers out of sequence
    }
    rank++;
  }
  if (blockIdx.x == 0 && tid == 0) {
    starts[0] = 0;
    ends[nseg - 1] = n;
  }
}

struct BitReader {
  const uint8_t* d;
  uint32_t p, end;
  uint64_t acc;
  int n;
  __device__ __forceinline__ void fill() {
    while (n <= 56) {
      uint32_t b = 0;  // past the end of the interval: zeros, as jdhuff.c feeds after a marker
      if (p < end) {
        b = d[p++];
        if (b == 0xffu && p < end && d[p] == 0) p++;  // stuffed zero
      }
      acc = (acc << 8) | b;
      n += 8;
    }
  }
  __device__ __forceinline__ uint32_t peek(int k) const { return (uint32_t)(acc >> (n - k)) & ((1u << k) - 1u); }
  __device__ __forceinline__ void skip(int k) { n -= k; }
};

__device__ __forceinline__ int decode_symbol(BitReader& r, const HuffDecTable& t, bool& err) {
  const uint32_t e = t.lut[r.peek(9)];
  if (e) {
    r.skip((int)(e >> 8));
    return (int)(e & 255u);
  }
  for (int l = 10; l <= 16; l++) {
    const int code = (int)r.peek(l);
    if (t.maxcode[l] >= 0 && code <= t.maxcode[l]) {
      r.skip(l);
      return (int)t.vals[(t.valoff[l] + code) & 255];
    }
  }
  err = true;
  r.skip(16);
  return 0;
}

__device__ __forceinline__ int receive_extend(BitReader& r, int s) {  // jdhuff.c HUFF_EXTEND
  const int v = (int)r.peek(s);
  r.skip(s);
  return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v;
}

__global__ __launch_bounds__(64) void huff_decode_kernel(const HuffDecArgs a) {
  __shared__ __attribute__((aligned(16))) HuffDecTable s_t[4];
  __shared__ uint8_t s_zz[64];
  {
    static_assert(sizeof(HuffDecTable) * 4 % 16 == 0, "staged with 16-byte loads");
    copy_words_to_lds<4>((uint32_t*)s_t, (const uint32_t*)a.tabs, (uint32_t)(sizeof(HuffDecTable) * 4 / 4), threadIdx.x, 64);
    s_zz[threadIdx.x] = a.zigzag[threadIdx.x];
  }
  __syncthreads();
  const int ri = a.ri > 0 ? a.ri : a.total_mcus;
  for (int seg = (int)(blockIdx.x * 64 + threadIdx.x); seg < a.nseg; seg += (int)gridDim.x * 64) {
    BitReader r = {a.data, a.starts[seg], a.ends[seg], 0, 0};
    if (r.end > a.nbytes) r.end = a.nbytes;
    if (r.p > r.end) r.p = r.end;
    int last_dc[3] = {0, 0, 0};
    bool err = false;
    const int m_end = min((seg + 1) * ri, a.total_mcus);
    for (int m = seg * ri; m < m_end && !err; m++) {
      const int my = m / a.mcus_per_row, mx = m - my * a.mcus_per_row;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        if (c >= a.ncomp) break;
        const int hs = a.ncomp > 1 ? a.hs[c] : 1, vs = a.ncomp > 1 ? a.vs[c] : 1;
        const HuffDecTable& dct = s_t[c ? 2 : 0];
        const HuffDecTable& act = s_t[c ? 3 : 1];
        for (int yi = 0; yi < vs; yi++) {
          for (int xi = 0; xi < hs; xi++) {
            const int by = my * vs + yi, bx = mx * hs + xi;
            const bool real = by < a.bh[c] && bx < a.bw[c];
            int16_t* blk = a.coef[c] + ((size_t)(real ? by : 0) * a.bw[c] + (real ? bx : 0)) * 64;
            r.fill();
            int s = decode_symbol(r, dct, err);
            if (s > 15) { err = true; s = 0; }
            const int diff = s ? receive_extend(r, s) : 0;
            last_dc[c] += diff;
            if (real) blk[0] = (int16_t)last_dc[c];
            for (int k = 1; k < 64 && !err;) {
              r.fill();
              const int rs = decode_symbol(r, act, err);
              const int run = rs >> 4;
              s = rs & 15;
              if (s) {
                k += run;
                if (k > 63) { err = true; break; }
                const int v = receive_extend(r, s);
                if (real) blk[s_zz[k]] = (int16_t)v;
                k++;
              } else if (run == 15) {
                k += 16;
              } else {
                break;
              }
            }
          }
        }
      }
    }
    if (err) atomicOr(a.status, 2u);
  }
}

}  // namespace

int huff_marker_chunks(uint64_t nbytes) { return (int)((nbytes + kChunk - 1) / kChunk); }

// counts: huff_marker_chunks(nbytes) words of scratch; starts / ends: nseg words each; status: status[0] flags, status[1] marker count
hipError_t launch_huffman_decode(const HuffDecArgs& a, uint32_t* counts, uint32_t* starts, uint32_t* ends, hipStream_t s) {
  const int nchunks = huff_marker_chunks(a.nbytes);
  hipLaunchKernelGGL(huff_count_markers_kernel, dim3(nchunks), dim3(256), 0, s, a.data, a.nbytes, counts);
  hipLaunchKernelGGL(huff_scan_counts_kernel, dim3(1), dim3(1024), 0, s, counts, nchunks, a.status + 1);
  hipLaunchKernelGGL(huff_emit_intervals_kernel, dim3(nchunks), dim3(256), 0, s, a.data, a.nbytes, counts, a.nseg, starts, ends, a.status);
  const int grid = (a.nseg + 63) / 64;
  hipLaunchKernelGGL(huff_decode_kernel, dim3(grid < 4096 ? grid : 4096), dim3(64), 0, s, a);
  return hipGetLastError();
}

}  // namespace uhdr
