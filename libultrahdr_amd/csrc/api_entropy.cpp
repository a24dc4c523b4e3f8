// api_entropy.cpp -- baseline Huffman coding of coefficient blocks behind the C ABI (see api_internal.h).
#include "api_internal.h"

#include <condition_variable>
#include <functional>
#include <mutex>

// -------------------------------------------------------------------------------------------------
// JPEG entropy stage: baseline Huffman coding of coefficient blocks, one restart interval per wavefront
// -------------------------------------------------------------------------------------------------
uhdr_error_info_t uhdr_api::check_scan(const uhdr_hip_jpeg_scan_t* sc, bool need_coef, int* mcus_per_row, int* mcu_rows, int* blocks_per_mcu) {
  if (!sc) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for the scan description");
  if (sc->num_components != 1 && sc->num_components != 3)
    return err_status(UHDR_CODEC_INVALID_PARAM, "a scan has 1 or 3 components, received %d", sc->num_components);
  if (sc->w == 0 || sc->h == 0 || sc->w > 65535 || sc->h > 65535)
    return err_status(UHDR_CODEC_INVALID_PARAM, "image dimensions %ux%u are outside JPEG's 1..65535", sc->w, sc->h);
  int hmax = 1, vmax = 1, bpm = 0;
  for (int i = 0; i < sc->num_components; i++) {
    const int hs = sc->num_components == 1 ? 1 : sc->h_samp[i], vs = sc->num_components == 1 ? 1 : sc->v_samp[i];
    if (hs < 1 || hs > 2 || vs < 1 || vs > 2) return err_status(UHDR_CODEC_INVALID_PARAM, "component %d: sampling factors %dx%d not in {1, 2}", i, hs, vs);
    if (hs > hmax) hmax = hs;
    if (vs > vmax) vmax = vs;
    bpm += hs * vs;
    if (need_coef && (!sc->coef[i] || ((uintptr_t)sc->coef[i] & 15)))
      return err_status(UHDR_CODEC_INVALID_PARAM, "coefficient buffer %d is null or not 16-byte aligned", i);
    if (sc->blocks_w[i] < 1 || sc->blocks_h[i] < 1) return err_status(UHDR_CODEC_INVALID_PARAM, "component %d: empty block grid", i);
  }
  if (sc->num_components == 1) {  // non-interleaved: an MCU is one block (jcmaster.c per_scan_setup)
    *mcus_per_row = sc->blocks_w[0];
    *mcu_rows = sc->blocks_h[0];
    if ((unsigned)sc->blocks_w[0] != (sc->w + 7) / 8 || (unsigned)sc->blocks_h[0] != (sc->h + 7) / 8)
      return err_status(UHDR_CODEC_INVALID_PARAM, "a %dx%d block grid does not match a %ux%u image", sc->blocks_w[0], sc->blocks_h[0], sc->w, sc->h);
  } else {
    *mcus_per_row = (int)((sc->w + 8u * hmax - 1) / (8u * hmax));
    *mcu_rows = (int)((sc->h + 8u * vmax - 1) / (8u * vmax));
    for (int i = 0; i < 3; i++) {
      // jpeg_component_info::width_in_blocks (real blocks; libjpeg pads MCUs with dummy blocks) up to the MCU-padded grid
      const unsigned cw = (sc->w * sc->h_samp[i] + hmax - 1) / hmax, chh = (sc->h * sc->v_samp[i] + vmax - 1) / vmax;
      const int min_w = (int)((cw + 7) / 8), min_h = (int)((chh + 7) / 8);
      if (sc->blocks_w[i] < min_w || sc->blocks_w[i] > *mcus_per_row * sc->h_samp[i] || sc->blocks_h[i] < min_h ||
          sc->blocks_h[i] > *mcu_rows * sc->v_samp[i])
        return err_status(UHDR_CODEC_INVALID_PARAM, "component %d: a %dx%d block grid does not match a %ux%u image at %dx%d sampling", i,
                          sc->blocks_w[i], sc->blocks_h[i], sc->w, sc->h, sc->h_samp[i], sc->v_samp[i]);
    }
  }
  *blocks_per_mcu = bpm;
  return ok_status();
}

// Wait for everything enqueued on the context's stream.  hipStreamSynchronize blocks on the queue's completion signal (an interrupt and a
// scheduler wake-up); the entropy coder's waits sit at the end of sub-millisecond calls, so they poll the stream for a bounded while first
// (UHDR_HIP_SYNC_SPIN_US, default 2000; 0: the blocking wait only).
static hipError_t wait_stream(uhdr_hip_ctx* c) {
  static const int spin_us = [] { const char* e = getenv("UHDR_HIP_SYNC_SPIN_US"); return e ? atoi(e) : 2000; }();
  if (spin_us > 0) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t q = hipStreamQuery(c->stream);
      if (q == hipSuccess) return hipSuccess;
      if (q != hipErrorNotReady) return q;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
  }
  return hipStreamSynchronize(c->stream);
}


uhdr_error_info_t uhdr_hip_huffman_encode_dev(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_scan_t* sc, uint8_t* out, size_t out_capacity,
                                              size_t* out_bytes) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!out || !out_bytes) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for the output buffer or size");
  int mpr = 0, mrows = 0, bpm = 0;
  UHDR_TRY(check_scan(sc, true, &mpr, &mrows, &bpm));
  const bool stream = sc->restart_interval == 0;  // no restart markers: the reference's own stream (jpegencoderhelper.cpp:187-201)
  if (!stream && (sc->restart_interval < 1 || sc->restart_interval > 65535 || sc->restart_interval * bpm > 64))
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "restart_interval must be 0 (no markers) or in 1..%d for %d blocks per MCU (one wavefront "
                      "encodes one restart interval of at most 64 blocks); received %d", 64 / bpm, bpm, sc->restart_interval);
  if (stream && 2 * bpm > 64) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "%d blocks per MCU: outside the HIP path", bpm);
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_huff) {
    std::vector<uint32_t> blob(host::jpeg_huff_code_tables());
    blob.resize((size_t)host::kHuffTabWords + 16);
    memcpy(blob.data() + host::kHuffTabWords, host::jpeg_zigzag_to_natural(), 64);
    HIP_TRY(hipMalloc((void**)&c->d_huff, blob.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(c->d_huff, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  HuffArgs a;
  memset(&a, 0, sizeof a);
  a.ncomp = sc->num_components;
  for (int i = 0; i < a.ncomp; i++) {
    a.coef[i] = sc->coef[i];
    a.bw[i] = sc->blocks_w[i]; a.bh[i] = sc->blocks_h[i];
    a.hs[i] = a.ncomp == 1 ? 1 : sc->h_samp[i]; a.vs[i] = a.ncomp == 1 ? 1 : sc->v_samp[i];
  }
  a.mcus_per_row = mpr;
  a.total_mcus = mpr * mrows;
  a.ri = stream ? huff_stream_segment_mcus(bpm) : sc->restart_interval;
  a.blocks_per_mcu = bpm;
  a.nseg = (a.total_mcus + a.ri - 1) / a.ri;
  a.tables = c->d_huff;
  a.zigzag = (const uint8_t*)(c->d_huff + host::kHuffTabWords);
  if (stream) {
    // scratch[4]: the unstuffed stream (as many bytes as the caller's buffer holds: stuffing only adds bytes);
    // scratch[5]: segment starts (nseg + 1) | stuffed size | meta (4 words) | segment bit counts | chunk counts
    if (out_capacity > 0xFFFFFFF0u) out_capacity = 0xFFFFFFF0u;
    HuffStream t;
    memset(&t, 0, sizeof t);
    t.raw_words = ((uint64_t)out_capacity + 3) / 4 + 1;
    const uint64_t worst_words = (uint64_t)a.total_mcus * bpm * 52 + 2;  // 1660 bits per block at most (11 + 11 + 63 * 26)
    if (t.raw_words > worst_words) t.raw_words = worst_words;
    const int nchunks = huff_stuff_chunks(t.raw_words * 4u);
    UHDR_TRY(ensure(c->scratch[4], (size_t)t.raw_words * 4));
    UHDR_TRY(ensure(c->scratch[5], ((size_t)a.nseg + 2) * sizeof(uint64_t) + (4 + (size_t)a.nseg + (size_t)nchunks) * sizeof(uint32_t)));
    t.raw = (uint32_t*)c->scratch[4].p;
    t.seg_start = (uint64_t*)c->scratch[5].p;
    uint64_t* d_total = t.seg_start + (size_t)a.nseg + 1;
    t.meta = (uint32_t*)(d_total + 1);
    t.seg_bits = t.meta + 4;
    uint32_t* chunk_counts = t.seg_bits + a.nseg;
    {
      ProfScope ps(c, "huffman_encode");
      HIP_TRY(launch_huffman_encode_stream(a, t, chunk_counts, d_total, out, (uint64_t)out_capacity, c->stream));
    }
    c->stats.entropy_encode_stream++;
    // the stuffed size and the meta words sit next to each other: ONE copy into pinned memory (round 6: were two copies into the stack,
    // i.e. two staged pageable transfers)
    if (!c->h_flags) HIP_TRY(hipHostMalloc((void**)&c->h_flags, 64 * sizeof(uint32_t), hipHostMallocDefault));
    uint32_t* hp = c->h_flags + 48;
    HIP_TRY(hipMemcpyAsync(hp, d_total, sizeof(uint64_t) + 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(wait_stream(c));
    uint64_t total = 0;
    uint32_t meta[4] = {0, 0, 0, 0};
    memcpy(&total, hp, sizeof total);
    memcpy(meta, hp + 2, sizeof meta);
    if (meta[2]) return err_status(UHDR_CODEC_INVALID_PARAM, "coefficients outside the baseline range (DC difference beyond 11 bits / AC beyond 10 bits)");
    const uint64_t raw_bytes = ((((uint64_t)meta[1] << 32) | meta[0]) + 7) / 8;
    if (raw_bytes > t.raw_words * 4u) total = raw_bytes + raw_bytes / 64 + 64;  // the stuffing passes saw a truncated stream: ask for room to spare
    *out_bytes = (size_t)total;
    if (total > out_capacity)
      return err_status(UHDR_CODEC_MEM_ERROR, "entropy-coded data needs %llu bytes, the output buffer holds %zu", (unsigned long long)total, out_capacity);
    return ok_status();
  }
  a.slot_stride = huff_slot_stride();
  // scratch: interval slots | interval sizes | offsets (nseg + 1) | status
  UHDR_TRY(ensure(c->scratch[4], (size_t)a.nseg * a.slot_stride));
  const size_t meta = (size_t)a.nseg * sizeof(uint32_t) + 16 + ((size_t)a.nseg + 1) * sizeof(uint64_t) + 16;
  UHDR_TRY(ensure(c->scratch[5], meta));
  a.slots = (uint8_t*)c->scratch[4].p;
  uint64_t* offsets = (uint64_t*)c->scratch[5].p;                       // 8-byte aligned first
  uint32_t* status = (uint32_t*)(offsets + (size_t)a.nseg + 1);
  a.seg_bytes = status + 2;
  {
    ProfScope ps(c, "huffman_encode");
    HIP_TRY(launch_huffman_encode(a, offsets, status, out, (uint64_t)out_capacity, c->stream));
  }
  c->stats.entropy_encode_intervals++;
  uint64_t total = 0;
  uint32_t bad = 0;
  HIP_TRY(hipMemcpyAsync(&total, offsets + a.nseg, sizeof total, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&bad, status, sizeof bad, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(wait_stream(c));
  if (bad) return err_status(UHDR_CODEC_INVALID_PARAM, "coefficients outside the baseline range (DC difference beyond 11 bits / AC beyond 10 bits)");
  *out_bytes = (size_t)total;
  if (total > out_capacity)
    return err_status(UHDR_CODEC_MEM_ERROR, "entropy-coded data needs %llu bytes, the output buffer holds %zu", (unsigned long long)total, out_capacity);
  return ok_status();
}

// jdhuff.c jpeg_make_d_derived_tbl: decode form of one DHT table; false for an invalid table
static bool make_dec_table(const uint8_t bits[17], const uint8_t vals[256], HuffDecTable* t) {
  memset(t, 0, sizeof *t);
  int nsym = 0;
  for (int l = 1; l <= 16; l++) nsym += bits[l];
  if (nsym > 256) return false;
  int code = 0, k = 0;
  for (int l = 1; l <= 16; l++) {
    if (bits[l]) {
      t->valoff[l] = k - code;
      for (int i = 0; i < bits[l]; i++, k++, code++) {
        if (code >= (1 << l)) return false;  // over-subscribed
        if (l <= 9) {
          const int lo = code << (9 - l);
          for (int x = 0; x < (1 << (9 - l)); x++) t->lut[lo + x] = (uint16_t)((l << 8) | vals[k]);
        }
      }
      t->maxcode[l] = code - 1;
    } else {
      t->maxcode[l] = -1;
    }
    code <<= 1;
  }
  t->maxcode[17] = 0x7fffffff;
  memcpy(t->vals, vals, 256);
  return true;
}

// two-level form for the self-synchronising decoder; false when the table needs more than kHuffL2Max sub-tables
static bool make_fast_table(const uint8_t bits[17], const uint8_t vals[256], HuffFastTable* t, int* nsub_out = nullptr) {
  memset(t, 0, sizeof *t);
  int code = 0, k = 0, nsub = 0;
  int sub_of[512];
  for (int i = 0; i < 512; i++) sub_of[i] = -1;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < bits[l]; i++, k++, code++) {
      if (code >= (1 << l) || k >= 256) return false;
      if (l <= 9) {
        const int lo = code << (9 - l);
        for (int x = 0; x < (1 << (9 - l)); x++) t->l1[lo + x] = (uint16_t)((l << 8) | vals[k]);
      } else {
        const int prefix = code >> (l - 9);
        if (sub_of[prefix] < 0) {
          if (nsub >= kHuffL2Max) return false;
          sub_of[prefix] = nsub++;
          t->l1[prefix] = (uint16_t)(0x8000u | (unsigned)sub_of[prefix]);
        }
        const int rest = (code << (16 - l)) & 127;
        for (int x = 0; x < (1 << (16 - l)); x++) t->l2[sub_of[prefix]][rest + x] = (uint16_t)((l << 8) | vals[k]);
      }
    }
    code <<= 1;
  }
  if (nsub_out) *nsub_out = nsub;
  return true;
}

// The state-tracking form of a fast table (huffman_decode_sync.hip: track_span): an entry says how many bits the symbol
// consumes (code + magnitude bits) and how far the zig-zag index moves -- exactly decode_step's transitions:
//   DC symbol (size category s):  bits = len + s, advance 1
//   AC symbol run/size:           bits = len + s, advance run + 1;  ZRL: len, 16;  EOB: len, 64 (to the end of the block)
//   undefined code:               16 bits; advance 64 (AC) / 1 (DC)
// l1 entries of long codes keep the 0x8000 | sub-table form.
static void make_track_table(const HuffFastTable& f, bool is_dc, HuffFastTable* t) {
  auto conv = [&](uint16_t e) -> uint16_t {
    if (e & 0x8000u) return e;
    unsigned adv, kinc;
    if (e == 0) {
      adv = 16;
      kinc = is_dc ? 1 : 64;
    } else {
      const unsigned len = (e >> 8) & 31u, rs = e & 255u;
      if (is_dc) {
        const unsigned sz = rs > 15u ? 0u : rs;  // decode_step: a category beyond 15 is "bad", no magnitude bits
        adv = len + sz;
        kinc = 1;
      } else {
        const unsigned sz = rs & 15u, run = rs >> 4;
        adv = len + sz;
        kinc = sz ? run + 1 : (run == 15u ? 16u : 64u);
      }
    }
    return (uint16_t)(adv | (kinc << 5));
  };
  for (int i = 0; i < 512; i++) t->l1[i] = conv(f.l1[i]);
  for (int s = 0; s < kHuffL2Max; s++)
    for (int i = 0; i < 128; i++) t->l2[s][i] = conv(f.l2[s][i]);
}

// The value form of a fast table for the write pass (huffman_decode_sync.hip: write_span): the tracking form's fields plus
// the number of magnitude bits and a "malformed" flag, one 32-bit word per entry, first level then the sub-tables.
static void make_value_table(const HuffFastTable& f, bool is_dc, uint32_t* out /* kHuffValWords */) {
  auto conv = [&](uint16_t e, bool first_level) -> uint32_t {
    if (first_level && (e & 0x8000u)) return 0x80000000u | (e & 31u);
    unsigned adv, kinc, sz = 0, bad = 0;
    if (e == 0) {
      adv = 16;
      kinc = is_dc ? 1 : 64;
      bad = 1;
    } else {
      const unsigned len = (e >> 8) & 31u, rs = e & 255u;
      if (is_dc) {
        if (rs > 15u) bad = 1; else sz = rs;
        adv = len + sz;
        kinc = 1;
      } else {
        sz = rs & 15u;
        const unsigned run = rs >> 4;
        adv = len + sz;
        kinc = sz ? run + 1 : (run == 15u ? 16u : 64u);
      }
    }
    return adv | (kinc << 5) | (sz << 12) | (bad << 16);
  };
  for (int i = 0; i < 512; i++) out[i] = conv(f.l1[i], true);
  for (int s = 0; s < kHuffL2Max; s++)
    for (int i = 0; i < 128; i++) out[512 + s * 128 + i] = conv(f.l2[s][i], false);
}

// The pair form of a tracking table (uhdr_types.h: kHuffPairBits; huffman_decode_sync.hip: track_span_pair).  f / t: the table in
// symbol and in tracking form; t_ac: the AC table of the same component in tracking form, f_ac in symbol form -- the table the symbol AFTER
// a DC or AC symbol of this table is read with, as long as the block does not end (the kernel checks that, and that the
// subsequence does not end, before it takes the second symbol).  A second symbol is entered only when its CODE lies wholly inside
// the index bits that the first symbol leaves (a prefix code: the bits beyond cannot change which code it is).
static void make_pair_table(const HuffFastTable& f, const HuffFastTable& t, const HuffFastTable& f_ac, const HuffFastTable& t_ac, int l2_base, uint32_t* out /* kHuffPairWords */) {
  constexpr int B = kHuffPairBits;
  for (int idx = 0; idx < kHuffPairWords; idx++) {
    const int p9 = idx >> (B - 9);
    const uint16_t e = f.l1[p9];
    uint32_t first;  // tracking form of the first symbol: bits | advance << 5
    bool defined;
    if (e & 0x8000u) {
      const int sub = e & 31, rest = (idx & ((1 << (B - 9)) - 1)) << (16 - B);  // the index bits beyond the ninth, as the top bits of the 7-bit remainder
      const uint16_t e2 = f.l2[sub][rest];
      if (e2 == 0 || (int)((e2 >> 8) & 31u) > B) {  // not decided by the index bits
        out[idx] = 0x80000000u | (uint32_t)(l2_base + sub);  // (the index into the packed sub-tables of all four tables)
        continue;
      }
      first = t.l2[sub][rest];
      defined = true;
    } else {
      first = t.l1[p9];
      defined = e != 0;
    }
    uint32_t word = first & 0xfffu;
    const int adv1 = (int)(first & 31u), kinc1 = (int)((first >> 5) & 127u);
    if (defined && kinc1 < 64 && adv1 < B) {
      const int r = B - adv1;  // index bits left for the second code
      const int next9 = ((idx << adv1) & (kHuffPairWords - 1)) >> (B - 9);
      const uint16_t n = f_ac.l1[next9];
      if (n != 0 && !(n & 0x8000u) && (int)((n >> 8) & 31u) <= r) {
        const uint32_t second = t_ac.l1[next9];
        word |= (second & 31u) << 12 | ((second >> 5) & 127u) << 17;
      }
    }
    out[idx] = word;
  }
}

// ... and the pair form of the VALUE tables (write form 2, huffman_decode_sync.hip: write_span2): two words per index, the first symbol
// and the AC symbol behind it (0: none), each in make_value_table's form.  v / v_ac: the value forms (kHuffValWords words: first
// level, then the sub-tables).  A first symbol that is malformed, or whose code is longer than the index, is marked 0x80000000 | sub-table
// (its first-level word when the code has no sub-table: the kernel then reads the one-symbol entry the slow way).
static void make_pair_value_table(const HuffFastTable& f, const uint32_t* v, const HuffFastTable& f_ac, const uint32_t* v_ac, int l2_base, uint32_t* out /* 2 x kHuffPairWords */) {
  constexpr int B = kHuffPairBits;
  for (int idx = 0; idx < kHuffPairWords; idx++) {
    const int p9 = idx >> (B - 9);
    const uint16_t e = f.l1[p9];
    uint32_t first;
    bool defined;
    if (e & 0x8000u) {
      const int sub = e & 31, rest = (idx & ((1 << (B - 9)) - 1)) << (16 - B);
      const uint16_t e2 = f.l2[sub][rest];
      if (e2 == 0 || (int)((e2 >> 8) & 31u) > B) {
        out[2 * idx] = 0x80000000u | (uint32_t)(l2_base + sub);
        out[2 * idx + 1] = 0;
        continue;
      }
      first = v[512 + sub * 128 + rest];
      defined = true;
    } else {
      first = v[p9];
      defined = e != 0;
    }
    uint32_t second = 0;
    const int adv1 = (int)(first & 31u), kinc1 = (int)((first >> 5) & 127u);
    if (defined && !((first >> 16) & 1u) && kinc1 < 64 && adv1 < B) {
      const int r = B - adv1;
      const int next9 = ((idx << adv1) & (kHuffPairWords - 1)) >> (B - 9);
      const uint16_t n = f_ac.l1[next9];
      if (n != 0 && !(n & 0x8000u) && (int)((n >> 8) & 31u) <= r) second = v_ac[next9];
    }
    out[2 * idx] = first;
    out[2 * idx + 1] = second;
  }
}

uhdr_error_info_t uhdr_hip_huffman_decode_dev(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_scan_t* sc, const uhdr_hip_huff_tables_t* tables,
                                              const uint8_t* data, size_t data_bytes) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!data || data_bytes == 0 || data_bytes > 0xFFFFFFF0ull) return err_status(UHDR_CODEC_INVALID_PARAM, "received no (or more than 4 GiB of) entropy-coded data");
  int mpr = 0, mrows = 0, bpm = 0;
  UHDR_TRY(check_scan(sc, true, &mpr, &mrows, &bpm));
  if (sc->restart_interval < 0 || sc->restart_interval > 65535) return err_status(UHDR_CODEC_INVALID_PARAM, "restart_interval %d out of range", sc->restart_interval);
  HIP_TRY(hipSetDevice(c->device));
  if (!c->d_huff) {  // the zig-zag map lives behind the encoder's code tables
    std::vector<uint32_t> blob(host::jpeg_huff_code_tables());
    blob.resize((size_t)host::kHuffTabWords + 16);
    memcpy(blob.data() + host::kHuffTabWords, host::jpeg_zigzag_to_natural(), 64);
    HIP_TRY(hipMalloc((void**)&c->d_huff, blob.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpyAsync(c->d_huff, blob.data(), blob.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  const DbgClock dbg;
  // the five decode forms of the file's tables, resident on the device (huff_tabs: rebuilt only when the DHT bytes change)
  constexpr size_t kTabDec = 0, kTabFast = (sizeof(HuffDecTable) * 4 + 255) & ~(size_t)255,
                   kTabVal = (kTabFast + sizeof(HuffFastTable) * 8 + 255) & ~(size_t)255, kTabPair = kTabVal + (size_t)4 * kHuffValWords * 4,
                   kTabPairVal = kTabPair + (size_t)kHuffPairBlobWords * 4, kTabBytes = kTabPairVal + (size_t)kHuffPairValBlobWords * 4;
  {
    uint8_t key[4 * (17 + 256)];
    for (int t = 0; t < 4; t++) {
      uint8_t* bits = key + (size_t)t * (17 + 256);
      uint8_t* vals = bits + 17;
      if (tables) {
        memcpy(bits, tables->bits[t], 17);
        memcpy(vals, tables->vals[t], 256);
      } else {
        memset(bits, 0, 17 + 256);
        host::jpeg_std_huff_table(t & 1, t >> 1, bits, vals);
      }
    }
    uhdr_hip_ctx::HuffTabCache& hc = c->huff_tabs;
    if (!hc.valid || memcmp(hc.key, key, sizeof key) != 0) {
      hc.valid = false;
      std::vector<HuffDecTable> tabs(4);
      std::vector<HuffFastTable> ftabs(8);
      std::vector<uint32_t> vtabs((size_t)4 * kHuffValWords), ptabs((size_t)kHuffPairBlobWords), pvtabs((size_t)kHuffPairValBlobWords);
      bool fast_ok = true;
      int l2_count[4] = {0, 0, 0, 0};
      for (int t = 0; t < 4; t++) {
        const uint8_t* bits = key + (size_t)t * (17 + 256);
        if (!make_dec_table(bits, bits + 17, &tabs[(size_t)t])) return err_status(UHDR_CODEC_INVALID_PARAM, "Huffman table %d is not a valid DHT table", t);
        fast_ok = make_fast_table(bits, bits + 17, &ftabs[(size_t)t], &l2_count[t]) && fast_ok;
      }
      int l2_base[4] = {0, 0, 0, 0};
      for (int t = 1; t < 4; t++) l2_base[t] = l2_base[t - 1] + l2_count[t - 1];
      if (l2_base[3] + l2_count[3] > kHuffL2Total) fast_ok = false;  // more long-code prefixes than the packed sub-table array holds
      if (fast_ok)
        for (int t = 0; t < 4; t++) {
          make_track_table(ftabs[(size_t)t], (t & 1) == 0, &ftabs[(size_t)t + 4]);
          make_value_table(ftabs[(size_t)t], (t & 1) == 0, vtabs.data() + (size_t)t * kHuffValWords);
        }
      if (fast_ok)
        for (int t = 0; t < 4; t++)  // (after the loop above: the AC tables' tracking forms exist)
        {
          make_pair_table(ftabs[(size_t)t], ftabs[(size_t)t + 4], ftabs[(size_t)(t | 1)], ftabs[(size_t)(t | 1) + 4], l2_base[t], ptabs.data() + (size_t)t * kHuffPairWords);
          make_pair_value_table(ftabs[(size_t)t], vtabs.data() + (size_t)t * kHuffValWords, ftabs[(size_t)(t | 1)], vtabs.data() + (size_t)(t | 1) * kHuffValWords,
                                l2_base[t], pvtabs.data() + (size_t)t * 2 * kHuffPairWords);
          // the second levels behind them, 16 bits per entry (the kernels' LDS structs are copies of these blobs)
          uint16_t* l2t = (uint16_t*)(ptabs.data() + (size_t)4 * kHuffPairWords) + (size_t)l2_base[t] * 128;
          uint16_t* l2v = (uint16_t*)(pvtabs.data() + (size_t)8 * kHuffPairWords) + (size_t)l2_base[t] * 128;
          for (int i = 0; i < l2_count[t] * 128; i++) {
            l2t[i] = ftabs[(size_t)t + 4].l2[i / 128][i % 128];
            const uint32_t v = vtabs[(size_t)t * kHuffValWords + 512 + (size_t)i];
            l2v[i] = (uint16_t)(((v >> 16) & 1u) ? 0u : (v & 0xffffu));
          }
        }
      UHDR_TRY(ensure(hc.dev, kTabBytes));
      HIP_TRY(hipStreamSynchronize(c->stream));  // nothing in flight reads the old tables
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabDec, tabs.data(), sizeof(HuffDecTable) * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabFast, ftabs.data(), sizeof(HuffFastTable) * 8, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabVal, vtabs.data(), vtabs.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabPair, ptabs.data(), ptabs.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabPairVal, pvtabs.data(), pvtabs.size() * 4, hipMemcpyHostToDevice));
      memcpy(hc.key, key, sizeof key);
      hc.fast_ok = fast_ok;
      hc.valid = true;
    }
  }
  const bool fast_ok = c->huff_tabs.fast_ok;
  const uint8_t* tabs_dev = (const uint8_t*)c->huff_tabs.dev.p;
  if (!c->h_flags) HIP_TRY(hipHostMalloc((void**)&c->h_flags, 64 * sizeof(uint32_t), hipHostMallocDefault));
  HuffDecArgs a;
  memset(&a, 0, sizeof a);
  a.ncomp = sc->num_components;
  size_t zero_bytes[3] = {0, 0, 0};
  for (int i = 0; i < a.ncomp; i++) {
    a.coef[i] = const_cast<int16_t*>(sc->coef[i]);
    a.bw[i] = sc->blocks_w[i]; a.bh[i] = sc->blocks_h[i];
    a.hs[i] = a.ncomp == 1 ? 1 : sc->h_samp[i]; a.vs[i] = a.ncomp == 1 ? 1 : sc->v_samp[i];
    zero_bytes[i] = (size_t)a.bw[i] * a.bh[i] * 64 * sizeof(int16_t);
  }
  a.mcus_per_row = mpr;
  a.total_mcus = mpr * mrows;
  a.ri = sc->restart_interval;
  a.nseg = a.ri > 0 ? (a.total_mcus + a.ri - 1) / a.ri : 1;
  a.data = data;
  a.nbytes = (uint32_t)data_bytes;
  a.zigzag = (const uint8_t*)(c->d_huff + host::kHuffTabWords);
  // scratch: status[4] | chunk counts | starts | ends
  const int nchunks = huff_marker_chunks(data_bytes);
  const size_t need = 16 + ((size_t)nchunks + 2 * (size_t)a.nseg) * sizeof(uint32_t);
  UHDR_TRY(ensure(c->scratch[5], need));
  uint8_t* base = (uint8_t*)c->scratch[5].p;
  a.tabs = (const HuffDecTable*)(tabs_dev + kTabDec);
  a.status = (uint32_t*)base;
  uint32_t* counts = a.status + 4;
  uint32_t* starts = counts + nchunks;
  uint32_t* ends = starts + a.nseg;
  a.starts = starts;
  a.ends = ends;
  dbg.mark("huffman_decode_dev: tables ready");
  const bool rst_sync = a.nseg > 1 && data_bytes / (size_t)a.nseg >= 320 && !getenv("UHDR_HIP_HUFF_RST_INTERVALS");
  const bool try_sync = (a.nseg == 1 || rst_sync) && data_bytes >= 4096 && data_bytes < ((size_t)1 << 29) && fast_ok && !getenv("UHDR_HIP_HUFF_SERIAL") && bpm <= 16;
  // write pass, form 2 (marker-less scans): a scan-order scratch takes the zero fill, the JBLOCK arrays are written whole
  const int write_form = [] { const char* e = getenv("UHDR_HIP_HUFF_WRITE"); return e ? atoi(e) : 2; }();  // (read per call: tools/huff_exp.py sweeps it)
  const bool form2 = try_sync && !rst_sync && write_form != 1;
  bool coef_zeroed = !form2;  // the interval / single-lane decoder below stores into zero-initialised arrays
  if (!form2)
    for (int i = 0; i < a.ncomp; i++) HIP_TRY(hipMemsetAsync(a.coef[i], 0, zero_bytes[i], c->stream));
  // A scan without restart markers (every file the reference writes) is ONE interval: the per-interval kernel would
  // decode it on a single lane.  The self-synchronising decoder (huffman_decode_sync.hip) parallelises it; should its
  // fixed-point search not settle within the round budget (never seen; pathological streams), the serial kernel runs.
  bool sync_done = false;
  // (the self-synchronising decoder keeps bit positions in 32 bits: scans of 512 MiB and more take the other routes)
  // Restart-marker files take the same decoder when their intervals are long enough that one lane per interval leaves the
  // device idle (3240 intervals of 1 KB in a 4K file with ri = 10: 1081 us on 51 wavefronts): the unstuff pass drops the
  // markers, the decoders hop over the padding bits at the flagged interval starts and the DC scan starts over there
  // (huffman_decode_sync.hip: restart_jump).  Short intervals (a few hundred bytes) are faster one lane each.
  if (try_sync) {
    // Subsequence size: a power of two >= 256 bits (the lanes' chunks are staged in LDS: 64 x sub_bits / 8 bytes per wave).
    // Attempts, in order: the hypothesis scheme with seven (4:2:0; up to fifteen for fewer blocks per MCU) overflow levels at 512 bits (4K q95 photo-like data: 390 us) -- denser
    // streams start at 2048 / 4096 bits --, then at doubled sizes up to 4096 bits, then the rounds at 1024 bits.
    struct Attempt { uint32_t sub_bits; int levels; };  // levels 0: the rounds
    std::vector<Attempt> attempts;
    bool sparse = false;  // under 64 bits per block: long subsequences, many levels -- the stragglers' waves take over after ONE lockstep level
    {
      const char* eb = getenv("UHDR_HIP_HUFF_SUB_BITS");
      const char* el = getenv("UHDR_HIP_HUFF_LEVELS");
      const int vb = eb ? atoi(eb) : 0;
      const bool vb_ok = vb >= 256 && vb <= 4096 && (vb & (vb - 1)) == 0;
      if (vb_ok || el) {  // tuning / tests: exactly this configuration, then the rounds at the same size
        const uint32_t sbits = vb_ok ? (uint32_t)vb : 1024u;
        const int lv = el ? atoi(el) : (sbits <= 512 ? 7 : 4);
        if (lv >= 1 && lv <= 15 && bpm * (lv + 1) <= kHuffHypSlots) attempts.push_back({sbits, lv});
        attempts.push_back({sbits, 0});
      } else {
        // the window a path gets to fall in step (levels x subsequence) must cover the stream's synchronisation distance, which
        // grows with the bits per block (few EOBs in dense blocks): start where files of this density have settled, then widen
        const uint64_t bits_per_block = (uint64_t)data_bytes * 8u / (uint64_t)((uint64_t)a.total_mcus * (uint64_t)bpm);
        // Overflow levels: as many as the slots allow, up to 15.  Every further level is one more FRESH path a straggler can fall in
        // step with, and only stragglers pay for it (a path stops at its merge).  Smooth content is where it matters (a gain map:
        // 49 bits per block, a few short symbols each): seven trials lost the true path of a 4K three-channel map at 512 and at
        // 1024 bits (249 and 13 unmerged paths of 110 K), fifteen at 1024 do not.  Exactly flat regions -- a periodic bit pattern --
        // were the suspect and are not the problem (tools/flat_streams.py: one attempt each).
        const int lv_fit = kHuffHypSlots / bpm - 1;
        const int lv = lv_fit >= 15 ? 15 : (lv_fit >= 11 ? 11 : (lv_fit >= 7 ? 7 : (lv_fit >= 4 ? 4 : 0)));
        // ... and very sparse streams (under 64 bits per block: smooth content) start at 1024 bits for the same reason:
        // the 4K map above (49 bits per block) still loses its true path at 512 x 15 (8 unmerged paths) and settles at 1024.
        // A context also remembers the size a scan of the same shape and density settled at when the first attempt was lost.
        sparse = bits_per_block < 64;
        uint32_t first = bits_per_block < 64 ? 1024u : (bits_per_block < 200 ? 512u : (bits_per_block < 400 ? 2048u : 4096u));
        const uhdr_hip_ctx::HuffHint& hint = c->huff_hint[bpm & 15];
        if (hint.sub_bits > first && bits_per_block * 4 >= hint.bits_per_block * 3 && bits_per_block * 4 <= hint.bits_per_block * 5) first = hint.sub_bits;
        if (lv > 0)
          for (uint32_t sbits = first; sbits <= 4096u; sbits <<= 1) attempts.push_back({sbits, lv});
        attempts.push_back({1024u, 0});
      }
    }
    uint32_t sub_bits = attempts[0].sub_bits;  // the smallest size of the list: it sizes the per-subsequence buffers
    bool use_hyp = false;
    for (const Attempt& t : attempts) { if (t.sub_bits < sub_bits) sub_bits = t.sub_bits; use_hyp = use_hyp || t.levels > 0; }
    const int max_rounds = 40;
    const int nch = huff_sync_chunks(data_bytes);
    const uint32_t nsub = huff_sync_max_subsequences(data_bytes, sub_bits);
    const uint32_t total_blocks = (uint32_t)a.total_mcus * (uint32_t)bpm;
    // scratch[6]: clean | chunk counts | [flags | nblk | dcd | restart map: zeroed with ONE fill] | [state 0 | hypothesis map: one 0xff
    // fill] | state 1 | ...  (a fill is a 4 us launch of its own on the stream: eleven of them were 45 us of a 395 us decode)
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    const size_t rst_words = data_bytes / 32 + 2;
    const size_t o_clean = take(data_bytes + 16), o_cnt = take((size_t)nch * 4);
    const size_t o_flags = take(128), o_nblk = take((size_t)nsub * 4 + 4), o_dcd = take((size_t)total_blocks * 4), o_rm = rst_sync ? take(rst_words * 4) : 0;
    const size_t zero_bytes_sync = off - o_flags;
    // hypothesis scheme (interleaved scans): one decode per possible block position instead of rounds
    const size_t o_s0 = take((size_t)nsub * 8), o_hm = use_hyp ? take((size_t)nsub * kHuffHypSlots) : 0;
    const size_t ff_bytes = off - o_s0;
    const size_t o_s1 = take((size_t)nsub * 8), o_c0 = take(nsub), o_c1 = take(nsub), o_dcp = take((size_t)((total_blocks + 255) / 256) * 12 + 16),  // (form 1: chunks of 1024; form 2: huff_place_chunk() = 256)
                 o_ft = take(sizeof(HuffFastTable) * 8),  // symbol form x 4, state-tracking form x 4
                 o_st = take(((size_t)nsub / 2048 + 2) * 4), o_vt = take((size_t)4 * kHuffValWords * 4);
    const size_t o_hs = use_hyp ? take((size_t)nsub * kHuffHypSlots * 8) : 0, o_hc = use_hyp ? take((size_t)nsub * kHuffHypSlots * 2) : 0;
    size_t chain_tiles_off = 0;
    const size_t o_ch = use_hyp ? take(huff_hyp_chain_bytes(data_bytes, sub_bits, &chain_tiles_off)) : 0;  // sized for the smallest subsequences
    (void)chain_tiles_off;
    const size_t o_ds = rst_sync ? take((size_t)a.nseg * 12) : 0, o_rp = rst_sync ? take((size_t)nch * 12) : 0;
    // round 5: the straggler list of pass 1 (every path could end up on it) and the scan-order coefficient scratch of write form 2
    const size_t o_sl = use_hyp ? take((size_t)nsub * (size_t)bpm * 8) : 0;
    // round 6: write-pass pieces (HuffSyncArgs::pieces): up to four pieces per subsequence, for marker-less scans through the hypothesis scheme
    const int max_pieces = [&] {
      const char* e = getenv("UHDR_HIP_HUFF_PIECES");
      const int v = e ? atoi(e) : 4;
      return (use_hyp && form2 && !rst_sync && v >= 1 && v <= 4) ? v : 1;
    }();
    const size_t o_ms = max_pieces > 1 ? take((size_t)nsub * kHuffHypSlots * (size_t)(max_pieces - 1) * 8) : 0;
    const size_t o_mc = max_pieces > 1 ? take((size_t)nsub * kHuffHypSlots * (size_t)(max_pieces - 1) * 2) : 0;
    const size_t o_pe = max_pieces > 1 ? take((size_t)nsub * (size_t)max_pieces * 8 + 64) : 0;
    const size_t o_pc = max_pieces > 1 ? take(((size_t)nsub * (size_t)max_pieces + 4) * 4) : 0;
    const size_t o_st2 = max_pieces > 1 ? take(((size_t)nsub * (size_t)max_pieces / 2048 + 4) * 4) : 0;
    const size_t scan_bytes = form2 ? (size_t)total_blocks * 64 * sizeof(int16_t) : 0;
    const size_t o_cs = form2 ? take(scan_bytes + 256) : 0;
    UHDR_TRY(ensure(c->scratch[6], off));
    uint8_t* sb = (uint8_t*)c->scratch[6].p;
    HuffSyncArgs y;
    memset(&y, 0, sizeof y);
    y.clean = sb + o_clean;
    y.nbytes = (uint32_t)data_bytes;
    y.flags = (uint32_t*)(sb + o_flags);
    y.nstuffed = y.flags + 8;
    y.sub_bits = sub_bits;
    y.state[0] = (uint64_t*)(sb + o_s0); y.state[1] = (uint64_t*)(sb + o_s1);
    y.changed[0] = sb + o_c0; y.changed[1] = sb + o_c1;
    y.nblk = (uint32_t*)(sb + o_nblk);
    y.scan_tmp = (uint32_t*)(sb + o_st);
    y.dcd = (int*)(sb + o_dcd);
    y.total_blocks = total_blocks;
    y.blocks_per_mcu = bpm; y.ncomp = a.ncomp; y.mcus_per_row = a.mcus_per_row;
    uint32_t* rst_map = nullptr;
    if (rst_sync) {
      rst_map = (uint32_t*)(sb + o_rm);
      y.rst_map = rst_map;
      y.rst_blocks = (uint32_t)a.ri * (uint32_t)bpm;
      y.dc_seg = (int*)(sb + o_ds);
      y.rst_partial = (uint32_t*)(sb + o_rp);
      y.rst_chunks = (uint32_t)nch;
    }
    int j = 0;
    for (int i = 0; i < a.ncomp; i++) {
      y.bw[i] = a.bw[i]; y.bh[i] = a.bh[i]; y.hs[i] = a.hs[i]; y.vs[i] = a.vs[i]; y.coef[i] = a.coef[i];
      y.first_blk[i] = j;
      for (int k = 0; k < a.hs[i] * a.vs[i] && j < 16; k++) y.comp_of[j++] = (uint8_t)i;
    }
    y.ftabs = (const HuffFastTable*)(tabs_dev + kTabFast);
    y.ttabs = y.ftabs + 4;
    y.vtabs = (const uint32_t*)(tabs_dev + kTabVal);
    y.ptabs = (const uint32_t*)(tabs_dev + kTabPair);
    y.pvtabs = (const uint32_t*)(tabs_dev + kTabPairVal);
    y.zigzag = a.zigzag;
    (void)o_ft; (void)o_vt;
    if (use_hyp) {
      y.strag_list = (uint32_t*)(sb + o_sl);
      y.strag_cap = nsub * (uint32_t)bpm;
    }
    if (form2) y.coef_scan = (int16_t*)(sb + o_cs);
    // lockstep levels of pass 1 before the stragglers get a wave each (0: all levels in lockstep, the round-4 form); restart files keep the lockstep form
    // (a lane walks a 1024-bit subsequence in ~33 us, a straggler's wave in ~10: 4K three-channel gain map 634 us with all 15
    // levels in lockstep, 517 with two, 491 with one; the 4:2:0 base image's 512-bit levels are cheap in lockstep once compacted)
    const int main_levels_env = [&] { const char* e = getenv("UHDR_HIP_HUFF_MAIN_LEVELS"); return e ? atoi(e) : (sparse ? 1 : 2); }();
    if (j == bpm && bpm <= 16) {
      // Round 6: a first attempt of the hypothesis scheme gets its buffers' initial state from the decode's own first kernel (unstuff_count_kernel,
      // HuffInitFill): flags, nblk, dcd and the restart map zero, state 0 and the hypothesis map 0xff.  The rounds scheme and every retry use fills.
      const bool fills_in_unstuff = attempts[0].levels > 0 && zero_bytes_sync % 16 == 0 && ff_bytes % 16 == 0 && zero_bytes_sync / 16 < 0xFFFFFFFFull &&
                                    ff_bytes / 16 < 0xFFFFFFFFull && !getenv("UHDR_HIP_HUFF_FILL_LAUNCHES");
      HuffInitFill init_fill;
      memset(&init_fill, 0, sizeof init_fill);
      if (fills_in_unstuff) {
        init_fill.zero_ptr = (uint4*)y.flags;
        init_fill.zero_vec = (uint32_t)(zero_bytes_sync / 16);
        init_fill.ff_ptr = (uint4*)y.state[0];
        init_fill.ff_vec = (uint32_t)(ff_bytes / 16);
      } else {
        HIP_TRY(hipMemsetAsync(y.flags, 0, zero_bytes_sync, c->stream));  // flags, nblk, dcd and the restart map
      }
      // form 2's scratch: zero-filled by pass 0 of a hypothesis attempt itself (round 6: a 25-50 MB fill was a launch of its own at the head of
      // the decode); the rounds scheme gets a fill
      auto pass0_zeroes = [&](const Attempt& t) { return form2 && t.levels > 0 && scan_bytes % 16 == 0 && scan_bytes / 16 < 0xFFFFFFFFull; };
      if (form2 && !pass0_zeroes(attempts[0])) HIP_TRY(hipMemsetAsync(y.coef_scan, 0, scan_bytes, c->stream));
      dbg.mark("huffman_decode_dev: fills enqueued");
      int final_buf = 0;
      uint32_t* fl = c->h_flags;  // pinned; [9]: restart markers the unstuff pass dropped, [16] [17] / [0] [7]: their sequence sums as found / as due
      for (int q = 0; q < 24; q++) fl[q] = 0;
      bool hyp_done = false, unstuffed = false, rounds_ran = false;
      auto start_over = [&](const Attempt& next) -> uhdr_error_info_t {  // an attempt failed: everything it wrote goes back to its initial state
        HIP_TRY(hipMemsetAsync(y.flags, 0, 32, c->stream));  // not [8]: the stuffed-byte count stays
        HIP_TRY(hipMemsetAsync(y.flags + kHuffFlagStragglers, 0, 4, c->stream));  // (hyp_pass01_kernel appends to the list it finds)
        HIP_TRY(hipMemsetAsync(y.nblk, 0, (size_t)nsub * 4 + 4, c->stream));
        if (form2) {
          if (!pass0_zeroes(next)) HIP_TRY(hipMemsetAsync(y.coef_scan, 0, scan_bytes, c->stream));
        } else {
          HIP_TRY(hipMemsetAsync(y.dcd, 0, (size_t)total_blocks * 4, c->stream));
          for (int i = 0; i < a.ncomp; i++) HIP_TRY(hipMemsetAsync(a.coef[i], 0, zero_bytes[i], c->stream));
        }
        return ok_status();
      };
      for (size_t ti = 0; ti < attempts.size() && !hyp_done && !rounds_ran; ti++) {
        const Attempt& t = attempts[ti];
        if (ti > 0) UHDR_TRY(start_over(t));
        y.sub_bits = t.sub_bits;
        y.zero_ptr = pass0_zeroes(t) ? (uint4*)y.coef_scan : nullptr;
        y.zero_vec = pass0_zeroes(t) ? (uint32_t)(scan_bytes / 16) : 0u;
        const uint32_t nsub_t = huff_sync_max_subsequences(data_bytes, t.sub_bits);
        if (t.levels > 0) {
          y.hyp_h = bpm;
          y.hyp_levels = t.levels;
          y.hyp_main_levels = rst_sync ? 0 : main_levels_env;
          y.strag_levels = [&] { const char* e = getenv("UHDR_HIP_HUFF_STRAG_LEVELS"); return e ? atoi(e) : t.levels; }();
          y.hyp_state = (uint64_t*)(sb + o_hs);
          y.hyp_map = sb + o_hm;
          y.hyp_cnt = (uint16_t*)(sb + o_hc);
          y.hyp_hist = getenv("UHDR_HIP_HUFF_DEBUG") ? 1 : 0;
          // pieces of (at least) 256 bits; the buffers were sized for the smallest subsequences of the attempt list
          {
            // Measured (4K API-1 pair): the map's 1024-bit subsequences in four pieces take its write pass from 75 to 35 us; the base image's
            // 512-bit ones in two pieces gain nothing (40 -> 39 us: that launch already has 790 waves) and cost pass 1 four: pieces from 1024 bits on.
            int q = t.sub_bits >= 1024u ? (int)(t.sub_bits / 256u) : 1;
            if (getenv("UHDR_HIP_HUFF_PIECES_ALL")) q = (int)(t.sub_bits / 256u);
            if (q > max_pieces) q = max_pieces;
            if (q < 1) q = 1;
            while (q > 1 && (t.sub_bits % (uint32_t)q || (t.sub_bits / (uint32_t)q) % 32u)) q--;
            if (q == 3) q = 2;  // (the staged layout wants power-of-two pieces)
            if (getenv("UHDR_HIP_HUFF_QMERGE") && atoi(getenv("UHDR_HIP_HUFF_QMERGE")) == 0) q = 1;  // the round-3 form of pass 1 keeps no notes
            y.pieces = q;
            y.mid_state = q > 1 ? (uint64_t*)(sb + o_ms) : nullptr;
            y.mid_cnt = q > 1 ? (uint16_t*)(sb + o_mc) : nullptr;
            y.pend = q > 1 ? (uint64_t*)(sb + o_pe) : nullptr;
            y.pcnt = q > 1 ? (uint32_t*)(sb + o_pc) : nullptr;
            y.scan_tmp = q > 1 ? (uint32_t*)(sb + o_st2) : (uint32_t*)(sb + o_st);
          }
          // hyp_map <- 0xff (unmapped); state[0] <- 0xff: a start state the write pass skips, should the chain be lost
          if (!(fills_in_unstuff && ti == 0)) HIP_TRY(hipMemsetAsync(y.state[0], 0xff, ff_bytes, c->stream));
          // hyp_cnt needs no initialisation: a slot's count is written together with its map entry, and only mapped slots are read
          size_t tiles_off = 0;
          (void)huff_hyp_chain_bytes(data_bytes, t.sub_bits, &tiles_off);
          {
            ProfScope ps(c, "huffman_decode");
            if (!unstuffed)
              HIP_TRY(launch_huffman_unstuff(data, (uint32_t)data_bytes, (uint32_t*)(sb + o_cnt), y.flags + 8, sb + o_clean, c->stream, rst_map, y.rst_partial,
                                             fills_in_unstuff && ti == 0 ? &init_fill : nullptr));
            unstuffed = true;
            HIP_TRY(launch_huffman_decode_hyp(y, (int*)(sb + o_dcp), sb + o_ch, sb + o_ch + tiles_off, c->stream));
          }
          dbg.mark("huffman_decode_dev: hypothesis attempt enqueued");
          HIP_TRY(hipMemcpyAsync(fl, y.flags, 24 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(wait_stream(c));
          dbg.mark("huffman_decode_dev: hypothesis attempt finished");
          hyp_done = fl[2] == 0;
          if (hyp_done && ti > 0) {  // where the next scan like this one starts
            uhdr_hip_ctx::HuffHint& hint = c->huff_hint[bpm & 15];
            hint.sub_bits = t.sub_bits;
            hint.bits_per_block = (uint32_t)((uint64_t)data_bytes * 8u / ((uint64_t)a.total_mcus * (uint64_t)bpm));
          }
          if (getenv("UHDR_HIP_HUFF_DEBUG")) {
            uint32_t hist[32] = {};
            (void)hipMemcpy(hist, y.flags, sizeof hist, hipMemcpyDeviceToHost);
            fprintf(stderr, "uhdr_hip: hypothesis decode of %zu bytes, %u subsequences of %u bits x %d: merges per level %u %u %u %u %u %u+, %u paths unmerged after %d levels (%d in lockstep, %u paths handed to the straggler waves), true path %s\n",
                    data_bytes, nsub_t, t.sub_bits, bpm, hist[10], hist[11], hist[12], hist[13], hist[14], hist[15], fl[3], t.levels, y.hyp_main_levels, fl[kHuffFlagStragglers], hyp_done ? "resolved" : "LOST (next attempt)");
            fprintf(stderr, "uhdr_hip: the stragglers' merges at level 2 .. 10, 11+: %u %u %u %u %u %u %u %u %u %u\n", hist[22], hist[23], hist[24], hist[25], hist[26], hist[27], hist[28], hist[29],
                    hist[30], hist[31]);
          }
        } else {
          y.pieces = 1;
          y.scan_tmp = (uint32_t*)(sb + o_st);
          {
            ProfScope ps(c, "huffman_decode");
            if (!unstuffed) HIP_TRY(launch_huffman_unstuff(data, (uint32_t)data_bytes, (uint32_t*)(sb + o_cnt), y.flags + 8, sb + o_clean, c->stream, rst_map, y.rst_partial));
            unstuffed = true;
            HIP_TRY(launch_huffman_decode_sync(y, max_rounds, (int*)(sb + o_dcp), &final_buf, c->stream));
          }
          HIP_TRY(hipMemcpyAsync(fl, y.flags, 24 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(wait_stream(c));
          rounds_ran = true;
        }
      }
      const bool settled = hyp_done || fl[4 + max_rounds % 3] == 0;
      if (rst_sync && getenv("UHDR_HIP_HUFF_DEBUG"))
        fprintf(stderr, "uhdr_hip: restart file through the parallel decoder: %s, status bits %#x, %u markers dropped (%d expected), sequence sums %s\n",
                settled ? "settled" : "NOT settled", fl[1], fl[9], a.nseg - 1, fl[0] == fl[16] && fl[7] == fl[17] ? "equal" : "DIFFERENT");
      if (settled && rst_sync && ((fl[1] & 14u) != 0 || fl[9] != (uint32_t)(a.nseg - 1) || fl[0] != fl[16] || fl[7] != fl[17])) {
        // a restart file that is not what its headers say (markers missing, misplaced or out of step, damaged data): the
        // interval decoder below looks at every marker and words the error
        for (int i = 0; i < a.ncomp; i++) HIP_TRY(hipMemsetAsync(a.coef[i], 0, zero_bytes[i], c->stream));
        coef_zeroed = true;
      } else if (settled) {  // the fixed point was reached: the decode is the true one
        if (fl[1] & 8u) return err_status(UHDR_CODEC_INVALID_PARAM, "corrupt entropy-coded data (the scan ends before its last block)");
        if (fl[1] & 2u) return err_status(UHDR_CODEC_INVALID_PARAM, "corrupt entropy-coded data (undefined Huffman code or a run past the end of a block)");
        sync_done = true;
      } else {  // not settled: start over on the serial path
        if (a.nseg == 1 && !c->huff_serial_ok && data_bytes > (256u << 10)) {
          c->stats.entropy_decode_declined++;
          return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the parallel entropy decode did not settle in %d rounds; %zu bytes on one lane would take longer than the CPU", max_rounds, data_bytes);
        }
        for (int i = 0; i < a.ncomp; i++) HIP_TRY(hipMemsetAsync(a.coef[i], 0, zero_bytes[i], c->stream));
        coef_zeroed = true;
      }
    }
  }
  if (sync_done) {
    c->stats.entropy_decode_parallel++;
    return ok_status();
  }
  // one interval on one lane: fine for a thumbnail, slower than any CPU for a frame.  A caller that has a CPU decoder to
  // fall back on (uhdr_hip_jpeg_decode_scan behind the facade) gets the stream back instead -- this is also where a file
  // whose Huffman tables do not fit the two-level form (more than kHuffL2Max long-code prefixes) ends up
  if (a.nseg == 1 && !c->huff_serial_ok && data_bytes > (256u << 10)) {
    c->stats.entropy_decode_declined++;
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "a %zu-byte scan without restart markers that the parallel decoder does not take (Huffman tables outside its two-level form)", data_bytes);
  }
  if (!coef_zeroed)
    for (int i = 0; i < a.ncomp; i++) HIP_TRY(hipMemsetAsync(a.coef[i], 0, zero_bytes[i], c->stream));
  HIP_TRY(hipMemsetAsync(a.status, 0, 16, c->stream));  // (only this route reads it: the fill used to be the first launch of every decode, 5 us in front of the parallel route too)
  {
    ProfScope ps(c, "huffman_decode");
    HIP_TRY(launch_huffman_decode(a, counts, starts, ends, c->stream));
  }
  if (a.nseg == 1) c->stats.entropy_decode_single_lane++;
  else c->stats.entropy_decode_intervals++;
  uint32_t st[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(st, a.status, sizeof st, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(wait_stream(c));
  if (st[1] != (uint32_t)(a.nseg - 1))
    return err_status(UHDR_CODEC_INVALID_PARAM, "found %u restart markers, a restart interval of %d MCUs over %d MCUs needs %d", st[1], a.ri,
                      a.total_mcus, a.nseg - 1);
  if (st[0] & 4u) return err_status(UHDR_CODEC_INVALID_PARAM, "restart markers out of sequence");
  if (st[0] & 2u) return err_status(UHDR_CODEC_INVALID_PARAM, "corrupt entropy-coded data (undefined Huffman code or a run past the end of a block)");
  return ok_status();
}

// ---- the two scans of one UltraHDR file at once (round 5) ------------------------------------------------------------------
// JpegR::encodeJPEGR codes the gain map, then the base image (jpegr.cpp:253-316); JpegR::decodeJPEGR decodes the base image, then
// the map (jpegr.cpp:1478-1488) -- one after the other because libjpeg is serial anyway.  On the device the two scans are independent
// work whose passes are latency- or occupancy-bound for long stretches (single-workgroup scans, the write pass at under one wave per
// SIMD, the stragglers' scalar chains): the second scan runs on the context's auxiliary context (own stream, scratch and table
// cache) from a second host thread, the first on the caller's thread; both entry points are synchronous like their single-scan forms.
uhdr_error_info_t uhdr_api::aux_context(uhdr_hip_ctx* c, uhdr_hip_ctx** out) {
  if (!c->aux) {
    uhdr_error_info_t e = ok_status();
    c->aux = uhdr_hip_create(c->device, &e);
    if (!c->aux) return e.error_code != UHDR_CODEC_OK ? e : err_status(UHDR_CODEC_ERROR, "could not create the auxiliary context");
  }
  c->aux->prof = c->prof;
  c->aux->huff_serial_ok = c->huff_serial_ok;
  *out = c->aux;
  return ok_status();
}
void uhdr_api::aux_merge(uhdr_hip_ctx* c) {  // what the auxiliary context counted and timed belongs to the caller's context
  if (!c->aux) return;
  uhdr_hip_ctx* x = c->aux;
  c->stats.entropy_decode_parallel += x->stats.entropy_decode_parallel;
  c->stats.entropy_decode_intervals += x->stats.entropy_decode_intervals;
  c->stats.entropy_decode_single_lane += x->stats.entropy_decode_single_lane;
  c->stats.entropy_decode_declined += x->stats.entropy_decode_declined;
  c->stats.entropy_encode_stream += x->stats.entropy_encode_stream;
  c->stats.entropy_encode_intervals += x->stats.entropy_encode_intervals;
  x->stats = uhdr_hip_stats_t();
  for (auto& e : x->prof_entries) c->prof_entries.push_back(e);
  x->prof_entries.clear();
}

// The auxiliary context's thread (round 6).  One job at a time: run(job) hands it over and returns, wait() blocks until it is done.
struct AuxWorker {
  // Both hand-overs spin for a moment before they sleep: a condition variable's wake-up is 10-30 us of scheduler latency, paid once when the job
  // is handed over and once when the caller waits for it -- both on the critical path of a 900 us round trip.  The worker keeps polling for
  // kSpinUs after a job (the next call of a busy service follows at once), the caller polls while the job runs its last microseconds.
  static constexpr int kSpinUs = 300;
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<void()> job;
  std::atomic<int> state{0};  // 0 idle, 1 job posted, 2 quit
  std::atomic<bool> busy{false};
  static bool spin_until(const std::function<bool()>& pred, int us) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!pred()) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(us)) return pred();
#if defined(__x86_64__)
      __builtin_ia32_pause();
#endif
    }
    return true;
  }
  AuxWorker() {
    th = std::thread([this] {
      for (;;) {
        if (!spin_until([this] { return state.load(std::memory_order_acquire) != 0; }, kSpinUs)) {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [this] { return state.load(std::memory_order_acquire) != 0; });
        }
        if (state.load(std::memory_order_acquire) == 2) return;
        std::function<void()> j;
        {
          std::lock_guard<std::mutex> lk(mu);
          j = std::move(job);
          state.store(0, std::memory_order_release);
        }
        j();
        {
          std::lock_guard<std::mutex> lk(mu);
          busy.store(false, std::memory_order_release);
        }
        cv.notify_all();
      }
    });
  }
  void run(std::function<void()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      busy.store(true, std::memory_order_release);
      state.store(1, std::memory_order_release);
    }
    cv.notify_all();
  }
  void wait() {
    if (spin_until([this] { return !busy.load(std::memory_order_acquire); }, kSpinUs)) return;
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [this] { return !busy.load(std::memory_order_acquire); });
  }
  ~AuxWorker() {
    {
      std::lock_guard<std::mutex> lk(mu);
      state.store(2, std::memory_order_release);
    }
    cv.notify_all();
    if (th.joinable()) th.join();
  }
};
void aux_worker_destroy(uhdr_hip_ctx* c) {
  if (c && c->aux_worker) {
    delete c->aux_worker;
    c->aux_worker = nullptr;
  }
}
bool uhdr_api::aux_post(uhdr_hip_ctx* c, std::function<void()> job) {
  static const bool no_thread = getenv("UHDR_HIP_NO_AUX_THREAD") != nullptr;
  if (no_thread) return false;
  if (!c->aux_worker) {
    try {
      c->aux_worker = new AuxWorker();
    } catch (...) {
      c->aux_worker = nullptr;
    }
  }
  if (!c->aux_worker) return false;
  c->aux_worker->run(std::move(job));
  return true;
}
void uhdr_api::aux_wait(uhdr_hip_ctx* c) {
  if (c->aux_worker) c->aux_worker->wait();
}
namespace {
// job_b on the context's worker thread while job_a runs on the caller's; without a thread to be had: one after the other (still on two streams)
template <typename FA, typename FB>
void run_pair(uhdr_hip_ctx* c, FA&& job_a, FB&& job_b) {
  static const bool no_thread = getenv("UHDR_HIP_NO_AUX_THREAD") != nullptr;
  if (!c->aux_worker && !no_thread) {
    try {
      c->aux_worker = new AuxWorker();
    } catch (...) {
      c->aux_worker = nullptr;
    }
  }
  if (c->aux_worker) {
    c->aux_worker->run(job_b);
    job_a();
    c->aux_worker->wait();
  } else {
    job_a();
    job_b();
  }
}
}  // namespace

uhdr_error_info_t uhdr_hip_huffman_encode2_dev(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_scan_t* scan_a, uint8_t* out_a, size_t cap_a, size_t* bytes_a,
                                               const uhdr_hip_jpeg_scan_t* scan_b, uint8_t* out_b, size_t cap_b, size_t* bytes_b) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipSetDevice(c->device));
  uhdr_hip_ctx* x = nullptr;
  UHDR_TRY(aux_context(c, &x));
  // the coefficients both scans read were produced on this stream: the auxiliary stream waits for them by event (round 6: was a host
  // synchronisation, i.e. the device idled while the entropy stage's first launches were being prepared)
  if (!c->aux_ev) HIP_TRY(hipEventCreateWithFlags(&c->aux_ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->aux_ev, c->stream));
  HIP_TRY(hipStreamWaitEvent(x->stream, c->aux_ev, 0));
  uhdr_error_info_t ra = ok_status(), rb = ok_status();
  const int dev = c->device;
  run_pair(c, [&] { ra = uhdr_hip_huffman_encode_dev(c, scan_a, out_a, cap_a, bytes_a); },
           [&, dev] { (void)hipSetDevice(dev); rb = uhdr_hip_huffman_encode_dev(x, scan_b, out_b, cap_b, bytes_b); });
  aux_merge(c);
  return ra.error_code != UHDR_CODEC_OK ? ra : rb;
}

uhdr_error_info_t uhdr_hip_huffman_decode2_dev(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_scan_t* scan_a, const uhdr_hip_huff_tables_t* tables_a, const uint8_t* data_a,
                                               size_t bytes_a, const uhdr_hip_jpeg_scan_t* scan_b, const uhdr_hip_huff_tables_t* tables_b,
                                               const uint8_t* data_b, size_t bytes_b) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipSetDevice(c->device));
  uhdr_hip_ctx* x = nullptr;
  UHDR_TRY(aux_context(c, &x));
  // the bytes of both scans may have come up on this stream: ordered by event, as above
  if (!c->aux_ev) HIP_TRY(hipEventCreateWithFlags(&c->aux_ev, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(c->aux_ev, c->stream));
  HIP_TRY(hipStreamWaitEvent(x->stream, c->aux_ev, 0));
  uhdr_error_info_t ra = ok_status(), rb = ok_status();
  const int dev = c->device;
  run_pair(c, [&] { ra = uhdr_hip_huffman_decode_dev(c, scan_a, tables_a, data_a, bytes_a); },
           [&, dev] { (void)hipSetDevice(dev); rb = uhdr_hip_huffman_decode_dev(x, scan_b, tables_b, data_b, bytes_b); });
  aux_merge(c);
  return ra.error_code != UHDR_CODEC_OK ? ra : rb;
}
