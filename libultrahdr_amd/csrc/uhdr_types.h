// Internal types shared between the host layer (api_*.cpp, host_tables.cpp) and the
// kernel translation units.  Nothing here is part of the C ABI (include/uhdr_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.h"
#include "uhdr_hip.h"

namespace uhdr {

// Device-side view of a uhdr_raw_image_t: plane pointers + strides in PIXELS (ultrahdr_api.h:245).
struct ImageView {
  const void* p[3];
  uint32_t stride[3];
  uint32_t w, h;
  int fmt;
  int range;
};
struct ImageViewMut {
  void* p[3];
  uint32_t stride[3];
  uint32_t w, h;
  int fmt;
  int range;
};

// ---- applyGainMap ------------------------------------------------------------------------------
// Per-call table block uploaded by the host (one hipMemcpyAsync):
//   srgb[1024]      sRGB inverse-OETF LUT                     (gainmapmath.cpp:126-131)
//   gain[3][1024]   GainLUT, one table per channel            (gainmapmath.h:452-470)
//   u8f[256]        b / 255.0f for every byte b               (mapUintToFloat, gainmapmath.cpp:864)
//   fac[3][256]     scale-1 shortcut: the gain FACTOR for map byte b, i.e. GainLUT applied to
//                   u8f[b] (with the 1/gamma pow done on the host) -- at scale 1 the sampler
//                   returns e1*1 + e2*0 + e3*0 + e4*0 == e1 exactly, so byte -> factor is a table
//   idw[4][s*s*4]   ShepardsIDW default / NR / NB / C tables  (gainmapmath.cpp:43-80)
constexpr int kSrgbN = 1024;
constexpr int kGainN = 1024;
constexpr int kOetfN = 65536;
constexpr int kOetfThrN = 1028;  // thresholds T[0..1023] + 3 (+1 pad) sentinels for the 3-compare window
constexpr int kOetfEstN = 4066;  // packed (code at bucket start | code at bucket end << 16), buckets k = bits(v) >> 18
constexpr int kOetfTabFloats = kOetfThrN + kOetfEstN;
constexpr int kInvOetfN = 4096;
// quad kernel: the HLG / PQ output-code step function as a bucket table in LDS (host_tables.cpp: make_bucket_table)
constexpr int kOetfBucketShiftHlg = 15, kOetfBucketShiftPq = 16;
constexpr int kOetfBucketsHlg = 5376, kOetfBucketsPq = 2304;  // LDS capacity in entries (42 / 18 KiB); the host checks n <= capacity
// encode kernels: step tables (sRGB byte of the tone mapper, encodeGain's byte) in the same {thr, lo | hi << 16} form
constexpr int kStepTabMax = 2048;  // entries (16 KiB of LDS each)
struct StepTab {
  const uint2* tab;  // null: not available for this call (the kernels then run the float64 evaluation)
  uint32_t n, base8, shm3;  // entries, first bucket * 8, shift - 3
  uint32_t lo_bits, hi_bits;  // the device's clamp; lo_bits >= base8 << shm3, so bucket * 8 - base8 cannot underflow
};
constexpr int kMaxIdwScaleLds = 8;  // idw tables up to 4*8*8*4 floats = 4 KiB live in LDS

struct ApplyTables {  // layout of the device table block, in floats
  static constexpr int kSrgbOff = 0;
  static constexpr int kGainOff = kSrgbOff + kSrgbN;
  static constexpr int kU8fOff = kGainOff + 3 * kGainN;
  static constexpr int kFacOff = kU8fOff + 256;
  // p3YuvToRgb's chroma products per chroma byte (gainmapmath.cpp:177-181; applyGainMap always converts with the P3 / BT.601
  // coefficients, jpegr.cpp:1723): [v] -> {cr * vf, gcr * vf}, [u] -> {gcb * uf, cb * uf}, vf = (v - 128) * (1 / 255.0f)
  static constexpr int kChromaVOff = kFacOff + 3 * 256;
  static constexpr int kChromaUOff = kChromaVOff + 2 * 256;
  // LDS images for the quad kernels, laid out exactly as the kernels keep them so that the per-workgroup staging is a
  // handful of 16-byte copies: the sRGB LUT padded to 2048 entries with its last value (apply_gainmap.hip: kSrgbPad), the
  // byte -> float table with every entry doubled ({x, x}: the packed arithmetic reads register pairs), and -- behind the
  // reference-order IDW tables -- the same weights re-laid out for pixel pairs
  static constexpr int kSrgbPadOff = kChromaUOff + 2 * 256;
  static constexpr int kTapOff = kSrgbPadOff + 2048;
  static constexpr int kIdwOff = kTapOff + 512;
  static constexpr int idw_floats(int scale) { return 4 * scale * scale * 4; }
  static constexpr int idw_pair_off(int scale) { return kIdwOff + idw_floats(scale); }  // even scales only (0 floats otherwise)
  static constexpr int floats(int scale) { return kIdwOff + idw_floats(scale) + (scale % 2 == 0 ? idw_floats(scale) : 0); }
};

// batch mode: per-frame plane pointers; geometry / strides / metadata are shared
constexpr int kMaxBatchFrames = 16;  // frames per launch (api_gainmap.cpp: beyond ~16 separate allocations DRAM locality drops)
struct FramePtrs {
  const uint8_t *y, *u, *v, *map;
  uint8_t* dst;
};

// applyGainMap straight from JPEG coefficients: the 4:2:0 base image as jpeg_read_coefficients() yields it
// (device-resident descriptor; JBLOCK arrays in raster block order, tables in natural order)
struct CoefSrc {
  const int16_t* coef[3];  // Y, Cb, Cr
  int bw[3], bh[3];        // comp_info[c].width_in_blocks / height_in_blocks
  int q[3][64];            // comp_info[c].quant_table->quantval
};

struct ApplyParams {
  ImageView sdr;      // base image (this rank's stripe)
  ImageView gm;       // whole gain map
  ImageViewMut dst;   // destination stripe
  const float* tables;      // ApplyTables block
  const float* oetf_thr;    // generic kernel. HLG: output-code threshold block (kOetfTabFloats); PQ: 65536 uint16 output codes of pqOetfLUT's nodes; linear: null
  const uint2* oetf_buckets;   // quad kernel, HLG / PQ: bucket table {thr, lo | hi << 16} (null: not verified exact -> generic kernel)
  uint32_t oetf_n, oetf_base8; // entries, first bucket * 8
  uint32_t oetf_lo_bits;       // lower clamp (bit pattern): max(domain start, first bucket's start)
  uint32_t oetf_hi_bits;       // upper end of the table's domain (bit pattern)
  int oetf_prescaled;          // the table takes the value before the nit scaling (x * 203) / peak (no HDR-side gamut conversion)
  uint32_t y0;              // global row of stripe row 0
  uint32_t tiles_per_wave;  // quad kernel: loop trip count (even), set by the launcher
  uint32_t touch_ahead;     // quad kernel: 1 = the launch opens with a grid-wide read sweep over the input planes (set by the launcher)
  uint32_t prefetch_wgs;    // quad kernel: the first prefetch_wgs workgroups only stream the input planes into L2 / the infinity cache (set by the launcher)
  uint32_t inputs_hot;      // the host layer's guess that the input planes still sit in the infinity cache (api_gainmap.cpp: mall_model): no prefetchers then
  uint32_t row_groups;      // quad kernel: quad-row step of a wave, set by the launcher
  uint32_t n_frames;        // 0/1: single image; > 1: batch through `frame_tab` (quad kernel only)
  uint32_t scale;           // integer map scale factor (table path) or 0
  uint32_t scale_magic;     // ceil(2^32 / scale): x / scale == umulhi(x, magic) for x < 65536
  float scale_f;            // (float)w_sdr / w_map, for the non-integer path
  int map_bpp;              // 1, 3 or 4
  int map_ch;               // 1 or 3
  int out_ct;               // uhdr_color_transfer_t
  int sdr_is_rgb;           // base image read as RGB (RGBA8888) -- no YUV->RGB
  int sdr_gamut_on, hdr_gamut_on;
  int gamma_is_one[3];      // per-channel 1/gamma == 1
  float gamma_inv[3];
  float offset_sdr[3], offset_hdr[3];
  Mat3 gamut;               // hdr_cg <- sdr_cg
  Yuv2Rgb yuv;              // always the BT.601 set (jpegr.cpp:1723)
  const CoefSrc* coef_src;  // quad kernel, SRC 1 only: the base image in coefficient form (p.sdr then carries geometry only)
  FramePtrs frame_tab[kMaxBatchFrames];  // batch: plane pointers of the launch's frames, read with scalar loads from the kernel arguments
};

// ---- generateGainMap -----------------------------------------------------------------------------
struct GenParams {
  ImageView sdr, hdr;
  uint32_t map_w, map_h, scale;
  const float* srgb_lut;     // 1024
  const float* hdr_inv_lut;  // 4096 (HLG with the OOTF folded in / PQ) or 1024 (sRGB) or null (linear)
  int hdr_inv_n;
  const double* math_tab;    // exact_math.h tables (kMathTabDoublesAll)
  const float* lin10;        // RGBA1010102 HDR at scale 1: 10-bit code -> linear value, 1024 floats; else null
  const float* srgb_of_byte; // 256: sRGB inverse OETF of byte / 255.0f (the fused front end reads its own SDR bytes back)
  float gain_cap;            // two pass: the ratio-domain form of computeGain's 2.3f cap (host_tables.cpp: gain_cap_ratio)
  StepTab gain8;             // one pass, gamma 1: clamped gain -> map byte
  int sdr_is_rgb, hdr_is_rgb;
  int sdr_gamut_on, hdr_gamut_on;
  Mat3 sdr_gamut, hdr_gamut;
  Yuv2Rgb sdr_yuv, hdr_yuv;
  float lum[3];              // luminance coefficients of the SDR gamut (used for both images)
  int multichannel, use_luminance;
  float hdr_nits;            // hdrSampleToNitsFactor
  // one pass
  float min_boost, max_boost, log2min, log2max, gamma;
  double log2_range, log2_range_rcp;  // (double)(log2max - log2min) and its reciprocal
  uint8_t* out;              // map bytes
  uint32_t out_stride;       // pixels
  // two pass
  float* gain_log2;          // map_w*map_h*(3|1): the float gain RATIOS (round 4; log2 of the six extrema only, see encode_core.h)
  float* minmax;             // 6 floats: the reference's log2 extrema
};

// Two-pass generation: the final per-channel range and everything derived from it live in device memory, so the
// sequence pass 1 -> [all-reduce] -> finalize -> pass 2 needs no host round trip.
// Round 4: pass 1 stores the gain RATIO q = (hdr + eps) / (sdr + eps) (encode_core.h: gain_ratio); pass 2's byte
//   q -> (float)log2(q) -> (g - min) / (max - min) [-> pow gamma] -> * 255 + 0.5 -> clip -> truncate   (jpegr.cpp:900-1013)
// is a monotone step function of q with at most 255 steps, tabulated per channel ON THE DEVICE once the range is known
// (minmax_table_kernel): buckets of 2^shift consecutive bit patterns with at most one threshold each, found by
// bisection through the exact float64 evaluation -- the same form as the host-built step tables (StepTab).
constexpr int kAffTabMax = 1024;  // entries per channel (8 KiB of LDS each)
struct AffineTabDev {
  uint32_t n, base8, shm3;     // entries, first bucket * 8, shift - 3
  uint32_t lo_bits, hi_bits;   // the table's domain (bucket aligned); the step function is constant beyond it
  uint32_t ok;                 // 0: no table for this range (too dense, not monotone) -> pass 2 evaluates per sample
  uint32_t pad[2];
};
struct AffineDev {
  float mn[3], mx[3];
  double range_rcp[3];
  AffineTabDev tab[3];
};
constexpr size_t kAffineTablesOff = 256;  // byte offset of the 3 x kAffTabMax table entries behind an AffineDev
constexpr size_t kAffineDevBytes = kAffineTablesOff + 3 * (size_t)kAffTabMax * 8;
static_assert(sizeof(AffineDev) <= kAffineTablesOff, "AffineDev outgrew its slot");
struct AffineParams {
  const float* gain_log2;      // the float gain ratios of pass 1
  uint8_t* out;
  uint32_t map_w, map_h, out_stride, nch;
  float gamma;
  const AffineDev* dev;        // range + step tables (written by minmax_table_kernel); the tables follow at kAffineTablesOff
  const double* math_tab;      // exact_math.h tables in global memory (per-sample evaluation when a table is not ok)
};
// One launch for everything between the passes (generate_gainmap.hip: minmax_table_kernel), one workgroup per channel:
//   reduce    partials of pass 1 (ratio extrema) -> log2 extrema mm6 {min0..2, max0..2} (+ the {min, -max} form for the
//             single min-all-reduce of the striped path)
//   finalize  jpegr.cpp:969-986: clamp, user hints, epsilon guard -> AffineDev, the final {min, max} for the metadata
//   table     the per-channel step table of pass 2
struct MinmaxTableParams {
  int do_reduce, do_finalize, do_table;
  // reduce
  const float* partials;  // n_partials x 6 ratio extrema
  int n_partials;
  int empty;              // this rank's stripe holds no sample: contribute the identity of the merge (127 / -128)
  float* mm6;             // out: log2 extrema
  float* merged6;         // out (may be null): {min0..2, -max0..2}
  // finalize
  const float* merged_in; // {min0..2, -max0..2} (after the all-reduce); null: this launch's own mm6
  int nch;
  int has_max_hint, has_min_hint;
  float log2_max_hint, log2_min_hint;  // log2f of the user's content-boost recommendations (host-computed)
  float* out_mm;          // out (may be null): 9 floats -- the final {min0..2, max0..2} for the host's metadata fill, then 1 / 0 per channel: it got a step table
  // table (and finalize's output)
  float final_mm[6];      // do_finalize == 0: the already-final range (uhdr_hip_generate_gainmap_pass2_dev)
  AffineDev* dev;
  float gamma;
  const double* math_tab;
};

// ---- toneMap -------------------------------------------------------------------------------------
struct ToneMapParams {
  ImageView hdr;
  ImageViewMut sdr;
  const float* hdr_inv_lut;  // as in GenParams
  int hdr_inv_n;
  const double* math_tab;    // exact_math.h tables (kMathTabDoublesAll)
  const float* lin10;        // RGBA1010102 input: 10-bit code -> linear value (unpack + inverse OETF [+ OOTF]), 1024 floats; else null
  StepTab srgb8;             // RGBA8888 output: clamped linear value -> sRGB byte
  int hdr_is_rgb, is_normalized;
  float headroom, headroom_sq, headroom_sq_rcp;  // hdr_white / 203, its square (float product) and 1 / square
  int gamut_on;
  Mat3 gamut;      // P3 <- hdr gamut
  Yuv2Rgb hdr_yuv;
  Rgb2Yuv p3;
};

// ---- fused API-0 front end (encode_fused.hip) --------------------------------------------------------
struct FusedParams {
  ToneMapParams tm;  // tm.hdr = input; tm.sdr.p[0] = optional RGBA8888 output (null: not stored)
  GenParams gen;     // gain-map side; gen.sdr / gen.hdr views are unused (the pixels arrive in registers)
  Rgb2Yuv base_k;    // Display-P3 (BT.601) coefficients of the base image conversion
  ImageViewMut ycc;  // base image, YCbCr 4:4:4 planes
};

// ---- convertYuv / convert_raw_input_to_ycbcr --------------------------------------------------------
struct YuvXformParams {
  ImageViewMut img;
  Mat3 c;
};
struct RgbToYcbcrParams {
  ImageView src;
  ImageViewMut dst;
  Rgb2Yuv k;
};

// launchers (defined in the .hip files)
hipError_t launch_apply_gainmap(const ApplyParams& p, hipStream_t s);
hipError_t launch_apply_gainmap_coef(const ApplyParams& p, hipStream_t s);
int apply_quad_mode(const ApplyParams& p);  // >= 0: the quad kernel (and batch mode) applies
hipError_t launch_generate_gainmap(const GenParams& p, bool two_pass, hipStream_t s);
hipError_t launch_affine_map(const AffineParams& p, hipStream_t s);
hipError_t launch_minmax_table(const MinmaxTableParams& p, hipStream_t s);
// encode_api1_fused.hip
hipError_t launch_map_blocks(const float* ratio, const AffineDev* dev, const double* math_tab, int nch, int bw, int bh, const uint16_t* qt_luma,
                             const uint16_t* qt_chroma, int16_t* const coef[3], uint8_t* map_out, uint32_t out_stride, hipStream_t s);
hipError_t launch_base_blocks(const ImageView& yuv420, const Mat3* c, const uint16_t* qt_luma, const uint16_t* qt_chroma, int16_t* const coef[3],
                              hipStream_t s);
hipError_t launch_selftest(int which, unsigned long long* out, uint32_t arg0, uint32_t arg1, uint32_t seed, const double* math_tab, const AffineDev* dev,
                           hipStream_t s);  // selftest.hip
int gen_partials_count(const GenParams& p);  // workgroups (= partials) launch_generate_gainmap(p, two_pass = true) writes
hipError_t launch_encode_api0_fused(const FusedParams& p, bool two_pass, int* grid_out, hipStream_t s);
hipError_t launch_tone_map(const ToneMapParams& p, hipStream_t s);
hipError_t launch_transform_yuv(const YuvXformParams& p, hipStream_t s);
hipError_t launch_rgb_to_ycbcr(const RgbToYcbcrParams& p, hipStream_t s);
// Samples of partial edge blocks, made up on the device the way JpegEncoderHelper::compressYCbCr does on the host
// (jpegencoderhelper.cpp:246-309): w x h valid samples; col_mode 0 = the plane's pitch covers the block-aligned width and the
// bytes beyond w are read as they are, rows beyond h are constant `fill`; col_mode 1 = columns beyond w are `fill`, rows beyond
// h repeat row y - mcu_rows (the helper's stale scratch row; zeros before the first).  fill: 0 for component 0, 128 else.
struct FdctEdge {
  int on, w, h, col_mode, fill, mcu_rows;
};
hipError_t launch_fdct_quant(const uint8_t* plane, size_t stride, int bw, int bh,
                             const uint16_t* qt_host, int16_t* coef, hipStream_t s, const FdctEdge* edge = nullptr);

// vw x vh > 0: valid pixels; blocks reaching beyond them replicate the last column / row (libjpeg's own padding of scanline input)
hipError_t launch_fdct_quant_rgb(const uint8_t* rgb, size_t pitch, int bpp, int bw, int bh, const uint16_t* qt_luma_host,
                                 const uint16_t* qt_chroma_host, int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr, hipStream_t s, int vw = 0,
                                 int vh = 0);
hipError_t launch_repack(int mode, const void* src, size_t src_pitch, void* dst, size_t dst_pitch, uint32_t w, uint32_t h,
                         hipStream_t s);  // 0: RGB888 -> RGBA8888, 1: RGBA8888 -> Y400
// ---- image effects (effects.hip) -----------------------------------------------------------------------
struct EffectPlane {
  const void* src;
  void* dst;
  uint32_t elem;                       // bytes per element: 1, 2, 4, 8
  uint32_t src_w, src_h, src_stride;   // in elements
  uint32_t dst_w, dst_h, dst_stride;
  uint32_t mode;                       // 0 / 1 / 2 rotate 90 / 180 / 270, 3 mirror vertical, 4 mirror horizontal, 5 crop, 6 resize
  uint32_t a0, a1;                     // crop: left, top; resize: src_w / dst_w, src_h / dst_h
};
hipError_t launch_effect_plane(const EffectPlane& p, hipStream_t s);

// ---- baseline Huffman entropy coding (huffman_encode.hip) -------------------------------------------
struct HuffArgs {
  const int16_t* coef[3];  // JBLOCK arrays of bw x bh REAL blocks
  int bw[3], bh[3];
  int hs[3], vs[3];        // sampling factors (ncomp > 1)
  int ncomp;
  int mcus_per_row, total_mcus;
  int ri;                  // MCUs per restart interval
  int blocks_per_mcu;
  int nseg;                // restart intervals
  const uint32_t* tables;  // host::jpeg_huff_code_tables()
  const uint8_t* zigzag;   // zig-zag position -> natural index
  uint8_t* slots;          // nseg x slot_stride bytes of scratch
  uint32_t slot_stride;
  uint32_t* seg_bytes;     // [nseg] stuffed bytes per interval (0xFFFFFFFF: coefficients outside the baseline range)
};
uint32_t huff_slot_stride();
// marker-less scans (restart_interval 0): the stream the reference writes
struct HuffStream {
  uint32_t* seg_bits;   // [nseg] bits per wavefront segment (0xFFFFFFFF: coefficients outside the baseline range)
  uint64_t* seg_start;  // [nseg + 1] first bit of every segment; [nseg] = total bits
  uint32_t* raw;        // the unstuffed stream, raw_words words
  uint64_t raw_words;
  uint32_t* meta;       // [0..1] total bits, [2] status
};
int huff_stream_segment_mcus(int blocks_per_mcu);
int huff_stuff_chunks(uint64_t raw_bytes);
hipError_t launch_huffman_encode_stream(const HuffArgs& a, const HuffStream& t, uint32_t* chunk_counts, uint64_t* out_bytes, uint8_t* out, uint64_t cap,
                                        hipStream_t s);
hipError_t launch_huffman_encode(const HuffArgs& a, uint64_t* offsets, uint32_t* status, uint8_t* out, uint64_t cap, hipStream_t s);

// ---- baseline Huffman decoding (huffman_decode.hip) ---------------------------------------------------
struct HuffDecTable {    // one DHT table, decode form (jdhuff.c jpeg_make_d_derived_tbl)
  uint16_t lut[512];     // 9-bit look-ahead: code length << 8 | symbol, 0 = longer than 9 bits
  int maxcode[18];       // largest code of length l (-1: none)
  int valoff[17];        // vals index of the first symbol of length l minus its code
  uint8_t vals[256];
};
struct HuffDecArgs {
  const uint8_t* data;   // entropy-coded bytes (device)
  uint32_t nbytes;
  const uint32_t* starts;
  const uint32_t* ends;  // interval k = [starts[k], ends[k])
  int nseg, ri, total_mcus, mcus_per_row, ncomp;
  int bw[3], bh[3], hs[3], vs[3];
  int16_t* coef[3];      // zero-initialised JBLOCK arrays
  const HuffDecTable* tabs;  // DC luma, AC luma, DC chroma, AC chroma
  const uint8_t* zigzag;
  uint32_t* status;      // [0]: 2 = bad code / block overrun, 4 = markers out of sequence; [1]: markers found
};
// the self-synchronising decoder for scans without restart markers (huffman_decode_sync.hip)
constexpr int kHuffL2Max = 16;  // second-level sub-tables (one per 9-bit prefix that has longer codes; Annex K tables need 2-8)
struct HuffFastTable {          // two-level decode form of one DHT table: 9 bits, then 7 more
  uint16_t l1[512];             // length << 8 | symbol;  0x8000 | sub-table for a longer code;  0 = undefined
  uint16_t l2[kHuffL2Max][128]; // length << 8 | symbol;  0 = undefined
};
constexpr int kHuffValWords = 512 + kHuffL2Max * 128;  // one table of the write pass: first level, then the sub-tables
// Round 6: the state-tracking passes take up to TWO symbols per step.  Pair form of one table (host: make_pair_table), indexed by the
// next kHuffPairBits bits of the stream: bits 0-4 bits consumed by the first symbol (code + magnitude), 5-11 its zig-zag advance,
// 12-16 / 17-23 the same for the AC symbol that follows it IF its code also lies inside the index (0: no second symbol);
// bit 31: the first code is longer than the index, bits 0-3 = its second-level sub-table (tracking form, as before).
constexpr int kHuffPairBits = 10, kHuffPairWords = 1 << kHuffPairBits;
// the blobs the kernels stage with ONE copy: ptabs = 4 pair tables, then the four tracking tables' second levels (uint16); pvtabs = 4 x
// kHuffPairWords x {first, second} value words, then the four value tables' second levels as uint16 (bits | advance << 5 | magnitude bits
// << 12; 0 = malformed)
// The sub-tables of all four tables are PACKED behind the pair words (round 6): kHuffL2Total of them in all -- the Annex K tables use 0 + 5 + 1 + 5 -- and a
// long-code entry carries the index into the packed array.  (Four times kHuffL2Max sub-tables, 16 KB of which 13 held nothing, were LDS that kept a
// workgroup of the other scan's kernels off the CU.)
constexpr int kHuffL2Total = 32;
constexpr int kHuffPairBlobWords = 4 * kHuffPairWords + kHuffL2Total * 128 / 2;
constexpr int kHuffPairValBlobWords = 8 * kHuffPairWords + kHuffL2Total * 128 / 2;
struct HuffSyncArgs {
  const uint8_t* clean;       // unstuffed entropy-coded bytes (device)
  uint32_t nbytes;            // size of the STUFFED stream (upper bound of the clean size)
  const uint32_t* nstuffed;   // device word: bytes dropped by the unstuff pass
  uint32_t sub_bits;          // subsequence size in bits
  uint64_t* state[2];         // end state per subsequence, double buffered: bit position | block in MCU << 32 | zig-zag index << 40
  uint8_t* changed[2];
  uint32_t* nblk;             // blocks completed per subsequence; later their exclusive scan
  uint32_t* scan_tmp;         // scratch of that scan: one word per 2048 subsequences
  uint32_t* flags;            // [1]: status bits (2 bad code / run, 4 restart marker out of place, 8 truncated); [9]: restart markers found, [16] [17]: their sequence sums, [0] [7]: the sums the write pass expects; [4..6]: change counters of the rounds (r % 3); [8]: stuffed bytes
  int* dcd;                   // DC differences of all blocks in scan order
  uint32_t total_blocks;
  int blocks_per_mcu, ncomp, mcus_per_row;
  int bw[3], bh[3], hs[3], vs[3], first_blk[3];  // first_blk[c]: index of component c's first block inside an MCU
  uint8_t comp_of[16];        // block inside the MCU -> component
  int16_t* coef[3];           // zero-initialised JBLOCK arrays
  const HuffFastTable* ftabs; // DC luma, AC luma, DC chroma, AC chroma
  const HuffFastTable* ttabs; // the same four tables in state-tracking form: bits consumed | zig-zag advance << 5 (make_track_table)
  const uint32_t* vtabs;      // ... and in value form for the write pass, 4 x kHuffValWords words (make_value_table)
  const uint32_t* ptabs;      // ... and in pair form for the tracking passes of the hypothesis scheme, 4 x kHuffPairWords words (make_pair_table)
  const uint32_t* pvtabs;     // ... and the pair form of the VALUE tables for write form 2: 4 x kHuffPairWords x {first symbol, second symbol or 0} (make_pair_value_table)
  const uint8_t* zigzag;
  // hypothesis decode (launch_huffman_decode_hyp): slot s < hyp_h is "started at the subsequence's first bit as block s of
  // an MCU"; slot l * hyp_h + h is the path of hypothesis h of the subsequence l places back that has not merged yet
  int hyp_h, hyp_levels;      // hypotheses per subsequence (= blocks per MCU), overflow depth; slots = (levels + 1) * h <= kHuffHypSlots
  uint64_t* hyp_state;        // [nsub][kHuffHypSlots]: state of the slot's path at the END of the subsequence
  uint8_t* hyp_map;           // [nsub][kHuffHypSlots]: the slot of subsequence i + 1 the path is in at ITS end (0xff: none)
  uint16_t* hyp_cnt;          // [nsub][kHuffHypSlots]: blocks the path completes while crossing subsequence i + 1
  int hyp_hist;               // debug: count the merges per level in flags[10..15], the stragglers' in flags[22..31]
  // round 5: pass 1 runs hyp_main_levels levels in lockstep (0: all of them) and hands the paths still alive to
  // hyp_straggler_kernel, one WAVE per path (lane o looks the symbol at bit o of a 64-bit window up in all four tables, one
  // scalar chain follows the true boundaries with v_readlane): strag_list[n] = {start subsequence, hypothesis | level << 8},
  // n = flags[kHuffFlagStragglers] (zeroed by pass 0)
  int hyp_main_levels;
  int strag_levels;           // the last level a straggler's wave walks (<= hyp_levels); a path still unmerged there is given up (its map stays 0xff)
  uint32_t* strag_list;
  uint32_t strag_cap;
  // round 5: write pass, form 2 (marker-less scans): coefficients go to a scan-order scratch (block t at coef_scan + 64 t, zig-zag
  // order, the DC DIFFERENCE at [0]; zero-initialised) and coef_place_kernel moves them to the component arrays in natural
  // order with the DC prediction applied; nullptr: form 1 (stores into zero-initialised JBLOCK arrays + dcd[])
  int16_t* coef_scan;
  // round 6: pass 0 zero-fills that scratch on the side (its lanes wait on table lookups, the memory system is idle): zero_vec 16-byte
  // pieces from zero_ptr, grid-stride; 0: the host has enqueued a fill instead (the rounds scheme)
  uint4* zero_ptr;
  uint32_t zero_vec;
  // round 6: the write pass runs on PIECES of a subsequence.  It has one lane per subsequence -- 290 waves for a 4K gain map -- and therefore
  // runs at one lane's latency; the tracking passes (pass 1's lockstep levels, the stragglers, pass 0's very first lane) already walk every
  // subsequence along what may be the true path, so they also note the state at the first symbol boundary at or beyond each interior cut
  // (pieces - 1 of them) and the blocks completed inside each piece, per (subsequence, slot) like hyp_cnt; the chain's walk picks the true
  // path's notes (pend / pcnt), and the write pass is launched with pieces x the lanes on pieces of sub_bits / pieces bits.  1: off.
  int pieces;
  uint64_t* mid_state;  // [nsub][kHuffHypSlots][pieces - 1]
  uint16_t* mid_cnt;    // [nsub][kHuffHypSlots][pieces - 1]
  uint64_t* pend;       // [nsub * pieces]: the true path's state at the end of every piece (the write pass's state[] array)
  uint32_t* pcnt;       // [nsub * pieces + 1]: blocks completed inside every piece, then their exclusive scan (the write pass's nblk[] array)
  // restart intervals (nullptr / 0: a scan without markers): see restart_jump in huffman_decode_sync.hip
  const uint32_t* rst_map;    // one bit per byte of the clean stream: an interval starts here
  uint32_t rst_blocks;        // blocks per interval (restart interval x blocks per MCU)
  int* dc_seg;                // [intervals][3]: the components' running DC sums before the interval
  uint32_t* rst_partial;      // [unstuff chunks][3]: markers found | their two sequence sums, per chunk (summed into flags[9], [16], [17])
  uint32_t rst_chunks;
};
constexpr int kHuffHypSlots = 48;
constexpr int kHuffFlagStragglers = 20;  // flags[] word: paths handed to hyp_straggler_kernel
constexpr int kHuffFlagStray = 21;       // flags[] word: != 0 when the uploaded bytes hold a marker other than RSTn / a stuffed zero / a fill byte
// scratch of the chain kernels: per-thread prefix maps, then the tile maps
size_t huff_hyp_chain_bytes(uint64_t nbytes, uint32_t sub_bits, size_t* tiles_offset);
hipError_t launch_huffman_decode_hyp(const HuffSyncArgs& a, int* dc_partial, uint8_t* chain_prefix, uint8_t* chain_tiles, hipStream_t s);
int huff_sync_chunks(uint64_t nbytes);
uint32_t huff_sync_max_subsequences(uint64_t nbytes, uint32_t sub_bits);
// fill: the initial state of the decoder's buffers, written by the FIRST kernel of the decode instead of by fill launches of their own (round 6): zero_vec
// 16-byte pieces of zeros from zero_ptr (flags, block counts, DC differences, restart map), ff_vec pieces of 0xff from ff_ptr (state 0, hypothesis map)
struct HuffInitFill {
  uint4* zero_ptr;
  uint32_t zero_vec;
  uint4* ff_ptr;
  uint32_t ff_vec;
};
hipError_t launch_huffman_unstuff(const uint8_t* data, uint32_t nbytes, uint32_t* chunk_counts, uint32_t* nstuffed_dev, uint8_t* clean, hipStream_t s,
                                  uint32_t* rst_map, uint32_t* rst_partial, const HuffInitFill* fill = nullptr);
hipError_t launch_huffman_decode_sync(const HuffSyncArgs& a, int max_rounds, int* dc_partial, int* final_buf, hipStream_t s);
void launch_profile_mark(hipStream_t s);  // selftest.hip: the empty kernel profilers cut their traces at
int huff_marker_chunks(uint64_t nbytes);
int huff_place_chunk();  // scan positions per DC-prediction chunk of write form 2 (sizes the partial-sum scratch)
// *flag (device-visible, e.g. pinned host memory) |= 1 when data[0, nbytes) holds a 0xFF followed by anything but 0x00 (stuffing),
// RSTn or another 0xFF (fill byte), i.e. a marker that ends the entropy-coded data before nbytes
hipError_t launch_stray_marker_check(const uint8_t* data, uint32_t nbytes, uint32_t* flag, hipStream_t s);
hipError_t launch_huffman_decode(const HuffDecArgs& a, uint32_t* counts, uint32_t* starts, uint32_t* ends, hipStream_t s);

hipError_t launch_idct_dequant(const int16_t* coef, int bw, int bh, const uint16_t* qt_host, uint8_t* plane,
                               size_t stride, hipStream_t s);
hipError_t launch_idct_dequant_rgb(const int16_t* coef_y, const int16_t* coef_cb, const int16_t* coef_cr, int bw, int bh,
                                   const uint16_t* qt_luma_host, const uint16_t* qt_chroma_host, int variant,
                                   const ImageViewMut& rgb, hipStream_t s);
hipError_t launch_jpeg_rgb_to_ycc(const ImageView& rgb, const ImageViewMut& ycc, hipStream_t s);
hipError_t launch_jpeg_ycc_to_rgb(const ImageView& ycc, const ImageViewMut& rgb, int variant, hipStream_t s);

}  // namespace uhdr
