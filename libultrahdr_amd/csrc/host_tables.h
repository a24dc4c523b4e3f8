// Host-side table builders.  The look-up tables the reference builds lazily on the CPU
// (LookUpTable, GainLUT, ShepardsIDW) are built here with the host's own libm -- the same libm the
// reference would call on this machine -- and uploaded, because device transcendental functions
// do not round identically to glibc's.  See host_tables.cpp for the per-call-site float/double
// notes.
#pragma once
#include <functional>
#include <vector>

#include "uhdr_types.h"

namespace uhdr {
namespace host {

const std::vector<float>& srgb_inv_oetf_lut();  // 1024  (gainmapmath.cpp:126-131)
const std::vector<float>& hlg_inv_oetf_lut();   // 4096  (gainmapmath.cpp:271-277)
const std::vector<float>& pq_inv_oetf_lut();    // 4096  (gainmapmath.cpp:339-345)
const std::vector<float>& hlg_inv_oetf_ootf_lut();  // 4096: hlgInvOetfLUT then hlgOotfApprox per node
const std::vector<float>& hlg_oetf_lut();       // 65536 (gainmapmath.cpp:248-254)
const std::vector<float>& pq_oetf_lut();        // 65536 (gainmapmath.cpp:320-326)
const std::vector<float>& pq_oetf_code_lut();   // 65536 uint16 codes (pqOetfLUT then colorToRgba1010102), packed two per float

// 10-bit output-code thresholds of the HLG / PQ tail (kOetfThrN floats; see host_tables.cpp)
const std::vector<float>& oetf_code_thresholds(int ct);
uint32_t oetf_code(int ct, float v);  // the composite itself, evaluated with the host libm
// the same step function as a bucket table for the quad kernel (see host_tables.cpp)
struct OetfBuckets {
  uint32_t shift = 0, base = 0, n = 0;  // bucket k covers bit patterns [(base + k) << shift, (base + k + 1) << shift)
  uint32_t lo_bits = 0, hi_bits = 0;    // the domain: the device clamps the bit pattern into it first
  uint32_t clamp_lo_bits = 0;           // max(lo_bits, base << shift): the device's lower clamp, so that bucket - base never underflows
  bool exact = false;                   // construction verified (one threshold per bucket, replay against the composite)
  std::vector<uint32_t> entries;        // n x {thr, lo | hi << 16}
};
OetfBuckets build_step_table(const std::function<uint32_t(uint32_t)>& code_of_bits, uint32_t lo_bits, uint32_t hi_bits, uint32_t shift, uint32_t capacity);
const OetfBuckets& oetf_code_buckets(int ct, bool prescaled = false);  // prescaled: argument is the value before (x * 203) / peak
// encode side (see host_tables.cpp): toneMap's sRGB byte, encodeGain's byte, RGBA1010102 code -> linear value
const OetfBuckets& srgb_code8_buckets();
OetfBuckets gain_code8_buckets(float min_boost, float max_boost, float log2min, double log2_range, double log2_range_rcp);
std::vector<float> lin10_table(const float* lut, int n);
float gain_cap_ratio();                            // computeGain's 2.3f cap in the ratio domain (0: not representable)
const std::vector<float>& srgb_inv_oetf_of_byte(); // 256: byte -> byte / 255.0f -> sRGB inverse-OETF table value

// float64 tables of exact_math.h (table-driven pow / log2 of the encode path)
const std::vector<double>& math_tables();

Yuv2Rgb yuv2rgb_coeffs(int cg);  // gainmapmath.cpp:94,104-105 / 164,174-175 / 194,226-227
Rgb2Yuv rgb2yuv_coeffs(int cg);
void luminance_coeffs(int cg, float out[3]);  // gainmapmath.cpp:86, 156, 187
// getGamutConversionFn(dst, src) (gainmapmath.cpp:1087-1129): returns false for UNSPECIFIED;
// *identity set when dst == src.
bool gamut_matrix(int dst_cg, int src_cg, Mat3* out, bool* identity);
// convertYuv coefficient choice (jpegr.cpp:436-502): 0 ok, 1 identity (nothing to do), <0 error
int yuv_encoding_matrix(int src_cg, int dst_cg, Mat3* out);

float reference_peak_nits(int ct);  // getReferenceDisplayPeakLuminanceInNits (gainmapmath.cpp:20-34)

bool metadata_channels_identical(const uhdr_gainmap_metadata_t& m);  // ultrahdrcommon.h:218-226
// gainmap_weight from the display boost (jpegr.cpp:1678-1689; log2f because jpegr.cpp has
// `using namespace std`)
float gainmap_weight(const uhdr_gainmap_metadata_t& m, float max_display_boost);
// Fills the ApplyTables block (uhdr_types.h) for one applyGainMap call.
void build_apply_tables(const uhdr_gainmap_metadata_t& m, float weight, int idw_scale,
                        std::vector<float>* out);
void fill_idw(float* w, int s, int inc_r, int inc_b);  // gainmapmath.cpp:43-80

// libjpeg quality -> quant table (jcparam.c jpeg_quality_scaling / jpeg_add_quant_table with
// force_baseline; natural order)
void jpeg_quant_table(int quality, int is_chroma, uint16_t qt[64]);

// baseline Huffman stage (T.81 Annex K.3 tables, Annex C codes)
constexpr int kHuffTabWords = 2 * (16 + 256);
const uint8_t* jpeg_zigzag_to_natural();  // 64 entries
int jpeg_std_huff_table(int is_ac, int is_chroma, uint8_t bits[17], uint8_t vals[256]);  // returns the number of symbols
const std::vector<uint32_t>& jpeg_huff_code_tables();  // kHuffTabWords packed entries (length << 16 | code)

}  // namespace host
}  // namespace uhdr
