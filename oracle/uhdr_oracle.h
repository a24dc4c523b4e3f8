/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the gain-map hot path.
 *
 * Plain-C restatement of the reference's per-pixel gain-map math and the libjpeg "islow"
 * FDCT/quantize stage.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product library (libultrahdr_amd/lib/libuhdr_hip.so) never links or calls it.
 *
 * Struct layouts are byte-identical to the reference's public C structs so the same ctypes
 * objects can be handed to the oracle, to oracle/_ref (the real reference) and to the HIP library:
 *   uo_image_t    == uhdr_raw_image_t         (/root/reference/ultrahdr_api.h:227-246)
 *   uo_metadata_t == uhdr_gainmap_metadata_t  (/root/reference/ultrahdr_api.h:266-283)
 */
#ifndef UHDR_ORACLE_H
#define UHDR_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* enum values: /root/reference/ultrahdr_api.h:108-160 */
enum {
  UO_FMT_P010 = 0,
  UO_FMT_YUV420 = 1,
  UO_FMT_Y400 = 2,
  UO_FMT_RGBA8888 = 3,
  UO_FMT_RGBAF16 = 4,
  UO_FMT_RGBA1010102 = 5,
  UO_FMT_YUV444 = 6,
  UO_FMT_YUV422 = 7,
  UO_FMT_RGB888 = 11,
  UO_FMT_YUV444_10 = 12
};
enum { UO_CG_UNSPEC = -1, UO_CG_709 = 0, UO_CG_P3 = 1, UO_CG_2100 = 2 };
enum { UO_CT_UNSPEC = -1, UO_CT_LINEAR = 0, UO_CT_HLG = 1, UO_CT_PQ = 2, UO_CT_SRGB = 3 };
enum { UO_CR_UNSPEC = -1, UO_CR_LIMITED = 0, UO_CR_FULL = 1 };
enum { UO_OK = 0, UO_ERROR = 1, UO_INVALID_PARAM = 3, UO_MEM_ERROR = 4, UO_UNSUPPORTED = 6 };
enum { UO_PRESET_REALTIME = 0, UO_PRESET_BEST_QUALITY = 1 };

typedef struct uo_image {
  int fmt, cg, ct, range;
  unsigned w, h;
  void* planes[3];
  unsigned stride[3]; /* in pixels */
} uo_image_t;

typedef struct uo_metadata {
  float max_content_boost[3];
  float min_content_boost[3];
  float gamma[3];
  float offset_sdr[3];
  float offset_hdr[3];
  float hdr_capacity_min;
  float hdr_capacity_max;
  int use_base_cg;
} uo_metadata_t;

/* encoder knobs = UltraHdr ctor args + generateGainMap's two flags
 * (/root/reference/lib/include/ultrahdr/ultrahdrcommon.h:450-457, 507-510) */
typedef struct uo_encode_cfg {
  int scale;
  int multichannel;
  float gamma;
  int preset;
  float min_boost;   /* FLT_MIN = unset */
  float max_boost;   /* FLT_MAX = unset */
  float target_nits; /* -1 = unset */
  int sdr_is_601;
  int use_luminance;
} uo_encode_cfg_t;

/* stage operators (return uhdr_codec_err_t values) */
int uo_apply_gainmap(const uo_image_t* sdr, const uo_image_t* gm, const uo_metadata_t* md,
                     int out_ct, int out_fmt, float max_display_boost, uo_image_t* dest);
/* gm_out->planes[0] caller-allocated, stride taken from gm_out->stride[0] (>= map width). */
int uo_generate_gainmap(const uo_image_t* sdr, const uo_image_t* hdr, const uo_encode_cfg_t* cfg,
                        uo_metadata_t* md_out, uo_image_t* gm_out);
/* the two-pass mode split at its only exchange step (jpegr.cpp:932-938), for row-stripe tests */
int uo_generate_gainmap_pass1(const uo_image_t* sdr, const uo_image_t* hdr, const uo_encode_cfg_t* cfg,
                              float* gain_log2, float minmax[6], int* use_base_cg);
void uo_generate_gainmap_pass2(const float* gain_log2, const float minmax[6], float gamma, int nch,
                               unsigned mw, unsigned mh, uint8_t* out, size_t out_stride);
int uo_tone_map(const uo_image_t* hdr, uo_image_t* sdr);
int uo_convert_yuv(uo_image_t* img, int src_cg, int dst_cg);
/* dst planes/strides caller-provided; dst->fmt decides the variant like the reference does. */
int uo_convert_raw_input_to_ycbcr(const uo_image_t* src, int chroma_sampling, uo_image_t* dst);

/* copy_raw_image(src, dst) (gainmapmath.cpp:1492-1613): equal formats, RGB888 -> RGBA8888, RGBA8888 -> Y400 */
int uo_copy_raw_image(const uo_image_t* src, uo_image_t* dst);

/* JPEG stage: quality -> quant table (natural order), then islow FDCT + quantize of one u8 plane.
 * coef: blocks in raster order, 64 int16 each, natural (row-major) order = libjpeg JBLOCK layout.
 * The plane is read for blocks_w*8 x blocks_h*8 samples: callers pad exactly as the reference's
 * helper does (jpegencoderhelper.cpp:246-309). */
void uo_jpeg_quant_table(int quality, int is_chroma, uint16_t qt[64]);
void uo_fdct_quant_plane(const uint8_t* plane, size_t stride, int blocks_w, int blocks_h,
                         const uint16_t qt[64], int16_t* coef);
/* JCS_RGB -> YCbCr as libjpeg's jccolor.c does for 3-channel gain maps (fixed point, 16 bit) */
void uo_jpeg_rgb_to_ycc(const uint8_t* rgb, size_t stride_px, int w, int h, uint8_t* y,
                        uint8_t* cb, uint8_t* cr, size_t out_stride);

/* JPEG decode stage: dequantize + islow IDCT + range limit of coefficient blocks (JBLOCK layout, as
 * uo_fdct_quant_plane writes them) into an 8-bit plane; libjpeg's YCbCr -> RGB for 3-channel maps
 * (variant 0: 6b / libjpeg-turbo constants, 1: IJG 9 constants; out_bpp 3 or 4). */
void uo_idct_dequant_plane(const int16_t* coef, int blocks_w, int blocks_h, const uint16_t qt[64],
                           uint8_t* plane, size_t stride);
void uo_jpeg_ycc_to_rgb(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, size_t in_stride, int w,
                        int h, uint8_t* rgb, size_t out_stride_px, int out_bpp, int variant);

/* Entropy stage: baseline Huffman coding of quantized coefficient blocks with the Annex K tables, as
 * libjpeg does behind JpegEncoderHelper (jpegencoderhelper.cpp:131-244), optionally with restart intervals.
 * coef[c]: JBLOCK arrays of bw[c] x bh[c] REAL blocks (width_in_blocks x height_in_blocks); blocks an MCU needs beyond
 * them are libjpeg's dummy blocks.  hs / vs: sampling factors (ignored for ncomp == 1).  restart_interval in MCUs, 0 = none. */
typedef struct uo_scan {
  int ncomp;
  const int16_t* coef[3];
  int bw[3], bh[3];
  int hs[3], vs[3];
  unsigned w, h;
  int restart_interval;
} uo_scan_t;
size_t uo_huffman_encode_scan(const uo_scan_t* sc, uint8_t* out, size_t cap); /* bytes between the SOS header and EOI; 0 = overflow */
size_t uo_jpeg_assemble(const uo_scan_t* sc, const uint16_t qt[2][64], const uint8_t* scan, size_t scan_len, uint8_t* out, size_t cap);
void uo_std_huff_table(int is_ac, int is_chroma, uint8_t bits[17], uint8_t vals[256], int* nvals);
/* inverse: entropy-coded data (RSTn markers included) -> coefficient blocks; tables as DHT content, order DC luma, AC luma,
 * DC chroma, AC chroma; coef_out[c] has room for bw[c] x bh[c] blocks.  0 on success. */
int uo_huffman_decode_scan(const uo_scan_t* sc, const uint8_t dht_bits[4][17], const uint8_t dht_vals[4][256], const uint8_t* data,
                           size_t size, int16_t* coef_out[3]);

/* scalar access for known-answer tests */
int uo_eval(int fn, const float* in, float* out, size_t n); /* ids as in ref_shim.cpp */
void uo_float_to_half(const float* in, uint16_t* out, size_t n);
void uo_oetf_code(int ct, const float* in, uint32_t* out, size_t n); /* HLG / PQ tail: clamped linear -> 10-bit code */
uint32_t uo_color_to_rgba1010102(float r, float g, float b);
uint64_t uo_color_to_rgbaf16(float r, float g, float b);
float uo_compute_gain(float sdr, float hdr);
uint8_t uo_affine_map_gain(float g, float mn, float mx, float gamma);
uint8_t uo_encode_gain(float y_sdr, float y_hdr, float min_boost, float max_boost, float gamma);
void uo_apply_gain(const float e[3], float gain, const uo_metadata_t* md, float weight,
                   int use_lut, float out[3]);
void uo_idw_weights(int scale, int which, float* out);
void uo_color_fn(int fn, const float in[3], float out[3]);
void uo_lut(int which, float* out); /* 0 srgb-inv(1024) 1 hlg-inv(4096) 2 pq-inv(4096) 3 hlg(65536) 4 pq(65536) */

#ifdef __cplusplus
}
#endif
#endif
