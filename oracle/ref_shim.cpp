// TEST INFRASTRUCTURE ONLY -- never linked into, imported by, or shipped with the product path.
//
// ref_shim: a plain-C doorway into the *real* reference (google/libultrahdr 2.0.2), which
// oracle/Makefile compiles from the sources where they lie under /root/reference (outputs only
// into oracle/_ref/).  Nothing of the reference is copied into this repository: this file only
// *calls* the reference's public C++ entry points so that tests can
//   (1) pin the C restatement in oracle/uhdr_oracle.c against the reference itself, and
//   (2) time the reference's CPU path (bench.py cpu_baseline.kind == "reference").
//
// Reference entry points used (file:line under /root/reference):
//   UltraHdr::toneMap            lib/src/jpegr.cpp:1985      (decl ultrahdrcommon.h:482)
//   UltraHdr::generateGainMap    lib/src/jpegr.cpp:530       (decl ultrahdrcommon.h:507)
//   UltraHdr::applyGainMap       lib/src/jpegr.cpp:1533      (decl ultrahdrcommon.h:531)
//   UltraHdr::convertYuv         lib/src/jpegr.cpp:436       (decl ultrahdrcommon.h:545)
//   convert_raw_input_to_ycbcr   lib/src/gainmapmath.cpp:1291
//   scalar math                  lib/src/gainmapmath.cpp / lib/include/ultrahdr/gainmapmath.h
//   JpegEncoderHelper            lib/src/jpegencoderhelper.cpp:101 (-> external libjpeg)
//   uhdr_encode / uhdr_decode    lib/src/ultrahdr_api.cpp:1200, 1918 (whole-API CPU baseline)
#include <csetjmp>
#include <cstdio>
#include <cstring>
#include <memory>

#include "ultrahdr/gainmapmath.h"
#include "ultrahdr/jpegencoderhelper.h"
#include "ultrahdr/jpegdecoderhelper.h"
#include "ultrahdr/jpegr.h"
#include "ultrahdr/ultrahdrcommon.h"
#include "ultrahdr_api.h"

using namespace ultrahdr;

#define REF_API extern "C" __attribute__((visibility("default")))

static int finish(const uhdr_error_info_t& st, char* detail) {
  if (detail) {
    if (st.has_detail) {
      strncpy(detail, st.detail, 255);
      detail[255] = 0;
    } else {
      detail[0] = 0;
    }
  }
  return (int)st.error_code;
}

struct ref_encode_cfg {
  int scale;          // mMapDimensionScaleFactor
  int multichannel;   // mUseMultiChannelGainMap
  float gamma;        // mGamma
  int preset;         // uhdr_enc_preset_t
  float min_boost;    // FLT_MIN == unset
  float max_boost;    // FLT_MAX == unset
  float target_nits;  // -1 == unset
  int sdr_is_601;
  int use_luminance;
};

REF_API int ref_apply_gainmap(uhdr_raw_image_t* sdr, uhdr_raw_image_t* gm,
                              const uhdr_gainmap_metadata_t* md, int out_ct, int out_fmt,
                              float max_display_boost, uhdr_raw_image_t* dest, char* detail) {
  uhdr_gainmap_metadata_t tmp = *md;
  uhdr_gainmap_metadata_ext_t ext(tmp, kJpegrVersion);
  UltraHdr u;
  return finish(u.applyGainMap(sdr, gm, &ext, (uhdr_color_transfer_t)out_ct,
                               (uhdr_img_fmt_t)out_fmt, max_display_boost, dest),
                detail);
}

// gm_out->planes[0] must point at a caller buffer of >= ALIGN64(w/scale) * (h/scale) * 3 bytes;
// on return fmt/w/h/stride describe the reference's own (64-aligned) layout.
REF_API int ref_generate_gainmap(uhdr_raw_image_t* sdr, uhdr_raw_image_t* hdr,
                                 const ref_encode_cfg* cfg, uhdr_gainmap_metadata_t* md_out,
                                 uhdr_raw_image_t* gm_out, char* detail) {
  UltraHdr u(nullptr, cfg->scale, 95, cfg->multichannel != 0, cfg->gamma,
             (uhdr_enc_preset_t)cfg->preset, cfg->min_boost, cfg->max_boost, cfg->target_nits);
  uhdr_gainmap_metadata_ext_t ext(kJpegrVersion);
  std::unique_ptr<uhdr_raw_image_ext_t> gm;
  uhdr_error_info_t st =
      u.generateGainMap(sdr, hdr, &ext, gm, cfg->sdr_is_601 != 0, cfg->use_luminance != 0);
  if (st.error_code == UHDR_CODEC_OK) {
    *md_out = static_cast<uhdr_gainmap_metadata_t&>(ext);
    void* buf = gm_out->planes[0];
    gm_out->fmt = gm->fmt;
    gm_out->cg = gm->cg;
    gm_out->ct = gm->ct;
    gm_out->range = gm->range;
    gm_out->w = gm->w;
    gm_out->h = gm->h;
    gm_out->stride[0] = gm->stride[0];
    gm_out->stride[1] = gm_out->stride[2] = 0;
    gm_out->planes[1] = gm_out->planes[2] = nullptr;
    size_t bpp = gm->fmt == UHDR_IMG_FMT_24bppRGB888 ? 3 : 1;
    memcpy(buf, gm->planes[0], bpp * gm->stride[0] * gm->h);
  }
  return finish(st, detail);
}

REF_API int ref_tone_map(uhdr_raw_image_t* hdr, uhdr_raw_image_t* sdr, char* detail) {
  UltraHdr u;
  return finish(u.toneMap(hdr, sdr), detail);
}

REF_API int ref_convert_yuv(uhdr_raw_image_t* img, int src_cg, int dst_cg, char* detail) {
  UltraHdr u;
  return finish(u.convertYuv(img, (uhdr_color_gamut_t)src_cg, (uhdr_color_gamut_t)dst_cg), detail);
}

// dst planes must be caller-allocated with 64-aligned strides exactly as the reference allocates
// (uhdr_raw_image_ext, lib/src/ultrahdr_api.cpp:55-117); the result is copied plane by plane.
REF_API int ref_convert_raw_input_to_ycbcr(uhdr_raw_image_t* src, int chroma_sampling,
                                           uhdr_raw_image_t* dst) {
  std::unique_ptr<uhdr_raw_image_ext_t> out = convert_raw_input_to_ycbcr(src, chroma_sampling != 0);
  if (!out) return (int)UHDR_CODEC_UNSUPPORTED_FEATURE;
  dst->fmt = out->fmt;
  dst->cg = out->cg;
  dst->ct = out->ct;
  dst->range = out->range;
  dst->w = out->w;
  dst->h = out->h;
  size_t bps = (out->fmt == UHDR_IMG_FMT_24bppYCbCrP010 || out->fmt == UHDR_IMG_FMT_30bppYCbCr444)
                   ? 2
                   : 1;
  for (int p = 0; p < 3; p++) {
    dst->stride[p] = out->stride[p];
    if (!out->planes[p] || !dst->planes[p]) continue;
    size_t rows = out->h;
    if ((out->fmt == UHDR_IMG_FMT_24bppYCbCrP010 || out->fmt == UHDR_IMG_FMT_12bppYCbCr420) && p)
      rows = out->h / 2;
    memcpy(dst->planes[p], out->planes[p], bps * out->stride[p] * rows);
  }
  return 0;
}

// copy_raw_image(src, dst) (lib/src/gainmapmath.cpp:1492-1613) on caller-provided descriptors
REF_API int ref_copy_raw_image(uhdr_raw_image_t* src, uhdr_raw_image_t* dst) {
  return (int)copy_raw_image(src, dst).error_code;
}

// ---- scalar / small-vector access to gainmapmath for KATs --------------------------------------
enum {
  REF_FN_SRGB_INVOETF = 0,
  REF_FN_SRGB_INVOETF_LUT,
  REF_FN_SRGB_OETF,
  REF_FN_HLG_OETF,
  REF_FN_HLG_OETF_LUT,
  REF_FN_HLG_INVOETF,
  REF_FN_HLG_INVOETF_LUT,
  REF_FN_PQ_OETF,
  REF_FN_PQ_OETF_LUT,
  REF_FN_PQ_INVOETF,
  REF_FN_PQ_INVOETF_LUT,
  REF_FN_HALF_TO_FLOAT,  // input: uint16 stored as float value
  REF_FN_HLG_OOTF_APPROX,     // per channel pow(x, 1.2)
  REF_FN_HLG_INV_OOTF_APPROX  // per channel pow(x, 1/1.2)
};

REF_API int ref_eval(int fn, const float* in, float* out, size_t n) {
  for (size_t i = 0; i < n; i++) {
    float x = in[i];
    switch (fn) {
      case REF_FN_SRGB_INVOETF: out[i] = srgbInvOetf(x); break;
      case REF_FN_SRGB_INVOETF_LUT: out[i] = srgbInvOetfLUT(x); break;
      case REF_FN_SRGB_OETF: out[i] = srgbOetf(x); break;
      case REF_FN_HLG_OETF: out[i] = hlgOetf(x); break;
      case REF_FN_HLG_OETF_LUT: out[i] = hlgOetfLUT(x); break;
      case REF_FN_HLG_INVOETF: out[i] = hlgInvOetf(x); break;
      case REF_FN_HLG_INVOETF_LUT: out[i] = hlgInvOetfLUT(x); break;
      case REF_FN_PQ_OETF: out[i] = pqOetf(x); break;
      case REF_FN_PQ_OETF_LUT: out[i] = pqOetfLUT(x); break;
      case REF_FN_PQ_INVOETF: out[i] = pqInvOetf(x); break;
      case REF_FN_PQ_INVOETF_LUT: out[i] = pqInvOetfLUT(x); break;
      case REF_FN_HALF_TO_FLOAT: out[i] = halfToFloat((uint16_t)x); break;
      case REF_FN_HLG_OOTF_APPROX: {
        Color c = hlgOotfApprox({{{x, x, x}}}, nullptr);
        out[i] = c.r;
        break;
      }
      case REF_FN_HLG_INV_OOTF_APPROX: {
        Color c = hlgInverseOotfApprox({{{x, x, x}}});
        out[i] = c.r;
        break;
      }
      default: return -1;
    }
  }
  return 0;
}

REF_API void ref_float_to_half(const float* in, uint16_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = floatToHalf(in[i]);
}
REF_API uint32_t ref_color_to_rgba1010102(float r, float g, float b) {
  return colorToRgba1010102({{{r, g, b}}});
}
REF_API uint64_t ref_color_to_rgbaf16(float r, float g, float b) {
  return colorToRgbaF16({{{r, g, b}}});
}
REF_API float ref_compute_gain(float sdr, float hdr) { return computeGain(sdr, hdr); }
REF_API uint8_t ref_affine_map_gain(float g, float mn, float mx, float gamma) {
  return affineMapGain(g, mn, mx, gamma);
}
REF_API uint8_t ref_encode_gain(float y_sdr, float y_hdr, float min_boost, float max_boost,
                                float gamma) {
  uhdr_gainmap_metadata_ext_t md(kJpegrVersion);
  for (int i = 0; i < 3; i++) {
    md.min_content_boost[i] = min_boost;
    md.max_content_boost[i] = max_boost;
    md.gamma[i] = gamma;
  }
  return encodeGain(y_sdr, y_hdr, &md, 0);
}
// out[0..2] = applyGain / applyGainLUT of colour e with single gain value.
REF_API void ref_apply_gain(const float e[3], float gain, const uhdr_gainmap_metadata_t* md,
                            float weight, int use_lut, float out[3]) {
  uhdr_gainmap_metadata_t tmp = *md;
  uhdr_gainmap_metadata_ext_t ext(tmp, kJpegrVersion);
  Color c = {{{e[0], e[1], e[2]}}};
  Color r;
  if (use_lut) {
    GainLUT lut(&ext, weight);
    r = applyGainLUT(c, gain, lut, &ext);
  } else {
    r = applyGain(c, gain, &ext, weight);
  }
  out[0] = r.r;
  out[1] = r.g;
  out[2] = r.b;
}
REF_API void ref_idw_weights(int scale, int which, float* out) {
  ShepardsIDW t(scale);
  const float* src = which == 0 ? t.mWeights : which == 1 ? t.mWeightsNR
                     : which == 2 ? t.mWeightsNB : t.mWeightsC;
  memcpy(out, src, sizeof(float) * scale * scale * 4);
}
REF_API void ref_color_fn(int fn, const float in[3], float out[3]) {
  Color c = {{{in[0], in[1], in[2]}}};
  Color r = c;
  switch (fn) {
    case 0: r = srgbYuvToRgb(c); break;
    case 1: r = p3YuvToRgb(c); break;
    case 2: r = bt2100YuvToRgb(c); break;
    case 3: r = srgbRgbToYuv(c); break;
    case 4: r = p3RgbToYuv(c); break;
    case 5: r = bt2100RgbToYuv(c); break;
    case 6: r = bt709ToP3(c); break;
    case 7: r = bt709ToBt2100(c); break;
    case 8: r = p3ToBt709(c); break;
    case 9: r = p3ToBt2100(c); break;
    case 10: r = bt2100ToBt709(c); break;
    case 11: r = bt2100ToP3(c); break;
    case 12: r = {{{srgbLuminance(c), p3Luminance(c), bt2100Luminance(c)}}}; break;
  }
  out[0] = r.r;
  out[1] = r.g;
  out[2] = r.b;
}

// ---- JPEG (external libjpeg behind the reference's helper) --------------------------------------
// Compress with the reference helper (jpegencoderhelper.cpp:101); returns bytes written or <0.
REF_API long ref_jpeg_compress(const uhdr_raw_image_t* img, int quality, uint8_t* out, size_t cap) {
  JpegEncoderHelper enc;
  uhdr_error_info_t st = enc.compressImage(img, quality, nullptr, 0);
  if (st.error_code != UHDR_CODEC_OK) return -(long)st.error_code;
  uhdr_compressed_image_t c = enc.getCompressedImage();
  if (c.data_sz > cap) return -100;
  memcpy(out, c.data, c.data_sz);
  return (long)c.data_sz;
}

struct ref_jerr {
  jpeg_error_mgr pub;
  jmp_buf jb;
};
static void ref_jerr_exit(j_common_ptr c) { longjmp(((ref_jerr*)c->err)->jb, 1); }

// Read back the quantized DCT coefficients of a JPEG produced above (libjpeg
// jpeg_read_coefficients). coef[c] receives blocks in raster order, 64 int16 each, natural
// (row-major, NOT zig-zag) order as libjpeg stores them; qt[c] the 64-entry quant table
// (natural order); blocks_w/h[c] the per-component block grid actually stored.
REF_API int ref_jpeg_read_coefficients(const uint8_t* data, size_t size, int16_t* coef[3],
                                       uint16_t qt[3][64], int blocks_w[3], int blocks_h[3],
                                       int* ncomp) {
  jpeg_decompress_struct cinfo;
  ref_jerr jerr;
  cinfo.err = jpeg_std_error(&jerr.pub);
  jerr.pub.error_exit = ref_jerr_exit;
  if (setjmp(jerr.jb)) {
    jpeg_destroy_decompress(&cinfo);
    return -1;
  }
  jpeg_create_decompress(&cinfo);
  jpeg_mem_src(&cinfo, const_cast<uint8_t*>(data), (unsigned long)size);
  jpeg_read_header(&cinfo, TRUE);
  jvirt_barray_ptr* arrays = jpeg_read_coefficients(&cinfo);
  *ncomp = cinfo.num_components;
  for (int c = 0; c < cinfo.num_components && c < 3; c++) {
    jpeg_component_info* ci = &cinfo.comp_info[c];
    blocks_w[c] = (int)ci->width_in_blocks;
    blocks_h[c] = (int)ci->height_in_blocks;
    for (int i = 0; i < 64; i++) qt[c][i] = ci->quant_table->quantval[i];
    if (!coef[c]) continue;
    for (JDIMENSION by = 0; by < ci->height_in_blocks; by++) {
      JBLOCKARRAY rows =
          (*cinfo.mem->access_virt_barray)((j_common_ptr)&cinfo, arrays[c], by, 1, FALSE);
      memcpy(coef[c] + (size_t)by * ci->width_in_blocks * 64, rows[0],
             sizeof(int16_t) * 64 * ci->width_in_blocks);
    }
  }
  jpeg_finish_decompress(&cinfo);
  jpeg_destroy_decompress(&cinfo);
  return 0;
}

// Decode with the reference helper (jpegdecoderhelper.cpp:169): planar YCbCr for the base image
// (mode 0, DECODE_TO_YCBCR_CS) or the stream's own colour space for gain maps (mode 1).
REF_API int ref_jpeg_decompress(const uint8_t* data, size_t size, int mode, uhdr_raw_image_t* dst,
                                uint8_t* buf, size_t cap) {
  JpegDecoderHelper dec;
  uhdr_error_info_t st = dec.decompressImage(
      data, size, mode == 0 ? DECODE_TO_YCBCR_CS : DECODE_STREAM);
  if (st.error_code != UHDR_CODEC_OK) return -(int)st.error_code;
  uhdr_raw_image_t img = dec.getDecompressedImage();
  *dst = img;
  size_t off = 0;
  for (int p = 0; p < 3; p++) {
    if (!img.planes[p]) continue;
    size_t rows = img.h, bpp = 1;
    if (img.fmt == UHDR_IMG_FMT_12bppYCbCr420 && p) rows = img.h / 2;
    if (img.fmt == UHDR_IMG_FMT_24bppRGB888) bpp = 3;
    if (img.fmt == UHDR_IMG_FMT_32bppRGBA8888) bpp = 4;
    size_t sz = bpp * img.stride[p] * rows;
    if (off + sz > cap) return -100;
    memcpy(buf + off, img.planes[p], sz);
    dst->planes[p] = buf + off;
    off += sz;
  }
  return 0;
}

// ---- the whole public API (ultrahdr_api.cpp:1200 uhdr_encode, :1918 uhdr_decode): CPU baseline of the end-to-end calls ----
// API-1 (raw HDR + raw SDR) or API-0 (sdr == nullptr) encode with the C API's own defaults except what is passed.
// Returns the stream size (> 0), or -(error code).
REF_API long ref_uhdr_encode(uhdr_raw_image_t* hdr, uhdr_raw_image_t* sdr, int quality, int preset, uint8_t* out, size_t cap) {
  uhdr_codec_private_t* enc = uhdr_create_encoder();
  if (!enc) return -1000;
  uhdr_error_info_t st = uhdr_enc_set_raw_image(enc, hdr, UHDR_HDR_IMG);
  if (st.error_code == UHDR_CODEC_OK && sdr) st = uhdr_enc_set_raw_image(enc, sdr, UHDR_SDR_IMG);
  if (st.error_code == UHDR_CODEC_OK) st = uhdr_enc_set_quality(enc, quality, UHDR_BASE_IMG);
  if (st.error_code == UHDR_CODEC_OK) st = uhdr_enc_set_preset(enc, (uhdr_enc_preset_t)preset);
  if (st.error_code == UHDR_CODEC_OK) st = uhdr_encode(enc);
  long n = -(long)st.error_code;
  if (st.error_code == UHDR_CODEC_OK) {
    uhdr_compressed_image_t* o = uhdr_get_encoded_stream(enc);
    if (o && o->data_sz <= cap) {
      memcpy(out, o->data, o->data_sz);
      n = (long)o->data_sz;
    } else {
      n = -100;
    }
  }
  uhdr_release_encoder(enc);
  return n;
}
// dest: caller-allocated packed buffer of w * h * (8 for RGBA_F16, 4 otherwise) bytes.  Returns 0 or -(error code).
REF_API int ref_uhdr_decode(const uint8_t* data, size_t size, int out_ct, int out_fmt, uint8_t* dest, size_t cap, int* w, int* h) {
  uhdr_codec_private_t* dec = uhdr_create_decoder();
  if (!dec) return -1000;
  uhdr_compressed_image_t in;
  memset(&in, 0, sizeof in);
  in.data = const_cast<uint8_t*>(data);
  in.data_sz = in.capacity = size;
  uhdr_error_info_t st = uhdr_dec_set_image(dec, &in);
  if (st.error_code == UHDR_CODEC_OK) st = uhdr_dec_set_out_color_transfer(dec, (uhdr_color_transfer_t)out_ct);
  if (st.error_code == UHDR_CODEC_OK) st = uhdr_dec_set_out_img_format(dec, (uhdr_img_fmt_t)out_fmt);
  if (st.error_code == UHDR_CODEC_OK) st = uhdr_decode(dec);
  int rc = -(int)st.error_code;
  if (st.error_code == UHDR_CODEC_OK) {
    uhdr_raw_image_t* o = uhdr_get_decoded_image(dec);
    const size_t bpp = o->fmt == UHDR_IMG_FMT_64bppRGBAHalfFloat ? 8 : 4;
    *w = (int)o->w;
    *h = (int)o->h;
    if (dest && (size_t)o->w * o->h * bpp <= cap) {
      for (unsigned y = 0; y < o->h; y++)
        memcpy(dest + (size_t)y * o->w * bpp, (const uint8_t*)o->planes[0] + (size_t)y * o->stride[0] * bpp, (size_t)o->w * bpp);
    } else if (dest) {
      rc = -100;
    }
  }
  uhdr_release_decoder(dec);
  return rc;
}

REF_API const char* ref_info() {
  static char s[256];
  snprintf(s, sizeof s, "libultrahdr %s (reference sources) + libjpeg JPEG_LIB_VERSION %d",
           UHDR_LIB_VERSION_STR, JPEG_LIB_VERSION);
  return s;
}
