// uhdr_hip_seam.h -- the HIP seam of the 43-symbol libuhdr facade.
//
// The facade is the REFERENCE's own control plane (C API state machine, containers, metadata, libjpeg
// entropy coding: everything SURVEY.md marks out of scope) compiled from /root/reference at build time with
// facade/reference_hip_seam.patch applied out of tree; the patch adds one call into this header at each
// place where the reference enters its per-pixel hot path.  Nothing of the reference is stored in this
// repository.  With uhdr_enable_gpu_acceleration(codec, 1) (ultrahdr_api.h:849) the calls below route the
// stage to libuhdr_hip.so; without it the library behaves exactly like the reference.
//
//   reference site (file:line)                                   seam call
//   UltraHdr::applyGainMap          lib/src/jpegr.cpp:1533       uhdr_hip_seam::apply_gainmap
//   UltraHdr::generateGainMap       lib/src/jpegr.cpp:530        uhdr_hip_seam::generate_gainmap
//   UltraHdr::toneMap               lib/src/jpegr.cpp:1985       uhdr_hip_seam::tone_map
//   JpegR::encodeJPEGR (API-1)      lib/src/jpegr.cpp:253        uhdr_hip_seam::encode_api1 (the whole sample -> bytes part of the
//                                                                call in one device sequence; the four seams above and below
//                                                                take over for what it declines)
//   UltraHdr::convertYuv            lib/src/jpegr.cpp:436        uhdr_hip_seam::convert_yuv
//   convert_raw_input_to_ycbcr      lib/src/gainmapmath.cpp:1291 uhdr_hip_seam::convert_raw_input_to_ycbcr
//   JpegEncoderHelper::encode       lib/src/jpegencoderhelper.cpp:131  uhdr_hip_seam::fdct_planes (FDCT + quantize on
//                                                                the device, libjpeg keeps the entropy coding)
//   JpegDecoderHelper::decode       lib/src/jpegdecoderhelper.cpp:283  uhdr_hip_seam::idct_planes (libjpeg keeps the
//                                                                entropy decoding, dequantize + IDCT on the device)
//   apply_rotate / _mirror / _crop / _resize  lib/src/editorhelper.cpp:210, 285, 351, 417  uhdr_hip_seam::effect
//   uhdr_encode / uhdr_decode       lib/src/ultrahdr_api.cpp:1200, 1918  uhdr_hip_seam::Scope (lazy context, like the
//                                                                GLES context at :1977-1989)
//
// Every seam function returns true when the device handled the call (*st is then the call's result, OK or the
// reference's own error) and false when the reference's CPU code must run: acceleration not enabled on this
// codec, or the combination is one the device path reports as UHDR_CODEC_UNSUPPORTED_FEATURE -- the
// contract of uhdr_enable_gpu_acceleration ("may have no effect", ultrahdr_api.h:839-849).  A device or
// runtime failure is never hidden: it is returned as the call's error.
#ifndef UHDR_HIP_SEAM_H
#define UHDR_HIP_SEAM_H

#include <memory>

#include "ultrahdr_api.h"
#include "ultrahdr/ultrahdrcommon.h"

namespace uhdr_hip_seam {

// RAII: makes `*slot` (the codec's lazily created uhdr_hip context) current on this thread for one
// uhdr_encode / uhdr_decode call.  enable == false: nothing happens, every seam returns false.
// lazy_downloads: the call allows decoded images to stay on the device (see lazy_downloads() below); uhdr_decode passes
// "no effects queued" -- apply_effects (ultrahdr_api.cpp:291-430) reads the gain-map image on the host.
class Scope {
 public:
  Scope(bool enable, void** slot, bool lazy_downloads = false);
  ~Scope();
  bool failed() const { return mFailed; }
  uhdr_error_info_t error() const { return mError; }

 private:
  void* mPrev;
  bool mFailed;
  uhdr_error_info_t mError;
};
// ~uhdr_codec_private
void release(void* ctxt);

// Lazy downloads of decodeJPEGR's decoded images (uhdr_hip_resident_lazy, include/uhdr_hip.h).
//   lazy_downloads(on)   jpegr.cpp:1478-1488: switched on in front of a JpegDecoderHelper::decompressImage whose result has no
//                        reader but applyGainMap and the copy below; the decode then leaves the helper's buffer unwritten.  Every
//                        seam entry point that hands a stage back to the reference's CPU code writes the buffers first.
//   defer_copy(src, dst) in place of copy_raw_image(&gainmap, gainmap_img) (jpegr.cpp:1490): true when the copy is now pending
//                        on the device (dst's cg / ct / range are set as copy_raw_image sets them), false when the caller
//                        copies on the host (the buffers are written by then).
//   materialize(ctxt)    uhdr_get_decoded_gainmap_image (ultrahdr_api.cpp:2032-2043): makes the pending copy.  false: a
//                        device error, there is no image to hand out.
//   forget(ctxt)         uhdr_reset_decoder: the destination is about to be freed.
void lazy_downloads(bool on);
bool defer_copy(uhdr_raw_image_t* src, uhdr_raw_image_t* dst);
bool materialize(void* ctxt);
void forget(void* ctxt);
// counters for tests / the demo app: how many stage calls ran on the device in this process
unsigned long calls_on_device();

bool apply_gainmap(uhdr_raw_image_t* sdr_intent, uhdr_raw_image_t* gainmap_img,
                   ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata, uhdr_color_transfer_t output_ct,
                   uhdr_img_fmt_t output_format, float max_display_boost, uhdr_raw_image_t* dest,
                   uhdr_error_info_t* st);
// scale_factor is UltraHdr::mMapDimensionScaleFactor: updated like setMapDimensionScaleFactor() does when the
// reference's tiny-image fallback (jpegr.cpp:690-706) triggers
bool generate_gainmap(uhdr_raw_image_t* sdr_intent, uhdr_raw_image_t* hdr_intent,
                      ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata,
                      std::unique_ptr<ultrahdr::uhdr_raw_image_ext_t>& gainmap_img, bool sdr_is_601,
                      bool use_luminance, int* scale_factor, bool multi_channel, float gamma,
                      uhdr_enc_preset_t preset, float min_content_boost, float max_content_boost,
                      float target_disp_peak_brightness, uhdr_error_info_t* st);
bool tone_map(uhdr_raw_image_t* hdr_intent, uhdr_raw_image_t* sdr_intent, uhdr_error_info_t* st);

// JpegR::encodeJPEGR API-1 (lib/src/jpegr.cpp:253-316) as ONE device sequence (uhdr_hip_encode_api1_scans, include/uhdr_hip.h):
// both raw intents go up once, generateGainMap (two passes) + compressGainMap + convertYuv + compressImage run without a sample
// or a coefficient coming back, and the two entropy-coded scans come down into complete JPEG files -- the bytes
// JpegEncoderHelper::compressImage (jpegencoderhelper.cpp:101-244) writes: SOI, JFIF APP0, the ICC profile as APP2, the gain
// map's COM marker, DQT, SOF0, DHT, SOS, data, EOI.  The reference then calls appendGainMap on them as it does on its own.
// false: not a combination for the fused chain (an RGB or 4:2:2 / 4:4:4 SDR intent, dimensions that are not multiples of 16,
// the one-pass preset, gamma != 1, a route option such as UHDR_HIP_SEAM_CPU_ENTROPY) -- nothing happened, encodeJPEGR goes on
// through the per-stage seams above.
struct Api1Files {
  std::unique_ptr<unsigned char[]> base_data, gainmap_data;  // (not vectors: 14 MB of capacity must not be zero-filled)
  size_t base_size = 0, gainmap_size = 0, base_capacity = 0, gainmap_capacity = 0;
  uhdr_compressed_image_t base() const {  // as JpegEncoderHelper::getCompressedImage (jpegencoderhelper.cpp:118-129)
    uhdr_compressed_image_t i;
    i.data = base_data.get(); i.data_sz = base_size; i.capacity = base_size;
    i.cg = UHDR_CG_UNSPECIFIED; i.ct = UHDR_CT_UNSPECIFIED; i.range = UHDR_CR_UNSPECIFIED;
    return i;
  }
  uhdr_compressed_image_t gainmap() const {
    uhdr_compressed_image_t i;
    i.data = gainmap_data.get(); i.data_sz = gainmap_size; i.capacity = gainmap_size;
    i.cg = UHDR_CG_UNSPECIFIED; i.ct = UHDR_CT_UNSPECIFIED; i.range = UHDR_CR_UNSPECIFIED;
    return i;
  }
};
bool encode_api1(uhdr_raw_image_t* hdr_intent, uhdr_raw_image_t* sdr_intent, int base_quality, int map_quality, int* scale_factor,
                 bool multi_channel, float gamma, uhdr_enc_preset_t preset, float min_content_boost, float max_content_boost,
                 float target_disp_peak_brightness, const void* base_icc, size_t base_icc_size, const void* map_icc,
                 size_t map_icc_size, const char* map_comment, ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata,
                 Api1Files* out, uhdr_error_info_t* st);
// JpegR::encodeJPEGR API-0 (jpegr.cpp:179-244) as ONE device sequence for RGBA1010102 / RGBA half-float HDR intents (round 6; BASELINE
// config 3): tone map + one-pass gain map + RGB -> YCbCr 4:4:4 fused, FDCTs, both scans Huffman-coded (uhdr_hip_encode_api0_scans).
// base_icc(cg): the caller's IccHelper::writeIccProfile(UHDR_CT_SRGB, cg) for the gamut the tone-mapped rendition gets -- asked for through
// the callback because that gamut is only known once the device has answered.  false: declined (P010 / YCbCr 4:4:4 intents, a scale factor
// other than 1, dimensions that are not multiples of 8, ...): the per-stage seams run as before.
struct IccBytes { const void* data; size_t size; };
bool encode_api0(uhdr_raw_image_t* hdr_intent, int base_quality, int map_quality, int* scale_factor, bool multi_channel, float gamma,
                 float min_content_boost, float max_content_boost, float target_disp_peak_brightness,
                 IccBytes (*base_icc)(void* user, uhdr_color_gamut_t cg), void* icc_user, const void* map_icc, size_t map_icc_size,
                 const char* map_comment, ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata, uhdr_color_gamut_t* sdr_cg, Api1Files* out,
                 uhdr_error_info_t* st);
bool convert_yuv(uhdr_raw_image_t* image, uhdr_color_gamut_t src_encoding, uhdr_color_gamut_t dst_encoding,
                 uhdr_error_info_t* st);
bool convert_raw_input_to_ycbcr(uhdr_raw_image_t* src, bool chroma_sampling_enabled,
                                std::unique_ptr<ultrahdr::uhdr_raw_image_ext_t>* dst);

// JPEG block stage.  fdct_planes: 8-bit component planes (already padded to whole blocks by the caller, which
// knows the reference's padding rules) -> quantized coefficient blocks in JBLOCK order, one contiguous array per
// component.  idct_planes: the inverse.  quant tables in natural order (cinfo.quant_tbl_ptrs[..]->quantval).
bool fdct_planes(int ncomp, const unsigned char* const planes[3], const unsigned int strides[3],
                 const unsigned int blocks_w[3], const unsigned int blocks_h[3], const unsigned short* const qtables[3],
                 short* const coefs[3], uhdr_error_info_t* st);
bool idct_planes(int ncomp, const short* const coefs[3], const unsigned int blocks_w[3], const unsigned int blocks_h[3],
                 const unsigned short* const qtables[3], unsigned char* const planes[3], const unsigned int strides[3],
                 uhdr_error_info_t* st);

// Whole baseline scan on the device (uhdr_hip_jpeg_decode_scan): hdr is a uhdr_hip_jpeg_header_t filled in from
// jpeg_decompress_struct after jpeg_read_header; data / bytes = what the source manager still holds.  false: the device
// path does not take this file (the caller goes on with jpeg_read_coefficients).
bool decode_scan(const void* hdr, const unsigned char* data, size_t bytes, int out_channels, int libjpeg_variant,
                 unsigned char* const planes[3], const unsigned int hstride[3], const unsigned int vstride[3],
                 uhdr_error_info_t* st);

// The mirror for the encoder (uhdr_hip_jpeg_encode_scan): samples -> entropy-coded data with restart markers, on the device.
// scan is a uhdr_hip_jpeg_scan_t, qtables a uint16_t[3][64].
bool encode_scan(const void* scan, const void* qtables, const unsigned char* const planes[3], const unsigned int strides[3],
                 int rgb_channels, unsigned char* out, size_t cap, size_t* bytes, uhdr_error_info_t* st);

// the effects chain (lib/src/editorhelper.cpp:210-520): kind 0 rotate (p0 = degrees), 1 mirror (p0 = direction), 2 crop
// (p0 = left, p1 = top), 3 resize; dst_w x dst_h = size of the result.  *dst receives a freshly allocated image exactly
// as the reference allocates it (strides aligned to 64).
bool effect(int kind, int p0, int p1, int dst_w, int dst_h, uhdr_raw_image_t* src,
            std::unique_ptr<ultrahdr::uhdr_raw_image_ext_t>* dst);
// libjpeg's colour conversions around a 3-channel gain map (jccolor.c rgb_ycc_convert / jdcolor.c ycc_rgb_convert)
bool jpeg_rgb_to_ycc(const uhdr_raw_image_t* rgb, uhdr_raw_image_t* ycc, uhdr_error_info_t* st);
bool jpeg_ycc_to_rgb(const uhdr_raw_image_t* ycc, int libjpeg_variant, uhdr_raw_image_t* rgb, uhdr_error_info_t* st);
// is a device context current on this thread (i.e. are we inside an accelerated uhdr_encode / uhdr_decode)?
bool enabled();
// The reference's CPU code is about to run in place of a device stage: device-resident copies of host buffers (kept between the
// stages of one accelerated call) are dropped, because that code may rewrite those buffers.  Called by every seam entry point
// that returns false.
void drop_resident();

}  // namespace uhdr_hip_seam

#endif  // UHDR_HIP_SEAM_H
