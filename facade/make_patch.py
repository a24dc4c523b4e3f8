#!/usr/bin/env python
"""Regenerates facade/reference_hip_seam.patch: applies the seam insertions below to scratch copies of the
reference files (under /tmp, never inside this repository) and diffs them against the originals with one line of
context.  The patch is what a libultrahdr maintainer would review; facade/Makefile applies it out of tree."""
import difflib
import os
import sys

REF = os.environ.get("REF", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_hip_seam.patch")


def insert_before(text, anchor, new, nth=0):
    idx = -1
    for _ in range(nth + 1):
        idx = text.index(anchor, idx + 1)
    return text[:idx] + new + text[idx:]


def insert_after(text, anchor, new, nth=0):
    idx = -1
    for _ in range(nth + 1):
        idx = text.index(anchor, idx + 1)
    idx += len(anchor)
    return text[:idx] + new + text[idx:]


EDITS = {}

# ---- ultrahdrcommon.h: per-codec switch + lazily created context (next to the GLES members) -------------------
def common_h(t):
    # uhdr_memory_block: zero-filled as before, but by calloc (facade/uhdr_zero_pages.h) -- pages the kernel hands out zeroed are
    # not written a second time (a 4K encode's 49.8 MB output buffer, of which 4 MB are used)
    t = t.replace("  std::unique_ptr<uint8_t[]> m_buffer; /**< data */\n",
                  "#ifdef UHDR_ENABLE_HIP\n  std::unique_ptr<uint8_t[], uhdr_zero_pages::block_free> m_buffer; /**< data */\n#else\n"
                  "  std::unique_ptr<uint8_t[]> m_buffer; /**< data */\n#endif\n", 1)
    t = insert_after(t, '#include "ultrahdr_api.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_zero_pages.h"\n#endif\n')
    return insert_before(t, "  bool m_sailed;\n\n  virtual ~uhdr_codec_private();",
                         "#ifdef UHDR_ENABLE_HIP\n  bool m_enable_hip = false;         // uhdr_enable_gpu_acceleration()\n"
                         "  void* m_uhdr_hip_ctxt = nullptr;   // uhdr_hip context, created by the first accelerated call\n#endif\n")
EDITS["lib/include/ultrahdr/ultrahdrcommon.h"] = common_h

# ---- jpegdecoderhelper.h: the decoded-image buffer, zero-filled without touching fresh pages ---------------------
def dec_h(t):
    t = t.replace("  std::vector<JOCTET> mResultBuffer;       // buffer to store decoded data\n",
                  "#ifdef UHDR_ENABLE_HIP\n  uhdr_zero_pages::bytes mResultBuffer;    // buffer to store decoded data (zero-filled by calloc)\n#else\n"
                  "  std::vector<JOCTET> mResultBuffer;       // buffer to store decoded data\n#endif\n", 1)
    assert "uhdr_zero_pages::bytes" in t
    return insert_after(t, '#include "ultrahdr_api.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_zero_pages.h"\n#endif\n')
EDITS["lib/include/ultrahdr/jpegdecoderhelper.h"] = dec_h

# ---- ultrahdr_api.cpp -------------------------------------------------------------------------------------------
def api_cpp(t):
    t = t.replace("  m_buffer = std::make_unique<uint8_t[]>(capacity);\n",
                  "#ifdef UHDR_ENABLE_HIP\n  m_buffer.reset(uhdr_zero_pages::zeroed(capacity));  // the same zero-filled block, fresh pages left alone\n#else\n"
                  "  m_buffer = std::make_unique<uint8_t[]>(capacity);\n#endif\n", 1)
    t = insert_after(t, '#include "ultrahdr/jpegr.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_hip_seam.h"\n#endif\n')
    t = insert_after(t, "uhdr_codec_private::~uhdr_codec_private() {\n",
                     "#ifdef UHDR_ENABLE_HIP\n  uhdr_hip_seam::release(m_uhdr_hip_ctxt);\n#endif\n")
    scope = ("#ifdef UHDR_ENABLE_HIP\n  uhdr_hip_seam::Scope hip_scope(handle->m_enable_hip, &handle->m_uhdr_hip_ctxt);\n"
             "  if (hip_scope.failed()) {\n    status = hip_scope.error();\n    return status;\n  }\n#endif\n")
    # uhdr_encode: the generated gain map may stay on the device -- every encodeJPEGR variant hands it straight to compressGainMap
    t = insert_after(t, "  uhdr_error_info_t& status = handle->m_encode_call_status;\n",
                     scope.replace("&handle->m_uhdr_hip_ctxt);", "&handle->m_uhdr_hip_ctxt, /* lazy_downloads */ true);"))
    # uhdr_decode: decoded images may stay on the device unless effects are queued (apply_effects reads the gain-map image on the host)
    t = insert_after(t, "  status = uhdr_dec_probe(dec);\n  if (status.error_code != UHDR_CODEC_OK) return status;\n\n  handle->m_sailed = true;\n",
                     scope.replace("&handle->m_uhdr_hip_ctxt);", "&handle->m_uhdr_hip_ctxt,\n                                  /* lazy_downloads */ handle->m_effects.empty());"))
    t = insert_before(t, "  return handle->m_gainmap_img_buffer.get();\n",
                      "#ifdef UHDR_ENABLE_HIP\n  if (!uhdr_hip_seam::materialize(handle->m_uhdr_hip_ctxt)) return nullptr;\n#endif\n")
    t = insert_after(t, "#ifdef UHDR_ENABLE_GLES\n  codec->m_enable_gles = enable;\n#endif\n",
                     "#ifdef UHDR_ENABLE_HIP\n  codec->m_enable_hip = enable;\n#endif\n")
    for nth in (0, 1):  # uhdr_reset_encoder, uhdr_reset_decoder
        t = insert_before(t, "    handle->m_sailed = false;\n",
                          "#ifdef UHDR_ENABLE_HIP\n    handle->m_enable_hip = false;\n" +
                          ("    uhdr_hip_seam::forget(handle->m_uhdr_hip_ctxt);  // a deferred gain-map image copy: its destination goes away below\n" if nth == 1 else "") +
                          "#endif\n", nth)
    return t
EDITS["lib/src/ultrahdr_api.cpp"] = api_cpp

# ---- jpegr.cpp: the four stage operators ---------------------------------------------------------------------
def jpegr_cpp(t):
    t = insert_after(t, '#include "ultrahdr/jpegr.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_hip_seam.h"\n#endif\n')
    t = insert_after(t, "uhdr_error_info_t UltraHdr::convertYuv(uhdr_raw_image_t* image, uhdr_color_gamut_t src_encoding,\n"
                        "                                       uhdr_color_gamut_t dst_encoding) {\n",
                     "#ifdef UHDR_ENABLE_HIP\n  {\n    uhdr_error_info_t hip_status;\n"
                     "    if (uhdr_hip_seam::convert_yuv(image, src_encoding, dst_encoding, &hip_status)) return hip_status;\n  }\n#endif\n")
    t = insert_after(t, "                                            bool sdr_is_601, bool use_luminance) {\n  uhdr_error_info_t status = g_no_error;\n",
                     "#ifdef UHDR_ENABLE_HIP\n"
                     "  if (uhdr_hip_seam::generate_gainmap(sdr_intent, hdr_intent, gainmap_metadata, gainmap_img, sdr_is_601,\n"
                     "                                      use_luminance, &mMapDimensionScaleFactor, mUseMultiChannelGainMap, mGamma,\n"
                     "                                      mEncPreset, mMinContentBoost, mMaxContentBoost,\n"
                     "                                      mTargetDispPeakBrightness, &status))\n    return status;\n"
                     "  status = g_no_error;\n#endif\n")
    # encodeJPEGR API-1: the whole sample -> bytes part in one device sequence (the stage seams take what it declines)
    t = insert_after(t, "/* Encode API-1 */\nuhdr_error_info_t JpegR::encodeJPEGR(uhdr_raw_image_t* hdr_intent, uhdr_raw_image_t* sdr_intent,\n"
                        "                                     uhdr_compressed_image_t* dest, int quality,\n"
                        "                                     uhdr_mem_block_t* exif) {\n"
                        "  // generate gain map\n  uhdr_gainmap_metadata_ext_t metadata(kJpegrVersion);\n",
                     "#ifdef UHDR_ENABLE_HIP\n  if (uhdr_hip_seam::enabled()) {\n"
                     "    std::shared_ptr<DataStruct> hip_icc_base = IccHelper::writeIccProfile(UHDR_CT_SRGB, sdr_intent->cg);\n"
                     "    std::shared_ptr<DataStruct> hip_icc_map =  // compressGainMap's choice (jpegr.cpp:520-528)\n"
                     "        kWriteXmpMetadata ? nullptr : IccHelper::writeIccProfile(hdr_intent->ct, hdr_intent->cg);\n"
                     "    char hip_comment[255];  // JpegEncoderHelper::encode's COM marker of a gain-map image\n"
                     "    snprintf(hip_comment, sizeof hip_comment,\n"
                     "             \"Source: google libuhdr v%s, Coder: libjpeg v%d, Attrib: GainMap Image\",\n"
                     "             UHDR_LIB_VERSION_STR, JPEG_LIB_VERSION);\n"
                     "    uhdr_hip_seam::Api1Files hip_files;\n    uhdr_error_info_t hip_status;\n"
                     "    if (uhdr_hip_seam::encode_api1(hdr_intent, sdr_intent, quality, mMapCompressQuality, &mMapDimensionScaleFactor,\n"
                     "                                   mUseMultiChannelGainMap, mGamma, mEncPreset, mMinContentBoost,\n"
                     "                                   mMaxContentBoost, mTargetDispPeakBrightness,\n"
                     "                                   hip_icc_base ? hip_icc_base->getData() : nullptr,  // (nullptr for a gamut IccHelper does not know:\n"
                     "                                   hip_icc_base ? hip_icc_base->getLength() : 0,      //  the device call then words the reference's error)\n"
                     "                                   hip_icc_map ? hip_icc_map->getData() : nullptr,\n"
                     "                                   hip_icc_map ? hip_icc_map->getLength() : 0, hip_comment, &metadata, &hip_files,\n"
                     "                                   &hip_status)) {\n"
                     "      if (hip_status.error_code != UHDR_CODEC_OK) return hip_status;\n"
                     "      uhdr_compressed_image_t gainmap_compressed = hip_files.gainmap();\n"
                     "      uhdr_compressed_image_t sdr_intent_compressed = hip_files.base();\n"
                     "      sdr_intent_compressed.cg = sdr_intent->cg;\n"
                     "      UHDR_ERR_CHECK(appendGainMap(&sdr_intent_compressed, &gainmap_compressed, exif, /* icc */ nullptr,\n"
                     "                                   /* icc size */ 0, &metadata, dest));\n"
                     "      return g_no_error;\n    }\n  }\n#endif\n")
    # encodeJPEGR API-0 (round 6): RGBA1010102 / RGBA half-float intents as one device sequence (BASELINE config 3); P010 keeps the stage seams
    t = insert_before(t, "  std::unique_ptr<uhdr_raw_image_ext_t> sdr_intent = std::make_unique<uhdr_raw_image_ext_t>(\n"
                         "      sdr_intent_fmt, UHDR_CG_UNSPECIFIED, UHDR_CT_UNSPECIFIED, UHDR_CR_UNSPECIFIED, hdr_intent->w,\n"
                         "      hdr_intent->h, 64);\n\n  // tone map\n",
                      "#ifdef UHDR_ENABLE_HIP\n  if (uhdr_hip_seam::enabled() && sdr_intent_fmt == UHDR_IMG_FMT_32bppRGBA8888) {\n"
                      "    uhdr_gainmap_metadata_ext_t hip_metadata(kJpegrVersion);\n"
                      "    std::shared_ptr<DataStruct> hip_icc_map =  // compressGainMap's choice (jpegr.cpp:520-528)\n"
                      "        kWriteXmpMetadata ? nullptr : IccHelper::writeIccProfile(hdr_intent->ct, hdr_intent->cg);\n"
                      "    char hip_comment[255];  // JpegEncoderHelper::encode's COM marker of a gain-map image\n"
                      "    snprintf(hip_comment, sizeof hip_comment,\n"
                      "             \"Source: google libuhdr v%s, Coder: libjpeg v%d, Attrib: GainMap Image\",\n"
                      "             UHDR_LIB_VERSION_STR, JPEG_LIB_VERSION);\n"
                      "    std::shared_ptr<DataStruct> hip_icc_base;  // written once the device has said which gamut the rendition has\n"
                      "    auto hip_icc_cb = [](void* user, uhdr_color_gamut_t cg) -> uhdr_hip_seam::IccBytes {\n"
                      "      auto* slot = static_cast<std::shared_ptr<DataStruct>*>(user);\n"
                      "      *slot = IccHelper::writeIccProfile(UHDR_CT_SRGB, cg);\n"
                      "      return *slot ? uhdr_hip_seam::IccBytes{(*slot)->getData(), (size_t)(*slot)->getLength()}\n"
                      "                   : uhdr_hip_seam::IccBytes{nullptr, 0};\n    };\n"
                      "    uhdr_hip_seam::Api1Files hip_files;\n    uhdr_error_info_t hip_status;\n"
                      "    uhdr_color_gamut_t hip_sdr_cg = UHDR_CG_UNSPECIFIED;\n"
                      "    if (uhdr_hip_seam::encode_api0(hdr_intent, quality, mMapCompressQuality, &mMapDimensionScaleFactor,\n"
                      "                                   mUseMultiChannelGainMap, mGamma, mMinContentBoost, mMaxContentBoost,\n"
                      "                                   mTargetDispPeakBrightness, hip_icc_cb, &hip_icc_base,\n"
                      "                                   hip_icc_map ? hip_icc_map->getData() : nullptr,\n"
                      "                                   hip_icc_map ? hip_icc_map->getLength() : 0, hip_comment, &hip_metadata,\n"
                      "                                   &hip_sdr_cg, &hip_files, &hip_status)) {\n"
                      "      if (hip_status.error_code != UHDR_CODEC_OK) return hip_status;\n"
                      "      mEncPreset = UHDR_USAGE_REALTIME;  // as below\n"
                      "      uhdr_compressed_image_t gainmap_compressed = hip_files.gainmap();\n"
                      "      uhdr_compressed_image_t sdr_intent_compressed = hip_files.base();\n"
                      "      sdr_intent_compressed.cg = hip_sdr_cg;\n"
                      "      UHDR_ERR_CHECK(appendGainMap(&sdr_intent_compressed, &gainmap_compressed, exif, /* icc */ nullptr,\n"
                      "                                   /* icc size */ 0, &hip_metadata, dest));\n"
                      "      return g_no_error;\n    }\n  }\n#endif\n")
    t = insert_before(t, "#ifdef UHDR_ENABLE_GLES\n  if (mUhdrGLESCtxt != nullptr) {\n",
                      "#ifdef UHDR_ENABLE_HIP\n  {\n    uhdr_error_info_t hip_status;\n"
                      "    if (uhdr_hip_seam::apply_gainmap(sdr_intent, gainmap_img, gainmap_metadata, output_ct, output_format,\n"
                      "                                     max_display_boost, dest, &hip_status))\n      return hip_status;\n  }\n#endif\n")
    # decodeJPEGR: the two decoded images stay on the device; the copy into the caller's gain-map image is deferred
    t = insert_before(t, "  UHDR_ERR_CHECK(jpeg_dec_obj_sdr.decompressImage(\n      primary_jpeg_image.data, primary_jpeg_image.data_sz,\n"
                         "      (output_ct == UHDR_CT_SRGB) ? DECODE_TO_RGB_CS : DECODE_TO_YCBCR_CS));\n\n  JpegDecoderHelper jpeg_dec_obj_gm;\n",
                      "#ifdef UHDR_ENABLE_HIP\n  uhdr_hip_seam::lazy_downloads(output_ct != UHDR_CT_SRGB);  // the planes' only reader is applyGainMap below\n#endif\n")
    t = insert_before(t, "    UHDR_ERR_CHECK(jpeg_dec_obj_gm.decompressImage(gainmap_jpeg_image.data,\n"
                         "                                                   gainmap_jpeg_image.data_sz, DECODE_STREAM));\n"
                         "    gainmap = jpeg_dec_obj_gm.getDecompressedImage();\n    if (gainmap_img != nullptr) {\n",
                      "#ifdef UHDR_ENABLE_HIP\n    uhdr_hip_seam::lazy_downloads(true);  // readers: the copy below and applyGainMap\n#endif\n")
    t = insert_before(t, "      UHDR_ERR_CHECK(copy_raw_image(&gainmap, gainmap_img));\n    }\n    gainmap.cg =\n",
                      "#ifdef UHDR_ENABLE_HIP\n      uhdr_hip_seam::lazy_downloads(false);\n      if (!uhdr_hip_seam::defer_copy(&gainmap, gainmap_img))\n#endif\n")
    t = insert_after(t, "uhdr_error_info_t UltraHdr::toneMap(uhdr_raw_image_t* hdr_intent, uhdr_raw_image_t* sdr_intent) {\n",
                     "#ifdef UHDR_ENABLE_HIP\n  {\n    uhdr_error_info_t hip_status;\n"
                     "    if (uhdr_hip_seam::tone_map(hdr_intent, sdr_intent, &hip_status)) return hip_status;\n  }\n#endif\n")
    return t
EDITS["lib/src/jpegr.cpp"] = jpegr_cpp

# ---- gainmapmath.cpp: convert_raw_input_to_ycbcr --------------------------------------------------------------
def gmm_cpp(t):
    t = insert_after(t, '#include "ultrahdr/gainmapmath.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_hip_seam.h"\n#endif\n')
    t = insert_after(t, "std::unique_ptr<uhdr_raw_image_ext_t> convert_raw_input_to_ycbcr(uhdr_raw_image_t* src,\n"
                        "                                                                 bool chroma_sampling_enabled) {\n"
                        "  std::unique_ptr<uhdr_raw_image_ext_t> dst = nullptr;\n",
                     "#ifdef UHDR_ENABLE_HIP\n  if (uhdr_hip_seam::convert_raw_input_to_ycbcr(src, chroma_sampling_enabled, &dst)) return dst;\n#endif\n")
    return t
EDITS["lib/src/gainmapmath.cpp"] = gmm_cpp

# ---- the JPEG block stage (FDCT / IDCT on the device, entropy coding stays in libjpeg) -------------------------
def jenc_cpp(t):
    t = insert_after(t, '#include "ultrahdr/jpegencoderhelper.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_hip_jpeg_seam.h"\n#endif\n')
    t = insert_before(t, "    // start compress\n    jpeg_start_compress(&cinfo, TRUE);\n",
                      "#ifdef UHDR_ENABLE_HIP\n    {\n      char hip_comment[255];\n"
                      "      snprintf(hip_comment, sizeof hip_comment,\n"
                      "               \"Source: google libuhdr v%s, Coder: libjpeg v%d, Attrib: GainMap Image\",\n"
                      "               UHDR_LIB_VERSION_STR, JPEG_LIB_VERSION);\n"
                      "      if (uhdr_hip_seam::jpeg_compress_on_device(&cinfo, planes, strides, format, iccBuffer, iccSize,\n"
                      "                                                 isGainMapImg ? hip_comment : nullptr, &status)) {\n"
                      "        jpeg_destroy_compress(&cinfo);\n        return status;\n      }\n    }\n#endif\n")
    return t
EDITS["lib/src/jpegencoderhelper.cpp"] = jenc_cpp

def jdec_cpp(t):
    t = insert_after(t, '#include "ultrahdr/jpegdecoderhelper.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_hip_jpeg_seam.h"\n#endif\n')
    t = insert_before(t, "    cinfo.dct_method = JDCT_ISLOW;\n    jpeg_start_decompress(&cinfo);\n",
                      "#ifdef UHDR_ENABLE_HIP\n"
                      "    if (uhdr_hip_seam::jpeg_decompress_on_device(&cinfo, DECODE_TO_RGB_CS == mode, mResultBuffer.data(),\n"
                      "                                                 mPlaneHStride, mPlaneVStride, &mOutFormat, &status)) {\n"
                      "      jpeg_destroy_decompress(&cinfo);\n      return status;\n    }\n#endif\n")
    return t
EDITS["lib/src/jpegdecoderhelper.cpp"] = jdec_cpp


# ---- editorhelper.cpp: the four effects ---------------------------------------------------------------------------------
def editor_cpp(t):
    t = insert_after(t, '#include "ultrahdr/editorhelper.h"\n', '#ifdef UHDR_ENABLE_HIP\n#include "uhdr_hip_seam.h"\n#endif\n')
    def seam(call):
        return ("#ifdef UHDR_ENABLE_HIP\n  {\n    std::unique_ptr<uhdr_raw_image_ext_t> hip_dst;\n"
                "    if (" + call + ") return hip_dst;\n  }\n#endif\n")
    t = insert_after(t, "std::unique_ptr<uhdr_raw_image_ext_t> apply_rotate(ultrahdr::uhdr_rotate_effect_t* desc,\n"
                        "                                                   uhdr_raw_image_t* src,\n"
                        "                                                   [[maybe_unused]] void* gl_ctxt,\n"
                        "                                                   [[maybe_unused]] void* texture) {\n",
                     seam("(desc->m_degree == 90 || desc->m_degree == 180 || desc->m_degree == 270) &&\n"
                          "        uhdr_hip_seam::effect(0, desc->m_degree, 0, desc->m_degree == 180 ? src->w : src->h,\n"
                          "                              desc->m_degree == 180 ? src->h : src->w, src, &hip_dst)"))
    t = insert_after(t, "std::unique_ptr<uhdr_raw_image_ext_t> apply_mirror(ultrahdr::uhdr_mirror_effect_t* desc,\n"
                        "                                                   uhdr_raw_image_t* src,\n"
                        "                                                   [[maybe_unused]] void* gl_ctxt,\n"
                        "                                                   [[maybe_unused]] void* texture) {\n",
                     seam("uhdr_hip_seam::effect(1, (int)desc->m_direction, 0, src->w, src->h, src, &hip_dst)"))
    t = insert_after(t, "                                                 int ht, [[maybe_unused]] void* gl_ctxt,\n"
                        "                                                 [[maybe_unused]] void* texture) {\n",
                     seam("uhdr_hip_seam::effect(2, left, top, wd, ht, src, &hip_dst)"))
    t = insert_after(t, "                                                   uhdr_raw_image_t* src, int dst_w, int dst_h,\n"
                        "                                                   [[maybe_unused]] void* gl_ctxt,\n"
                        "                                                   [[maybe_unused]] void* texture) {\n",
                     seam("uhdr_hip_seam::effect(3, 0, 0, dst_w, dst_h, src, &hip_dst)"))
    return t
EDITS["lib/src/editorhelper.cpp"] = editor_cpp


def main():
    chunks = []
    for rel, fn in EDITS.items():
        with open(os.path.join(REF, rel)) as f:
            old = f.read()
        new = fn(old)
        d = difflib.unified_diff(old.splitlines(True), new.splitlines(True), "a/" + rel, "b/" + rel, n=1)
        chunks.append("".join(d))
    with open(OUT, "w") as f:
        f.write("".join(chunks))
    print("wrote", OUT, sum(c.count("\n") for c in chunks), "lines")


if __name__ == "__main__":
    sys.exit(main())
