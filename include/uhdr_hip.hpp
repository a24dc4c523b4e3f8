// C++ host-side mirror of ultrahdr::UltraHdr's stage operators over the C ABI of uhdr_hip.h.
//
// The reference is C++ (lib/include/ultrahdr/ultrahdrcommon.h:448-546 declares the class whose four
// methods are the hot path).  This header gives a C++ caller the same class shape -- same
// constructor knobs in the same order, same method names, argument order and error convention
// (uhdr_error_info_t by value, nothing throws) -- with the MI355X library behind it, so code written
// against ultrahdr::UltraHdr moves over by changing the type name.  Header only; link -luhdr_hip.
//
// Differences, all forced by the boundary:
//   * the first constructor argument is the device ordinal (the reference passes its GLES context there);
//   * the metadata type is the C struct uhdr_gainmap_metadata_t (uhdr_gainmap_metadata_ext_t only adds
//     the std::string version, always "1.0", ultrahdrcommon.h:446);
//   * generateGainMap returns the map in a uhdr_hip::raw_image_ext, which allocates exactly as the
//     reference's uhdr_raw_image_ext_t does (one block, stride aligned to 64 pixels,
//     lib/src/ultrahdr_api.cpp:55-117).
#pragma once
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>

#include "uhdr_hip.h"

namespace uhdr_hip {

// Owning raw image for the packed / single-plane formats a gain map uses.
struct raw_image_ext : uhdr_raw_image_t {
  raw_image_ext(uhdr_img_fmt_t fmt_, uhdr_color_gamut_t cg_, uhdr_color_transfer_t ct_, uhdr_color_range_t range_,
                unsigned w_, unsigned h_, unsigned align_stride_to) {
    std::memset(static_cast<uhdr_raw_image_t*>(this), 0, sizeof(uhdr_raw_image_t));
    fmt = fmt_; cg = cg_; ct = ct_; range = range_; w = w_; h = h_;
    const unsigned aligned_w = ((w_ + align_stride_to - 1) / align_stride_to) * align_stride_to;
    size_t bpp = 1;
    if (fmt_ == UHDR_IMG_FMT_24bppRGB888) bpp = 3;
    else if (fmt_ == UHDR_IMG_FMT_32bppRGBA8888 || fmt_ == UHDR_IMG_FMT_32bppRGBA1010102) bpp = 4;
    else if (fmt_ == UHDR_IMG_FMT_64bppRGBAHalfFloat) bpp = 8;
    block_.reset(static_cast<uint8_t*>(std::calloc((size_t)aligned_w * h_ * bpp + 64, 1)));
    planes[0] = block_.get();
    stride[0] = aligned_w;
  }

 private:
  struct free_deleter {
    void operator()(uint8_t* p) const { std::free(p); }
  };
  std::unique_ptr<uint8_t, free_deleter> block_;
};

class UltraHdr {
 public:
  // ultrahdrcommon.h:450-457 (Android defaults: scale 4, quality 85, single channel, REALTIME)
  explicit UltraHdr(int device = -1, int mapDimensionScaleFactor = 4, int mapCompressQuality = 85,
                    bool useMultiChannelGainMap = false, float gamma = 1.0f,
                    uhdr_enc_preset_t preset = UHDR_USAGE_REALTIME, float minContentBoost = FLT_MIN,
                    float maxContentBoost = FLT_MAX, float targetDispPeakBrightness = -1.0f)
      : mMapDimensionScaleFactor(mapDimensionScaleFactor),
        mMapCompressQuality(mapCompressQuality),
        mUseMultiChannelGainMap(useMultiChannelGainMap),
        mGamma(gamma),
        mEncPreset(preset),
        mMinContentBoost(minContentBoost),
        mMaxContentBoost(maxContentBoost),
        mTargetDispPeakBrightness(targetDispPeakBrightness) {
    mCtx = uhdr_hip_create(device, &mCreateStatus);
  }
  ~UltraHdr() {
    if (mCtx) uhdr_hip_destroy(mCtx);
  }
  UltraHdr(const UltraHdr&) = delete;
  UltraHdr& operator=(const UltraHdr&) = delete;

  // UHDR_CODEC_OK, or why no MI355X-class device could be opened (there is no CPU fallback)
  uhdr_error_info_t status() const { return mCreateStatus; }
  uhdr_hip_ctx_t* context() const { return mCtx; }

  uhdr_error_info_t toneMap(uhdr_raw_image_t* hdr_intent, uhdr_raw_image_t* sdr_intent) {
    if (!mCtx) return mCreateStatus;
    return uhdr_hip_tone_map(mCtx, hdr_intent, sdr_intent);
  }

  uhdr_error_info_t generateGainMap(uhdr_raw_image_t* sdr_intent, uhdr_raw_image_t* hdr_intent,
                                    uhdr_gainmap_metadata_t* gainmap_metadata, std::unique_ptr<raw_image_ext>& gainmap_img,
                                    bool sdr_is_601 = false, bool use_luminance = true) {
    if (!mCtx) return mCreateStatus;
    if (!sdr_intent || !hdr_intent || !gainmap_metadata) return bad("received nullptr argument");
    uhdr_hip_encode_cfg_t cfg{mMapDimensionScaleFactor, mUseMultiChannelGainMap ? 1 : 0, mGamma, (int)mEncPreset,
                              mMinContentBoost, mMaxContentBoost, mTargetDispPeakBrightness, sdr_is_601 ? 1 : 0,
                              use_luminance ? 1 : 0};
    // map geometry: jpegr.cpp:693-716
    unsigned scale = mMapDimensionScaleFactor < 1 ? 1u : (unsigned)mMapDimensionScaleFactor;
    unsigned mw = sdr_intent->w / scale, mh = sdr_intent->h / scale;
    if (mw == 0 || mh == 0) {
      unsigned s = sdr_intent->w < sdr_intent->h ? sdr_intent->w : sdr_intent->h;
      s = s >= 8 ? s / 8 : 1;
      mw = sdr_intent->w / s;
      mh = sdr_intent->h / s;
    }
    gainmap_img = std::make_unique<raw_image_ext>(mUseMultiChannelGainMap ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_8bppYCbCr400,
                                                  hdr_intent->cg, hdr_intent->ct, hdr_intent->range, mw, mh, 64);
    return uhdr_hip_generate_gainmap(mCtx, sdr_intent, hdr_intent, &cfg, gainmap_metadata, gainmap_img.get());
  }

  uhdr_error_info_t applyGainMap(uhdr_raw_image_t* sdr_intent, uhdr_raw_image_t* gainmap_img,
                                 uhdr_gainmap_metadata_t* gainmap_metadata, uhdr_color_transfer_t output_ct,
                                 uhdr_img_fmt_t output_format, float max_display_boost, uhdr_raw_image_t* dest) {
    if (!mCtx) return mCreateStatus;
    return uhdr_hip_apply_gainmap(mCtx, sdr_intent, gainmap_img, gainmap_metadata, output_ct, output_format, max_display_boost, dest);
  }

  uhdr_error_info_t convertYuv(uhdr_raw_image_t* image, uhdr_color_gamut_t src_encoding, uhdr_color_gamut_t dst_encoding) {
    if (!mCtx) return mCreateStatus;
    return uhdr_hip_convert_yuv(mCtx, image, src_encoding, dst_encoding);
  }

 protected:
  void setMapDimensionScaleFactor(int v) { mMapDimensionScaleFactor = v; }
  int getMapDimensionScaleFactor() const { return mMapDimensionScaleFactor; }
  void setMapCompressQuality(int v) { mMapCompressQuality = v; }
  int getMapCompressQuality() const { return mMapCompressQuality; }
  void setGainMapGamma(float v) { mGamma = v; }
  float getGainMapGamma() const { return mGamma; }
  void setUseMultiChannelGainMap(bool v) { mUseMultiChannelGainMap = v; }
  bool isUsingMultiChannelGainMap() const { return mUseMultiChannelGainMap; }
  void setGainMapMinMaxContentBoost(float mn, float mx) { mMinContentBoost = mn; mMaxContentBoost = mx; }

 private:
  static uhdr_error_info_t bad(const char* msg) {
    uhdr_error_info_t st;
    std::memset(&st, 0, sizeof st);
    st.error_code = UHDR_CODEC_INVALID_PARAM;
    st.has_detail = 1;
    std::strncpy(st.detail, msg, sizeof st.detail - 1);
    return st;
  }
  uhdr_hip_ctx_t* mCtx = nullptr;
  uhdr_error_info_t mCreateStatus{};
  int mMapDimensionScaleFactor;
  int mMapCompressQuality;
  bool mUseMultiChannelGainMap;
  float mGamma;
  uhdr_enc_preset_t mEncPreset;
  float mMinContentBoost, mMaxContentBoost, mTargetDispPeakBrightness;
};

}  // namespace uhdr_hip
