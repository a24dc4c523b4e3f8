// TEST PROGRAM (tests/test_cpp_mirror.py builds and runs it): a C++ caller written against the class
// shape of ultrahdr::UltraHdr, using include/uhdr_hip.hpp, checked against the C oracle.
//   mirror_check            -> runs toneMap / generateGainMap / applyGainMap / convertYuv on an MI355X and
//                              compares every output plane with oracle/libuhdr_oracle.so (exit 0 = all equal)
//   mirror_check --no-gpu   -> only checks that construction fails loudly when no device is usable
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "uhdr_hip.hpp"
#include "uhdr_oracle.h"

static_assert(sizeof(uo_image_t) == sizeof(uhdr_raw_image_t), "oracle image struct mirrors the reference's");
static_assert(sizeof(uo_metadata_t) == sizeof(uhdr_gainmap_metadata_t), "oracle metadata struct mirrors the reference's");

struct Planes {
  std::vector<uint8_t> mem;
  uhdr_raw_image_t img;
};

static Planes make(uhdr_img_fmt_t fmt, unsigned w, unsigned h, uhdr_color_gamut_t cg, uhdr_color_transfer_t ct, uhdr_color_range_t rg) {
  Planes p;
  std::memset(&p.img, 0, sizeof p.img);
  p.img.fmt = fmt; p.img.cg = cg; p.img.ct = ct; p.img.range = rg; p.img.w = w; p.img.h = h;
  const unsigned aw = (w + 63) / 64 * 64;
  size_t sz[3] = {0, 0, 0};
  unsigned st[3] = {aw, 0, 0};
  switch (fmt) {
    case UHDR_IMG_FMT_24bppYCbCrP010: sz[0] = (size_t)aw * h * 2; sz[1] = (size_t)aw * (h / 2) * 2; st[1] = aw; break;
    case UHDR_IMG_FMT_12bppYCbCr420: sz[0] = (size_t)aw * h; sz[1] = sz[2] = (size_t)(aw / 2) * (h / 2); st[1] = st[2] = aw / 2; break;
    case UHDR_IMG_FMT_64bppRGBAHalfFloat: sz[0] = (size_t)aw * h * 8; break;
    case UHDR_IMG_FMT_32bppRGBA1010102: sz[0] = (size_t)aw * h * 4; break;
    default: sz[0] = (size_t)aw * h; break;
  }
  p.mem.assign(sz[0] + sz[1] + sz[2] + 64, 0);
  size_t off = 0;
  for (int i = 0; i < 3; i++) {
    if (!sz[i]) continue;
    p.img.planes[i] = p.mem.data() + off;
    p.img.stride[i] = st[i];
    off += sz[i];
  }
  return p;
}

static uint32_t lcg(uint32_t& s) { return s = s * 1664525u + 1013904223u; }

int main(int argc, char** argv) {
  uhdr_hip::UltraHdr hip(/*device*/ -1, /*scale*/ 1, 95, /*multichannel*/ true, 1.0f, UHDR_USAGE_BEST_QUALITY);
  if (argc > 1 && !std::strcmp(argv[1], "--no-gpu")) {
    const uhdr_error_info_t st = hip.status();
    if (hip.context() == nullptr && st.error_code != UHDR_CODEC_OK && st.has_detail) {
      std::printf("no device: construction reported '%s' (code %d) -- no CPU fallback, as intended\n", st.detail, (int)st.error_code);
      uhdr_raw_image_t dummy;
      std::memset(&dummy, 0, sizeof dummy);
      return hip.toneMap(&dummy, &dummy).error_code != UHDR_CODEC_OK ? 0 : 2;
    }
    std::printf("a device is present; nothing to check in --no-gpu mode\n");
    return 0;
  }
  if (!hip.context()) { std::printf("FAIL: %s\n", hip.status().detail); return 1; }

  const unsigned w = 256, h = 128;
  uint32_t seed = 12345;
  // synthetic pair: smooth luma ramp + noise, mid-grey chroma with a slow tint
  Planes hdr = make(UHDR_IMG_FMT_24bppYCbCrP010, w, h, UHDR_CG_BT_2100, UHDR_CT_HLG, UHDR_CR_LIMITED_RANGE);
  Planes sdr = make(UHDR_IMG_FMT_12bppYCbCr420, w, h, UHDR_CG_BT_709, UHDR_CT_SRGB, UHDR_CR_FULL_RANGE);
  for (unsigned y = 0; y < h; y++)
    for (unsigned x = 0; x < w; x++) {
      const float l = 0.5f + 0.45f * std::sin(x / 37.0f) * std::cos(y / 23.0f) + ((int)(lcg(seed) >> 24) - 128) / 4096.0f;
      const float lc = l < 0 ? 0 : (l > 1 ? 1 : l);
      ((uint16_t*)hdr.img.planes[0])[y * hdr.img.stride[0] + x] = (uint16_t)((64 + (int)std::lround(876 * lc)) << 6);
      ((uint8_t*)sdr.img.planes[0])[y * sdr.img.stride[0] + x] = (uint8_t)std::lround(255 * std::pow(lc, 0.8f));
    }
  for (unsigned y = 0; y < h / 2; y++)
    for (unsigned x = 0; x < w / 2; x++) {
      const int cu = 512 + (int)(60 * std::sin(x / 19.0f)), cv = 512 + (int)(60 * std::cos(y / 17.0f));
      ((uint16_t*)hdr.img.planes[1])[y * hdr.img.stride[1] + 2 * x] = (uint16_t)(cu << 6);
      ((uint16_t*)hdr.img.planes[1])[y * hdr.img.stride[1] + 2 * x + 1] = (uint16_t)(cv << 6);
      ((uint8_t*)sdr.img.planes[1])[y * sdr.img.stride[1] + x] = (uint8_t)(cu >> 2);
      ((uint8_t*)sdr.img.planes[2])[y * sdr.img.stride[2] + x] = (uint8_t)(cv >> 2);
    }
  int failures = 0;
  auto check = [&](const char* what, bool ok) { std::printf("%-34s %s\n", what, ok ? "equal" : "DIFFERENT"); failures += !ok; };
  auto same = [](const Planes& a, const Planes& b) { return a.mem == b.mem; };

  // ---- generateGainMap: two pass, 3 channels, scale 1 (the C-API defaults) -------------------------------------
  uhdr_gainmap_metadata_t md;
  std::unique_ptr<uhdr_hip::raw_image_ext> gm;
  uhdr_error_info_t st = hip.generateGainMap(&sdr.img, &hdr.img, &md, gm);
  if (st.error_code != UHDR_CODEC_OK) { std::printf("generateGainMap failed: %s\n", st.detail); return 1; }
  uo_encode_cfg_t cfg = {1, 1, 1.0f, /*BEST_QUALITY*/ 1, FLT_MIN, FLT_MAX, -1.0f, 0, 1};
  uo_metadata_t md_o;
  Planes gm_o = make(UHDR_IMG_FMT_24bppRGB888, w, h, UHDR_CG_BT_2100, UHDR_CT_HLG, UHDR_CR_LIMITED_RANGE);
  gm_o.mem.assign((size_t)gm->stride[0] * h * 3 + 64, 0);
  gm_o.img.planes[0] = gm_o.mem.data();
  gm_o.img.stride[0] = gm->stride[0];
  if (uo_generate_gainmap((const uo_image_t*)&sdr.img, (const uo_image_t*)&hdr.img, &cfg, &md_o, (uo_image_t*)&gm_o.img) != 0) { std::printf("oracle generate failed\n"); return 1; }
  check("generateGainMap map bytes", std::memcmp(gm->planes[0], gm_o.img.planes[0], (size_t)gm->stride[0] * h * 3) == 0);
  check("generateGainMap metadata", std::memcmp(&md, &md_o, sizeof md) == 0);

  // ---- applyGainMap -> linear RGBA F16 and -> HLG RGBA1010102 ---------------------------------------------------
  for (int k = 0; k < 2; k++) {
    const uhdr_color_transfer_t ct = k ? UHDR_CT_HLG : UHDR_CT_LINEAR;
    const uhdr_img_fmt_t fmt = k ? UHDR_IMG_FMT_32bppRGBA1010102 : UHDR_IMG_FMT_64bppRGBAHalfFloat;
    Planes out = make(fmt, w, h, UHDR_CG_UNSPECIFIED, UHDR_CT_UNSPECIFIED, UHDR_CR_UNSPECIFIED), out_o = out;
    out_o.img.planes[0] = out_o.mem.data();
    st = hip.applyGainMap(&sdr.img, gm.get(), &md, ct, fmt, FLT_MAX, &out.img);
    if (st.error_code != UHDR_CODEC_OK) { std::printf("applyGainMap failed: %s\n", st.detail); return 1; }
    if (uo_apply_gainmap((const uo_image_t*)&sdr.img, (const uo_image_t*)gm.get(), &md_o, ct, fmt, FLT_MAX, (uo_image_t*)&out_o.img) != 0) { std::printf("oracle apply failed\n"); return 1; }
    check(k ? "applyGainMap -> HLG 1010102" : "applyGainMap -> linear F16", same(out, out_o));
  }

  // ---- toneMap (P010 -> 4:2:0) and convertYuv (in place) --------------------------------------------------------------
  Planes tm = make(UHDR_IMG_FMT_12bppYCbCr420, w, h, UHDR_CG_UNSPECIFIED, UHDR_CT_UNSPECIFIED, UHDR_CR_UNSPECIFIED), tm_o = tm;
  for (int i = 0; i < 3; i++) tm_o.img.planes[i] = tm_o.mem.data() + ((uint8_t*)tm.img.planes[i] - tm.mem.data());
  st = hip.toneMap(&hdr.img, &tm.img);
  if (st.error_code != UHDR_CODEC_OK) { std::printf("toneMap failed: %s\n", st.detail); return 1; }
  if (uo_tone_map((const uo_image_t*)&hdr.img, (uo_image_t*)&tm_o.img) != 0) { std::printf("oracle toneMap failed\n"); return 1; }
  size_t ndiff = 0;
  for (size_t i = 0; i < tm.mem.size(); i++) ndiff += tm.mem[i] != tm_o.mem[i];
  std::printf("%-34s %zu of %zu bytes differ (allowed: 1e-4)\n", "toneMap", ndiff, tm.mem.size());
  failures += ndiff > tm.mem.size() / 10000;
  Planes cv = sdr, cv_o = sdr;
  for (int i = 0; i < 3; i++) {
    cv.img.planes[i] = cv.mem.data() + ((uint8_t*)sdr.img.planes[i] - sdr.mem.data());
    cv_o.img.planes[i] = cv_o.mem.data() + ((uint8_t*)sdr.img.planes[i] - sdr.mem.data());
  }
  st = hip.convertYuv(&cv.img, UHDR_CG_BT_709, UHDR_CG_DISPLAY_P3);
  if (st.error_code != UHDR_CODEC_OK) { std::printf("convertYuv failed: %s\n", st.detail); return 1; }
  if (uo_convert_yuv((uo_image_t*)&cv_o.img, UHDR_CG_BT_709, UHDR_CG_DISPLAY_P3) != 0) { std::printf("oracle convertYuv failed\n"); return 1; }
  check("convertYuv 709 -> 601", same(cv, cv_o));

  // ---- error convention: same code as the reference for a bad argument -------------------------------------------------
  Planes bad = make(UHDR_IMG_FMT_64bppRGBAHalfFloat, w, h, UHDR_CG_UNSPECIFIED, UHDR_CT_UNSPECIFIED, UHDR_CR_UNSPECIFIED);
  check("applyGainMap(HLG, F16) is rejected", hip.applyGainMap(&sdr.img, gm.get(), &md, UHDR_CT_HLG, UHDR_IMG_FMT_64bppRGBAHalfFloat, FLT_MAX, &bad.img).error_code == UHDR_CODEC_INVALID_PARAM);
  std::printf(failures ? "FAILED (%d)\n" : "ALL CHECKS PASSED\n", failures);
  return failures ? 1 : 0;
}
