// uhdr_hip_seam.cpp -- implementation of the facade's HIP seam (see uhdr_hip_seam.h).
// Compiled by g++ with the REFERENCE's headers on the include path (facade/Makefile) and linked against
// libuhdr_hip.so; it only marshals between the reference's C++ types and the C ABI of include/uhdr_hip.h.
#ifdef UHDR_ENABLE_HEIF
// HeifUltraHdr / AvifUltraHdr call the patched UltraHdr::generateGainMap too and hand the map to a CPU encoder with no seam behind it: the lazy
// gain-map download (tl_lazy_ok) is only sound while JpegR is the one caller.  A facade built with HEIF support has to flush residents there first.
#error "the facade's lazy gain-map download assumes JpegR is the only caller of generateGainMap: build without UHDR_ENABLE_HEIF or add the write-back"
#endif
#include "uhdr_hip_seam.h"

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "uhdr_hip.h"  // after ultrahdr_api.h: reuses the reference's own structs

namespace uhdr_hip_seam {

namespace {
thread_local void* tl_ctxt = nullptr;
thread_local bool tl_lazy_ok = false;
thread_local bool tl_lazy_now = false;  // what uhdr_hip_resident_lazy was last told on this thread's context
std::atomic<unsigned long> g_calls{0};

uhdr_hip_ctx_t* cur() { return static_cast<uhdr_hip_ctx_t*>(tl_ctxt); }

// true: the device produced the call's result.  UNSUPPORTED_FEATURE from the device library means "this
// combination is the reference's to handle" (uhdr_enable_gpu_acceleration may have no effect).
// UHDR_HIP_SEAM_TRACE=1 in the environment: one stderr line per stage call saying where it ran (tests use it to
// prove that an accelerated run really went through the device)
bool trace_on() {
  static const bool on = getenv("UHDR_HIP_SEAM_TRACE") != nullptr;
  return on;
}
// Contexts outlive their codec (round 5).  An application creates one encoder / decoder per image (uhdr_create_encoder ..
// uhdr_release_encoder around every frame is what ultrahdr_app and the reference's own tests do); a uhdr_hip context carries
// a stream, the transfer-function tables, the entropy decoder's table forms, a pinned staging ring and device scratch sized
// for the last image -- a few milliseconds of hipMalloc / hipHostMalloc / uploads that used to be paid inside every
// uhdr_encode / uhdr_decode.  release() parks up to kPoolMax idle contexts, the next codec's first accelerated call takes one.
// UHDR_HIP_SEAM_NO_CTX_POOL=1: one context per codec, destroyed with it (the round-4 behaviour).  Parked contexts are
// never destroyed at process exit (the HIP runtime may be gone by then).
constexpr size_t kPoolMax = 4;
// what a parked context may keep allocated (device scratch + pinned ring): enough for 8K frames; a context that has seen a 16K x 16K image
// gives the excess back before it is parked (uhdr_hip_recycle)
constexpr size_t kPoolKeepBytes = (size_t)1 << 30;
std::mutex g_pool_mu;
struct Parked { void* ctxt; int device; };
std::vector<Parked> g_pool;
bool pool_on() {
  static const bool on = getenv("UHDR_HIP_SEAM_NO_CTX_POOL") == nullptr;
  return on;
}
void* pool_take() {  // a context bound to the CALLING thread's current device only
  if (!pool_on()) return nullptr;
  const int dev = uhdr_hip_current_device();
  std::lock_guard<std::mutex> lk(g_pool_mu);
  for (size_t i = g_pool.size(); i-- > 0;)
    if (g_pool[i].device == dev) {
      void* c = g_pool[i].ctxt;
      g_pool.erase(g_pool.begin() + (long)i);
      return c;
    }
  return nullptr;
}
bool pool_give(void* c) {
  if (!pool_on()) return false;
  // the next codec must find nothing of this one's: latched errors, hints, counters, oversized buffers
  const int dev = uhdr_hip_recycle(static_cast<uhdr_hip_ctx_t*>(c), kPoolKeepBytes);
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (g_pool.size() >= kPoolMax) return false;
  g_pool.push_back({c, dev});
  return true;
}
thread_local double tl_enter_ms = -1.0;
double now_ms() {
  static const auto t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
thread_local double tl_scope_ms = -1.0;
void enter() { tl_enter_ms = now_ms(); }
// every stage outcome goes to the library's process-wide tally (uhdr_hip_seam_stats, include/uhdr_hip.h): tests, bench.py and
// applications read that table; the stderr trace is only for a person watching a run
void note(const char* stage, bool dev, const char* why) {
  const double t = now_ms(), took = tl_enter_ms >= 0 ? t - tl_enter_ms : 0.0;
  tl_enter_ms = -1.0;
  uhdr_hip_seam_note(stage, dev ? 1 : 0, took);
  if (trace_on())
    fprintf(stderr, "uhdr_hip_seam: [%8.2f ms, took %6.2f] %s -> %s%s%s\n", t, took, stage, dev ? "device" : "reference CPU path (", dev ? "" : (why ? why : ""),
            dev ? "" : ")");
}
bool handled(const uhdr_error_info_t& s, const char* stage) {
  const bool dev = s.error_code != UHDR_CODEC_UNSUPPORTED_FEATURE;
  note(stage, dev, s.has_detail ? s.detail : "");
  if (dev) g_calls.fetch_add(1, std::memory_order_relaxed);
  else if (cur()) uhdr_hip_resident_begin(cur());  // the reference's CPU code runs next and may write in place: drop the device copies
  return dev;
}
}  // namespace

Scope::Scope(bool enable, void** slot, bool lazy) : mPrev(tl_ctxt), mFailed(false) {
  memset(&mError, 0, sizeof mError);
  tl_lazy_ok = enable && lazy && !getenv("UHDR_HIP_SEAM_EAGER_DOWNLOADS");
  if (!enable) {
    tl_ctxt = nullptr;
    return;
  }
  if (*slot == nullptr) *slot = pool_take();
  if (*slot == nullptr) {
    uhdr_error_info_t err;
    memset(&err, 0, sizeof err);
    *slot = uhdr_hip_create(-1, &err);
    if (*slot == nullptr) {  // GPU acceleration was asked for and cannot be had: say so, do not compute on the CPU silently
      mFailed = true;
      mError = err;
      if (mError.error_code == UHDR_CODEC_OK) mError.error_code = UHDR_CODEC_ERROR;
      tl_ctxt = nullptr;
      return;
    }
  }
  tl_ctxt = *slot;
  tl_lazy_now = false;
  // one uhdr_encode / uhdr_decode: what the JPEG decode stage leaves in the JpegDecoderHelper buffers stays on the device
  // for the stage that reads those buffers next (decodeJPEGR -> applyGainMap, jpegr.cpp:1478-1530)
  uhdr_hip_resident_begin(cur());
  tl_scope_ms = now_ms();
  if (trace_on()) fprintf(stderr, "uhdr_hip_seam: [%8.2f ms] accelerated call begins\n", tl_scope_ms);
}
Scope::~Scope() {
  if (tl_ctxt && trace_on()) fprintf(stderr, "uhdr_hip_seam: [%8.2f ms] accelerated call ends\n", now_ms());
  if (tl_ctxt) uhdr_hip_resident_end(cur());
  if (tl_ctxt && tl_scope_ms >= 0) uhdr_hip_seam_note("uhdr_call", 1, now_ms() - tl_scope_ms);  // the whole accelerated uhdr_encode / uhdr_decode
  tl_scope_ms = -1.0;
  tl_ctxt = mPrev;
  tl_lazy_ok = false;
}

void lazy_downloads(bool on) {
  if (!cur()) return;
  tl_lazy_now = on && tl_lazy_ok;
  uhdr_hip_resident_lazy(cur(), tl_lazy_now);
}
bool defer_copy(uhdr_raw_image_t* src, uhdr_raw_image_t* dst) {
  if (!cur() || !src || !dst) return false;
  if (uhdr_hip_resident_adopt(cur(), src, dst)) {
    dst->cg = src->cg;  // gainmapmath.cpp:1505-1507
    dst->ct = src->ct;
    dst->range = src->range;
    uhdr_hip_seam_note("gainmap_copy_deferred", 1, 0.0);
    if (trace_on()) fprintf(stderr, "uhdr_hip_seam: [%8.2f ms] gain-map image copy deferred: the image stays on the device until asked for\n", now_ms());
    return true;
  }
  // the host copy reads the helper's buffer: it has to be written by now
  const uhdr_error_info_t s = uhdr_hip_resident_flush(cur());
  if (s.error_code != UHDR_CODEC_OK) fprintf(stderr, "uhdr_hip_seam: write-back of a decoded image failed: %s\n", s.has_detail ? s.detail : "");
  return false;
}
bool materialize(void* ctxt) {
  if (!ctxt) return true;
  const double t0 = now_ms();
  const uhdr_error_info_t s = uhdr_hip_resident_materialize(static_cast<uhdr_hip_ctx_t*>(ctxt));
  if (s.error_code != UHDR_CODEC_OK) {
    fprintf(stderr, "uhdr_hip_seam: download of the gain-map image failed: %s\n", s.has_detail ? s.detail : "");
    return false;
  }
  uhdr_hip_seam_note("gainmap_image_asked_for", 1, now_ms() - t0);
  if (trace_on()) fprintf(stderr, "uhdr_hip_seam: [%8.2f ms, took %6.2f] gain-map image asked for\n", now_ms(), now_ms() - t0);
  return true;
}
void forget(void* ctxt) {
  if (ctxt) uhdr_hip_resident_forget(static_cast<uhdr_hip_ctx_t*>(ctxt));
}

void release(void* ctxt) {
  if (!ctxt) return;
  uhdr_hip_resident_forget(static_cast<uhdr_hip_ctx_t*>(ctxt));  // a deferred copy into the codec that is going away
  if (!pool_give(ctxt)) uhdr_hip_destroy(static_cast<uhdr_hip_ctx_t*>(ctxt));
}
unsigned long calls_on_device() { return g_calls.load(std::memory_order_relaxed); }

bool apply_gainmap(uhdr_raw_image_t* sdr_intent, uhdr_raw_image_t* gainmap_img,
                   ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata, uhdr_color_transfer_t output_ct,
                   uhdr_img_fmt_t output_format, float max_display_boost, uhdr_raw_image_t* dest,
                   uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur() || !gainmap_metadata) return false;
    enter();
    // the version check is the one thing uhdr_gainmap_metadata_ext_t adds (jpegr.cpp:1546-1555)
    if (gainmap_metadata->version.compare(ultrahdr::kJpegrVersion)) return false;  // the reference words that error
    const uhdr_gainmap_metadata_t md = *gainmap_metadata;  // slice off the version string
    *st = uhdr_hip_apply_gainmap(cur(), sdr_intent, gainmap_img, &md, output_ct, output_format, max_display_boost, dest);
    return handled(*st, "apply_gainmap");
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool generate_gainmap(uhdr_raw_image_t* sdr_intent, uhdr_raw_image_t* hdr_intent,
                      ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata,
                      std::unique_ptr<ultrahdr::uhdr_raw_image_ext_t>& gainmap_img, bool sdr_is_601,
                      bool use_luminance, int* scale_factor, bool multi_channel, float gamma,
                      uhdr_enc_preset_t preset, float min_content_boost, float max_content_boost,
                      float target_disp_peak_brightness, uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur() || !sdr_intent || !hdr_intent || !gainmap_metadata) return false;
    enter();
    // map geometry and the tiny-image fallback exactly as jpegr.cpp:690-706
    const unsigned w = sdr_intent->w, h = sdr_intent->h;
    int s = *scale_factor;
    if (s < 1) return false;
    unsigned mw = w / s, mh = h / s;
    if (mw == 0 || mh == 0) {
      const unsigned m = w < h ? w : h;
      s = m / 8 ? (int)(m / 8) : 1;
      mw = w / s;
      mh = h / s;
    }
    if (mw == 0 || mh == 0) return false;
    uhdr_hip_encode_cfg_t cfg;
    cfg.map_dimension_scale_factor = s;
    cfg.use_multi_channel_gainmap = multi_channel;
    cfg.gamma = gamma;
    cfg.preset = preset;
    cfg.min_content_boost = min_content_boost;
    cfg.max_content_boost = max_content_boost;
    cfg.target_disp_peak_nits = target_disp_peak_brightness;
    cfg.sdr_is_601 = sdr_is_601;
    cfg.use_luminance = use_luminance;
    auto img = std::make_unique<ultrahdr::uhdr_raw_image_ext_t>(
        multi_channel ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_8bppYCbCr400, hdr_intent->cg, hdr_intent->ct,
        hdr_intent->range, mw, mh, 64);  // jpegr.cpp:714-716
    uhdr_gainmap_metadata_t md;
    memset(&md, 0, sizeof md);
    // the map's only reader is the compressImage that follows (jpegr.cpp:253-257 and its siblings), which finds it on the device:
    // it is not downloaded (any CPU stage in between gets it written first, drop_resident)
    // ... only inside a call whose Scope allows it (uhdr_encode passes lazy_downloads = true: every encodeJPEGR variant hands the
    // map straight to compressGainMap, jpegr.cpp:211-217, 253-257, 316-320, 377-381), and the previous state comes back afterwards
    const bool was = tl_lazy_now;
    if (tl_lazy_ok) uhdr_hip_resident_lazy(cur(), 1);
    *st = uhdr_hip_generate_gainmap(cur(), sdr_intent, hdr_intent, &cfg, &md, img.get());
    uhdr_hip_resident_lazy(cur(), was ? 1 : 0);
    if (!handled(*st, "generate_gainmap")) return false;
    if (st->error_code == UHDR_CODEC_OK) {
      static_cast<uhdr_gainmap_metadata_t&>(*gainmap_metadata) = md;
      gainmap_img = std::move(img);
      *scale_factor = s;
    }
    return true;
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool encode_api1(uhdr_raw_image_t* hdr_intent, uhdr_raw_image_t* sdr_intent, int base_quality, int map_quality, int* scale_factor,
                 bool multi_channel, float gamma, uhdr_enc_preset_t preset, float min_content_boost, float max_content_boost,
                 float target_disp_peak_brightness, const void* base_icc, size_t base_icc_size, const void* map_icc,
                 size_t map_icc_size, const char* map_comment, ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata,
                 Api1Files* out, uhdr_error_info_t* st) {
  // declining costs nothing: no upload has happened, no resident copy exists yet -- the per-stage seams run as before
  if (!cur() || !hdr_intent || !sdr_intent || !gainmap_metadata || !out || !scale_factor) return false;
  if (getenv("UHDR_HIP_SEAM_NO_FUSED_ENCODE") || getenv("UHDR_HIP_SEAM_CPU_ENTROPY") || getenv("UHDR_HIP_SEAM_RESTART_INTERVAL") ||
      getenv("UHDR_HIP_SEAM_DEVICE_ENTROPY") || getenv("UHDR_HIP_SEAM_EAGER_DOWNLOADS"))
    return false;
  if (sdr_intent->fmt != UHDR_IMG_FMT_12bppYCbCr420 || *scale_factor < 1) return false;
  const unsigned w = sdr_intent->w, h = sdr_intent->h;
  const int s = *scale_factor;
  if (w == 0 || h == 0 || w % 16 || h % 16 || w > 65535 || h > 65535) return false;
  const unsigned mw = w / (unsigned)s, mh = h / (unsigned)s;
  if (mw == 0 || mh == 0 || mw % 8 || mh % 8) return false;  // (the tiny-image fallback of jpegr.cpp:690-706 stays with generate_gainmap)
  if (base_icc_size > 65533 || map_icc_size > 65533 || (map_comment && strlen(map_comment) > 65533)) return false;
  if (base_quality < 0 || base_quality > 100 || map_quality < 0 || map_quality > 100) return false;
  enter();
  uhdr_hip_encode_cfg_t cfg;
  cfg.map_dimension_scale_factor = s;
  cfg.use_multi_channel_gainmap = multi_channel;
  cfg.gamma = gamma;
  cfg.preset = preset;
  cfg.min_content_boost = min_content_boost;
  cfg.max_content_boost = max_content_boost;
  cfg.target_disp_peak_nits = target_disp_peak_brightness;
  cfg.sdr_is_601 = 0;     // generateGainMap's defaults, as encodeJPEGR API-1 calls it (jpegr.cpp:259)
  cfg.use_luminance = 1;
  uint16_t qt_base[2][64], qt_map[2][64];  // jpeg_set_quality(quality, TRUE) (jpegencoderhelper.cpp:187)
  for (int t = 0; t < 2; t++) {
    uhdr_hip_jpeg_quant_table(base_quality, t, qt_base[t]);
    uhdr_hip_jpeg_quant_table(map_quality, t, qt_map[t]);
  }
  const int nch = multi_channel ? 3 : 1;
  // the two files' headers first (their sizes place the scans inside the output buffers)
  uhdr_hip_jpeg_scan_t sb, sm;
  memset(&sb, 0, sizeof sb);
  memset(&sm, 0, sizeof sm);
  sb.num_components = 3;
  sb.w = w; sb.h = h;
  for (int i = 0; i < 3; i++) {
    sb.blocks_w[i] = (int)(i ? w / 16 : w / 8);
    sb.blocks_h[i] = (int)(i ? h / 16 : h / 8);
    sb.h_samp[i] = sb.v_samp[i] = i ? 1 : 2;
  }
  sm.num_components = nch;
  sm.w = mw; sm.h = mh;
  for (int i = 0; i < nch; i++) { sm.blocks_w[i] = (int)(mw / 8); sm.blocks_h[i] = (int)(mh / 8); sm.h_samp[i] = sm.v_samp[i] = 1; }
  unsigned char hb[2048], hm[2048];
  const unsigned char none = 0;
  const size_t nhb = uhdr_hip_jpeg_assemble(&sb, qt_base[0], qt_base[1], &none, 0, hb, sizeof hb);
  const size_t nhm = uhdr_hip_jpeg_assemble(&sm, qt_map[0], qt_map[nch == 3 ? 1 : 0], &none, 0, hm, sizeof hm);
  if (nhb < 22 || nhm < 22) return false;
  // file = SOI + JFIF APP0 (20 bytes) | APP2 ICC | COM | DQT .. SOS | data | EOI  (jcmarker.c's order around the helper's markers)
  auto lead = [](size_t nh, size_t icc, const char* com) { return 20 + (icc ? 4 + icc : 0) + (com ? 4 + strlen(com) : 0) + (nh - 22); };
  const size_t lead_b = lead(nhb, base_icc ? base_icc_size : 0, nullptr), lead_m = lead(nhm, map_icc ? map_icc_size : 0, map_comment);
  size_t cap_b = (size_t)w * h * 3 / 2 + (1u << 16), cap_m = (size_t)mw * mh * nch + (1u << 16);  // one byte per coefficient: never seen exceeded
  if (const char* e = getenv("UHDR_HIP_SEAM_TEST_SMALL_CAP")) {  // (tests: the second attempt below)
    const size_t div = (size_t)(atoi(e) > 1 ? atoi(e) : 16);
    cap_b /= div;
    cap_m /= div;
  }
  uhdr_gainmap_metadata_t md;
  memset(&md, 0, sizeof md);
  uhdr_raw_image_t gm_desc;
  memset(&gm_desc, 0, sizeof gm_desc);
  size_t nb = 0, nm = 0;
  for (int attempt = 0;; attempt++) {
    out->base_data.reset(new (std::nothrow) unsigned char[lead_b + cap_b + 2]);
    out->gainmap_data.reset(new (std::nothrow) unsigned char[lead_m + cap_m + 2]);
    if (!out->base_data || !out->gainmap_data) return false;
    nb = nm = 0;
    *st = uhdr_hip_encode_api1_scans(cur(), sdr_intent, hdr_intent, &cfg, UHDR_CG_DISPLAY_P3, qt_base, qt_map, &md, &gm_desc, out->base_data.get() + lead_b,
                                     cap_b, &nb, out->gainmap_data.get() + lead_m, cap_m, &nm);
    // a stream busier than the first guess: the encoder reports the size it needs -- once more with room for that (the device chain is ~1 ms at 4K;
    // the per-stage seams this used to fall to carry every intermediate over PCIe).  Any other MEM_ERROR (a device allocation) has no such sizes.
    if (st->error_code == UHDR_CODEC_MEM_ERROR && attempt == 0 && (nb > cap_b || nm > cap_m)) {
      note("encode_api1_fused", true, "scan larger than one byte per coefficient: second attempt with the reported sizes");
      if (nb > cap_b) cap_b = nb + nb / 16 + (1u << 16);
      if (nm > cap_m) cap_m = nm + nm / 16 + (1u << 16);
      continue;
    }
    break;
  }
  if (st->error_code == UHDR_CODEC_MEM_ERROR) {  // not a matter of the output buffers (or still too small): the per-stage seams / the reference
    note("encode_api1_fused", false, st->has_detail ? st->detail : "per-stage seams");
    return false;
  }
  if (!handled(*st, "encode_api1_fused")) return false;
  if (st->error_code != UHDR_CODEC_OK) return true;
  auto finish = [](unsigned char* f, const unsigned char* hdr, size_t nh, const void* icc, size_t icc_size, const char* com, size_t lead_bytes, size_t scan_bytes) {
    unsigned char* p = f;
    memcpy(p, hdr, 20); p += 20;
    auto marker = [&](int code, const void* d, size_t n) {
      p[0] = 0xff; p[1] = (unsigned char)code; p[2] = (unsigned char)((n + 2) >> 8); p[3] = (unsigned char)((n + 2) & 0xff);
      memcpy(p + 4, d, n);
      p += 4 + n;
    };
    if (icc && icc_size) marker(0xe2, icc, icc_size);
    if (com) marker(0xfe, com, strlen(com));
    memcpy(p, hdr + 20, nh - 22); p += nh - 22;
    (void)lead_bytes;  // == p - f: the scan is already in place behind it
    p += scan_bytes;
    p[0] = 0xff; p[1] = 0xd9;
    return (size_t)(p + 2 - f);
  };
  out->base_size = finish(out->base_data.get(), hb, nhb, base_icc, base_icc ? base_icc_size : 0, nullptr, lead_b, nb);
  out->gainmap_size = finish(out->gainmap_data.get(), hm, nhm, map_icc, map_icc ? map_icc_size : 0, map_comment, lead_m, nm);
  out->base_capacity = lead_b + cap_b + 2;
  out->gainmap_capacity = lead_m + cap_m + 2;
  static_cast<uhdr_gainmap_metadata_t&>(*gainmap_metadata) = md;
  *scale_factor = s;
  return true;
}

namespace {
// file = SOI + JFIF APP0 (20 bytes) | APP2 ICC | COM | DQT .. SOS | data | EOI  (jcmarker.c's order around the helper's markers); the scan
// already sits at f + lead
size_t finish_file(unsigned char* f, const unsigned char* hdr, size_t nh, const void* icc, size_t icc_size, const char* com, size_t scan_bytes) {
  unsigned char* p = f;
  memcpy(p, hdr, 20); p += 20;
  auto marker = [&](int code, const void* d, size_t n) {
    p[0] = 0xff; p[1] = (unsigned char)code; p[2] = (unsigned char)((n + 2) >> 8); p[3] = (unsigned char)((n + 2) & 0xff);
    memcpy(p + 4, d, n);
    p += 4 + n;
  };
  if (icc && icc_size) marker(0xe2, icc, icc_size);
  if (com) marker(0xfe, com, strlen(com));
  memcpy(p, hdr + 20, nh - 22); p += nh - 22;
  p += scan_bytes;
  p[0] = 0xff; p[1] = 0xd9;
  return (size_t)(p + 2 - f);
}
size_t lead_bytes(size_t nh, size_t icc, const char* com) { return 20 + (icc ? 4 + icc : 0) + (com ? 4 + strlen(com) : 0) + (nh - 22); }
}  // namespace

bool encode_api0(uhdr_raw_image_t* hdr_intent, int base_quality, int map_quality, int* scale_factor, bool multi_channel, float gamma,
                 float min_content_boost, float max_content_boost, float target_disp_peak_brightness,
                 IccBytes (*base_icc)(void* user, uhdr_color_gamut_t cg), void* icc_user, const void* map_icc, size_t map_icc_size,
                 const char* map_comment, ultrahdr::uhdr_gainmap_metadata_ext_t* gainmap_metadata, uhdr_color_gamut_t* sdr_cg, Api1Files* out,
                 uhdr_error_info_t* st) {
  // declining costs nothing: no upload has happened -- the per-stage seams (tone_map, generate_gainmap, convert_raw_input_to_ycbcr, 2 x
  // jpeg_encode_scan) run as before
  if (!cur() || !hdr_intent || !gainmap_metadata || !out || !scale_factor || !base_icc || !sdr_cg) return false;
  if (getenv("UHDR_HIP_SEAM_NO_FUSED_ENCODE") || getenv("UHDR_HIP_SEAM_CPU_ENTROPY") || getenv("UHDR_HIP_SEAM_RESTART_INTERVAL") ||
      getenv("UHDR_HIP_SEAM_DEVICE_ENTROPY") || getenv("UHDR_HIP_SEAM_EAGER_DOWNLOADS"))
    return false;
  if (hdr_intent->fmt != UHDR_IMG_FMT_32bppRGBA1010102 && hdr_intent->fmt != UHDR_IMG_FMT_64bppRGBAHalfFloat) return false;
  if (*scale_factor != 1) return false;
  const unsigned w = hdr_intent->w, h = hdr_intent->h;
  if (w == 0 || h == 0 || w % 8 || h % 8 || w > 65535 || h > 65535) return false;
  if (map_icc_size > 65533 || (map_comment && strlen(map_comment) > 65533)) return false;
  if (base_quality < 0 || base_quality > 100 || map_quality < 0 || map_quality > 100) return false;
  enter();
  uhdr_hip_encode_cfg_t cfg;
  cfg.map_dimension_scale_factor = 1;
  cfg.use_multi_channel_gainmap = multi_channel;
  cfg.gamma = gamma;
  cfg.preset = UHDR_USAGE_REALTIME;  // jpegr.cpp:207
  cfg.min_content_boost = min_content_boost;
  cfg.max_content_boost = max_content_boost;
  cfg.target_disp_peak_nits = target_disp_peak_brightness;
  cfg.sdr_is_601 = 0;
  cfg.use_luminance = 0;  // jpegr.cpp:213-214
  uint16_t qt_base[2][64], qt_map[2][64];
  for (int t = 0; t < 2; t++) {
    uhdr_hip_jpeg_quant_table(base_quality, t, qt_base[t]);
    uhdr_hip_jpeg_quant_table(map_quality, t, qt_map[t]);
  }
  const int nch = multi_channel ? 3 : 1;
  uhdr_hip_jpeg_scan_t sb, sm;
  memset(&sb, 0, sizeof sb);
  memset(&sm, 0, sizeof sm);
  sb.num_components = 3;
  sm.num_components = nch;
  sb.w = sm.w = w;
  sb.h = sm.h = h;
  for (int i = 0; i < 3; i++) { sb.blocks_w[i] = (int)(w / 8); sb.blocks_h[i] = (int)(h / 8); sb.h_samp[i] = sb.v_samp[i] = 1; }
  for (int i = 0; i < nch; i++) { sm.blocks_w[i] = (int)(w / 8); sm.blocks_h[i] = (int)(h / 8); sm.h_samp[i] = sm.v_samp[i] = 1; }
  unsigned char hb[2048], hm[2048];
  const unsigned char none = 0;
  const size_t nhb = uhdr_hip_jpeg_assemble(&sb, qt_base[0], qt_base[1], &none, 0, hb, sizeof hb);
  const size_t nhm = uhdr_hip_jpeg_assemble(&sm, qt_map[0], qt_map[nch == 3 ? 1 : 0], &none, 0, hm, sizeof hm);
  if (nhb < 22 || nhm < 22) return false;
  // the base file's ICC profile depends on the gamut the tone map gives its rendition: room for the largest the helper writes is left in
  // front of the scan, the file is closed up afterwards
  constexpr size_t kIccRoom = 4096;
  const size_t lead_m = lead_bytes(nhm, map_icc ? map_icc_size : 0, map_comment);
  const size_t lead_b_max = lead_bytes(nhb, kIccRoom, nullptr);
  size_t cap_b = (size_t)w * h * 3 + (1u << 16), cap_m = (size_t)w * h * nch + (1u << 16);  // one byte per coefficient: never seen exceeded
  if (const char* e = getenv("UHDR_HIP_SEAM_TEST_SMALL_CAP")) {  // (tests: the second attempt below)
    const size_t div = (size_t)(atoi(e) > 1 ? atoi(e) : 16);
    cap_b /= div;
    cap_m /= div;
  }
  uhdr_gainmap_metadata_t md;
  memset(&md, 0, sizeof md);
  uhdr_raw_image_t gm_desc;
  memset(&gm_desc, 0, sizeof gm_desc);
  size_t nb = 0, nm = 0;
  uhdr_color_gamut_t cg = UHDR_CG_UNSPECIFIED;
  for (int attempt = 0;; attempt++) {
    out->base_data.reset(new (std::nothrow) unsigned char[lead_b_max + cap_b + 2]);
    out->gainmap_data.reset(new (std::nothrow) unsigned char[lead_m + cap_m + 2]);
    if (!out->base_data || !out->gainmap_data) return false;
    nb = nm = 0;
    *st = uhdr_hip_encode_api0_scans(cur(), hdr_intent, &cfg, qt_base, qt_map, &md, &gm_desc, &cg, out->base_data.get() + lead_b_max, cap_b, &nb,
                                     out->gainmap_data.get() + lead_m, cap_m, &nm);
    if (st->error_code == UHDR_CODEC_MEM_ERROR && attempt == 0 && (nb > cap_b || nm > cap_m)) {  // (as in encode_api1: once more with the reported sizes)
      note("encode_api0_fused", true, "scan larger than one byte per coefficient: second attempt with the reported sizes");
      if (nb > cap_b) cap_b = nb + nb / 16 + (1u << 16);
      if (nm > cap_m) cap_m = nm + nm / 16 + (1u << 16);
      continue;
    }
    break;
  }
  if (st->error_code == UHDR_CODEC_MEM_ERROR) {
    note("encode_api0_fused", false, st->has_detail ? st->detail : "per-stage seams");
    return false;
  }
  if (!handled(*st, "encode_api0_fused")) return false;
  if (st->error_code != UHDR_CODEC_OK) return true;
  const IccBytes icc = base_icc(icc_user, cg);
  if (icc.size > kIccRoom || icc.size > 65533) {
    note("encode_api0_fused", false, "ICC profile larger than expected");
    return false;
  }
  const size_t lead_b = lead_bytes(nhb, icc.data ? icc.size : 0, nullptr);
  if (lead_b != lead_b_max) memmove(out->base_data.get() + lead_b, out->base_data.get() + lead_b_max, nb);
  out->base_size = finish_file(out->base_data.get(), hb, nhb, icc.data, icc.data ? icc.size : 0, nullptr, nb);
  out->gainmap_size = finish_file(out->gainmap_data.get(), hm, nhm, map_icc, map_icc ? map_icc_size : 0, map_comment, nm);
  out->base_capacity = lead_b_max + cap_b + 2;
  out->gainmap_capacity = lead_m + cap_m + 2;
  static_cast<uhdr_gainmap_metadata_t&>(*gainmap_metadata) = md;
  *sdr_cg = cg;
  return true;
}

bool tone_map(uhdr_raw_image_t* hdr_intent, uhdr_raw_image_t* sdr_intent, uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur()) return false;
    enter();
    *st = uhdr_hip_tone_map(cur(), hdr_intent, sdr_intent);
    return handled(*st, "tone_map");
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool convert_yuv(uhdr_raw_image_t* image, uhdr_color_gamut_t src_encoding, uhdr_color_gamut_t dst_encoding,
                 uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur()) return false;
    enter();
    *st = uhdr_hip_convert_yuv(cur(), image, src_encoding, dst_encoding);
    return handled(*st, "convert_yuv");
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool convert_raw_input_to_ycbcr(uhdr_raw_image_t* src, bool chroma_sampling_enabled,
                                std::unique_ptr<ultrahdr::uhdr_raw_image_ext_t>* dst) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur() || !src) return false;
    if (src->fmt != UHDR_IMG_FMT_32bppRGBA1010102 && src->fmt != UHDR_IMG_FMT_32bppRGBA8888 && src->fmt != UHDR_IMG_FMT_24bppRGB888)
      return false;  // YCbCr inputs are a plain copy in the reference (gainmapmath.cpp:1475-1480)
    if (src->cg != UHDR_CG_BT_709 && src->cg != UHDR_CG_DISPLAY_P3 && src->cg != UHDR_CG_BT_2100) return false;
    enter();
    const bool ten = src->fmt == UHDR_IMG_FMT_32bppRGBA1010102;
    const uhdr_img_fmt_t fmt = ten ? (chroma_sampling_enabled ? UHDR_IMG_FMT_24bppYCbCrP010 : UHDR_IMG_FMT_30bppYCbCr444)
                                   : (chroma_sampling_enabled ? UHDR_IMG_FMT_12bppYCbCr420 : UHDR_IMG_FMT_24bppYCbCr444);
    auto img = std::make_unique<ultrahdr::uhdr_raw_image_ext_t>(fmt, src->cg, src->ct, UHDR_CR_FULL_RANGE, src->w, src->h, 64);
    const uhdr_error_info_t s = uhdr_hip_convert_raw_input_to_ycbcr(cur(), src, chroma_sampling_enabled, img.get());
    if (!handled(s, "convert_raw_input_to_ycbcr")) return false;
    if (s.error_code != UHDR_CODEC_OK) {
      fprintf(stderr, "uhdr_hip_seam: convert_raw_input_to_ycbcr failed on the device: %s\n", s.has_detail ? s.detail : "");
      *dst = nullptr;  // the reference's own failure value (callers check for nullptr)
      return true;
    }
    *dst = std::move(img);
    return true;
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool fdct_planes(int ncomp, const unsigned char* const planes[3], const unsigned int strides[3],
                 const unsigned int blocks_w[3], const unsigned int blocks_h[3], const unsigned short* const qtables[3],
                 short* const coefs[3], uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur()) return false;
    enter();
    for (int c = 0; c < ncomp; c++) {
      *st = uhdr_hip_fdct_quant(cur(), planes[c], strides[c], (int)blocks_w[c], (int)blocks_h[c], qtables[c], coefs[c]);
      if (st->error_code != UHDR_CODEC_OK) return c == 0 ? handled(*st, "fdct_planes") : true;
    }
    handled(*st, "fdct_planes");
    return true;
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool idct_planes(int ncomp, const short* const coefs[3], const unsigned int blocks_w[3], const unsigned int blocks_h[3],
                 const unsigned short* const qtables[3], unsigned char* const planes[3], const unsigned int strides[3],
                 uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur()) return false;
    enter();
    for (int c = 0; c < ncomp; c++) {
      *st = uhdr_hip_idct_dequant(cur(), coefs[c], (int)blocks_w[c], (int)blocks_h[c], qtables[c], planes[c], strides[c]);
      if (st->error_code != UHDR_CODEC_OK) return c == 0 ? handled(*st, "idct_planes") : true;
    }
    handled(*st, "idct_planes");
    return true;
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool decode_scan(const void* hdr, const unsigned char* data, size_t bytes, int out_channels, int libjpeg_variant,
                 unsigned char* const planes[3], const unsigned int hstride[3], const unsigned int vstride[3],
                 uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur()) return false;
    enter();
    // (the compressed bytes go up through the library's pinned ring -- round 6; round 5 kept a thread-lifetime copy of every scan here)
    *st = uhdr_hip_jpeg_decode_scan(cur(), static_cast<const uhdr_hip_jpeg_header_t*>(hdr), data, bytes, out_channels, libjpeg_variant, planes,
                                    hstride, vstride);
    // corrupt entropy-coded data: libjpeg decodes such files with warnings and padding; that behaviour stays libjpeg's
    if (st->error_code == UHDR_CODEC_INVALID_PARAM) {
      note("jpeg_decode_scan", false, st->has_detail ? st->detail : "");
      return false;
    }
    return handled(*st, "jpeg_decode_scan");
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool encode_scan(const void* scan, const void* qtables, const unsigned char* const planes[3], const unsigned int strides[3],
                 int rgb_channels, unsigned char* out, size_t cap, size_t* bytes, uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur()) return false;
    enter();
    // the IMAGE's planes with the caller's strides: partial edge blocks are padded on the device by the helper's own rules
    *st = uhdr_hip_jpeg_encode_image(cur(), static_cast<const uhdr_hip_jpeg_scan_t*>(scan), static_cast<const uint16_t(*)[64]>(qtables), planes, strides,
                                     rgb_channels, out, cap, bytes);
    return handled(*st, "jpeg_encode_scan");
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool effect(int kind, int p0, int p1, int dst_w, int dst_h, uhdr_raw_image_t* src,
            std::unique_ptr<ultrahdr::uhdr_raw_image_ext_t>* dst) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur() || !src || dst_w <= 0 || dst_h <= 0) return false;
    enter();
    auto img = std::make_unique<ultrahdr::uhdr_raw_image_ext_t>(src->fmt, src->cg, src->ct, src->range, (unsigned)dst_w, (unsigned)dst_h, 64);
    const uhdr_error_info_t s = uhdr_hip_apply_effect(cur(), kind, p0, p1, src, img.get());
    if (!handled(s, kind == 0 ? "effect_rotate" : kind == 1 ? "effect_mirror" : kind == 2 ? "effect_crop" : "effect_resize")) return false;
    if (s.error_code != UHDR_CODEC_OK) {  // an argument the reference's own code would also refuse, or a device error: let the CPU code speak
      fprintf(stderr, "uhdr_hip_seam: effect %d failed on the device: %s\n", kind, s.has_detail ? s.detail : "");
      return false;
    }
    *dst = std::move(img);
    return true;
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}

bool jpeg_rgb_to_ycc(const uhdr_raw_image_t* rgb, uhdr_raw_image_t* ycc, uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
    if (!cur()) return false;
    enter();
    *st = uhdr_hip_jpeg_rgb_to_ycc(cur(), rgb, ycc);
    return handled(*st, "jpeg_rgb_to_ycc");
  }();
  if (!on_device_) drop_resident();
  return on_device_;
}
bool jpeg_ycc_to_rgb(const uhdr_raw_image_t* ycc, int libjpeg_variant, uhdr_raw_image_t* rgb, uhdr_error_info_t* st) {
  if (!cur()) return false;
  enter();
  *st = uhdr_hip_jpeg_ycc_to_rgb(cur(), ycc, libjpeg_variant, rgb);
  return handled(*st, "jpeg_ycc_to_rgb");
}
bool enabled() { return cur() != nullptr; }
void drop_resident() {
  if (cur()) uhdr_hip_resident_begin(cur());
}

}  // namespace uhdr_hip_seam
