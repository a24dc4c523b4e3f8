// C ABI of libuhdr_hip.so (include/uhdr_hip.h): argument validation that mirrors the reference
// operators, per-call table preparation, host<->device staging and kernel launches.
// There is deliberately NO CPU implementation behind these entry points: if HIP is unusable the
// calls fail with UHDR_CODEC_ERROR.
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "exact_math.h"
#include "host_tables.h"
#include "uhdr_types.h"
#include "rccl_bind.h"

using namespace uhdr;

// -------------------------------------------------------------------------------------------------
// helpers
// -------------------------------------------------------------------------------------------------
static uhdr_error_info_t ok_status() {
  uhdr_error_info_t s;
  s.error_code = UHDR_CODEC_OK;
  s.has_detail = 0;
  s.detail[0] = 0;
  return s;
}
static uhdr_error_info_t err_status(uhdr_codec_err_t code, const char* fmt, ...) {
  uhdr_error_info_t s;
  s.error_code = code;
  s.has_detail = 1;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(s.detail, sizeof s.detail, fmt, ap);
  va_end(ap);
  return s;
}
#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return err_status(e_ == hipErrorOutOfMemory ? UHDR_CODEC_MEM_ERROR : UHDR_CODEC_ERROR, \
                        "HIP error '%s' at %s:%d", hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
#define UHDR_TRY(expr)                                 \
  do {                                                 \
    uhdr_error_info_t s_ = (expr);                     \
    if (s_.error_code != UHDR_CODEC_OK) return s_;     \
  } while (0)

namespace {
// UHDR_HIP_CLOCK_DEBUG: host-side timestamps inside the JPEG decode entry points (where a call's wall time goes)
struct DbgClock {
  bool on;
  std::chrono::steady_clock::time_point t0;
  DbgClock() : on(getenv("UHDR_HIP_CLOCK_DEBUG") != nullptr), t0(std::chrono::steady_clock::now()) {}
  void mark(const char* what) const {
    if (on) fprintf(stderr, "uhdr_hip:   [%7.1f us] %s\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(), what);
  }
};

struct ProfEntry {
  hipEvent_t a, b;
  std::string family;
};

constexpr int kTableSlots = 4;

struct DeviceBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace

struct uhdr_hip_ctx {
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // static LUTs on the device
  float* d_srgb = nullptr;
  float* d_hlg_inv = nullptr;
  float* d_pq_inv = nullptr;
  float* d_hlg_oetf = nullptr;
  float* d_pq_oetf = nullptr;
  float* d_hlg_buckets = nullptr;   // quad kernel: output-code bucket tables (host_tables.cpp: make_bucket_table)
  float* d_pq_buckets = nullptr;
  float* d_hlg_buckets_pre = nullptr;  // ... taking the value before the nit scaling
  float* d_pq_buckets_pre = nullptr;
  float* d_hlg_inv_ootf = nullptr;  // hlgInvOetfLUT followed by hlgOotfApprox, per table node
  double* d_math = nullptr;         // exact_math.h tables
  // per-call apply tables: ring of pinned host slots + matching device slots
  float* h_tab[kTableSlots] = {};
  float* d_tab[kTableSlots] = {};
  size_t tab_cap[kTableSlots] = {};
  hipEvent_t tab_ev[kTableSlots] = {};
  std::string tab_key[kTableSlots];
  int tab_next = 0;
  // scratch for host-buffer entry points and two-pass generation
  DeviceBuf scratch[8];
  uhdr_hip_stats_t stats = {};    // uhdr_hip_get_stats: which route the entropy stage took, call by call
  bool huff_serial_ok = true;  // uhdr_hip_jpeg_decode_scan clears it: a large marker-less scan that the parallel decoder cannot settle goes back to the caller
  DeviceBuf enc[3];  // uhdr_hip_encode_api1_scans: the six coefficient arrays | base scan | map scan
  DeviceBuf jpg[6];  // uhdr_hip_jpeg_decode_scan: entropy-coded data | coefficient arrays x 3 | decoded planes / pixels
  // uhdr_hip_resident_begin .. _end: the images uhdr_hip_jpeg_decode_scan wrote to the caller's buffers stay on the device,
  // keyed by those host pointers, so that the host variant of uhdr_hip_apply_gainmap does not upload them again
  struct Resident {
    DeviceBuf buf;
    bool valid = false;
    uhdr_img_fmt_t fmt = UHDR_IMG_FMT_UNSPECIFIED;
    unsigned int w = 0, h = 0;          // samples of plane 0 the device copy holds (whole blocks)
    const void* host[3] = {};           // the caller's planes ...
    unsigned int host_stride[3] = {};   // ... and their strides, in samples
    size_t off[3] = {};
    unsigned int dev_stride[3] = {};
    unsigned int prows[3] = {}, pcols[3] = {};  // samples of every plane the device copy holds
    // lazy downloads (uhdr_hip_resident_lazy): the host planes were NOT written, the device copy is the image
    bool host_unwritten = false;
    // uhdr_hip_resident_adopt: a copy_raw_image(this image, adopt_dst) the caller left to the library
    bool adopted = false;
    void* adopt_dst = nullptr;
    unsigned int adopt_stride = 0, adopt_w = 0, adopt_h = 0;
    bool adopt_expand = false;  // RGB888 kept, RGBA8888 wanted (copy_raw_image's conversion, gainmapmath.cpp:1566-1587)
  } resident[2];
  // entropy decode: the subsequence size a scan with this many blocks per MCU settled at after a lost first attempt
  struct HuffHint { uint32_t sub_bits = 0, bits_per_block = 0; } huff_hint[16];
  // entropy decode (round 5): the decode forms of the last DHT set seen stay on the device (every file of one encoder carries the
  // same four tables: building the five forms and uploading 73 KB per call was 40 us of host time), keyed by the DHT bytes
  struct HuffTabCache {
    bool valid = false, fast_ok = false;
    uint8_t key[4 * (17 + 256)] = {};
    DeviceBuf dev;  // HuffDecTable x 4 | HuffFastTable x 8 (symbol form, tracking form) | value form 4 x kHuffValWords words
  } huff_tabs;
  // host -> device staging of large caller-owned (pageable) planes (round 5, fast_h2d): pinned ring + the event of its last copy
  struct PinArena {
    void* p = nullptr;
    size_t cap = 0, off = 0;
    hipEvent_t ev = nullptr;
    bool ev_pending = false;
  } pin;
  uint32_t* h_flags = nullptr;  // pinned: status words of the entropy decoder come back here (a pageable read-back is a staged, blocking copy)
  bool resident_on = false;
  bool resident_lazy = false;
  // a lazily kept image whose write-back failed when its slot was given up (resident_retire cannot return it): latched, and returned
  // by the next entry point that stages an image or flushes -- the call fails instead of going on with an unwritten host buffer
  uhdr_error_info_t sticky = {};
  unsigned int resident_next = 0;
  // an adopted copy that outlived its session (uhdr_hip_resident_end): performed by uhdr_hip_resident_materialize
  struct PendingCopy {
    DeviceBuf buf;
    bool on = false;
    size_t off = 0, pitch = 0, dst_pitch = 0;
    unsigned int w = 0, h = 0, bps = 0;
    bool expand = false;
    void* dst = nullptr;
    DeviceBuf tmp;  // the RGBA8888 form of an RGB888 image on its way out
  } pending;
  // A host-side model of the 256 MiB infinity cache, for one decision: whether applyGainMap's input planes are worth a read
  // sweep by prefetcher workgroups (apply_gainmap.hip).  Reads allocate there, the kernels' nontemporal output stores do not
  // (a frame's inputs are still cached when 102 MB of other frames' inputs were read in between, and are not after 255 MB:
  // bench.py north_star_8k, three / six rotating buffer sets).  So: remember when (in bytes read by this context) a plane was
  // last read, call it hot if less than kMallHotBytes have been read since.
  struct MallEntry { const void* p; uint64_t stamp; };
  std::vector<MallEntry> mall;
  uint64_t mall_clock = 0;
  DeviceBuf minmax;  // 6 + 2048*6 floats
  uint32_t* d_huff = nullptr;     // Annex K code tables (kHuffTabWords) followed by the 64-byte zig-zag map
  CoefSrc* d_coef_src = nullptr;  // apply_gainmap_coef descriptors (rotating slots)
  unsigned int coef_src_next = 0;
  // encode-side step tables (host_tables.cpp): sRGB byte of the tone mapper (one per context), 10-bit code -> linear
  // value per HDR transfer, encodeGain's byte per (min boost, max boost)
  float* d_srgb8 = nullptr;
  StepTab srgb8_meta = {};
  float* d_lin10[5] = {};
  struct GainTab { float mn, mx; float* d; StepTab meta; };
  std::vector<GainTab> gain_tabs;
  // multi-GPU (row stripes): RCCL communicator of this rank + the exchange buffers of two-pass generation
  void* comm = nullptr;          // ncclComm_t
  int comm_rank = 0, comm_size = 0;
  bool comm_custom = false;      // uhdr_hip_comm_init_custom: the exchange steps go through comm_ops instead of RCCL
  uhdr_hip_comm_ops_t comm_ops = {};
  DeviceBuf exchange;            // merged[6] | (unused) | final mm[6]
  DeviceBuf affine;              // AffineDev + pass 2's per-channel step tables (kAffineDevBytes)
  float* d_srgb_of_byte = nullptr;  // 256: byte -> sRGB inverse OETF (the fused API-0 front end)
  float* h_mm = nullptr;         // pinned: the final {min, max} for the metadata fill
  // profiling
  bool prof = false;
  std::vector<ProfEntry> prof_entries;
};

namespace {

uhdr_error_info_t ensure(DeviceBuf& b, size_t bytes) {
  if (b.cap >= bytes) return ok_status();
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + (bytes >> 3) + 256;
  HIP_TRY(hipMalloc(&b.p, want));
  b.cap = want;
  return ok_status();
}

struct ProfScope {
  uhdr_hip_ctx* c;
  ProfEntry e;
  bool on;
  ProfScope(uhdr_hip_ctx* ctx, const char* family) : c(ctx), on(ctx->prof) {
    if (!on) return;
    e.family = family;
    if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) { on = false; return; }
    (void)hipEventRecord(e.a, c->stream);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e.b, c->stream);
    c->prof_entries.push_back(e);
  }
};

size_t bytes_per_sample(int fmt) {
  switch (fmt) {
    case UHDR_IMG_FMT_24bppYCbCrP010:
    case UHDR_IMG_FMT_30bppYCbCr444: return 2;
    case UHDR_IMG_FMT_24bppRGB888: return 3;
    case UHDR_IMG_FMT_32bppRGBA8888:
    case UHDR_IMG_FMT_32bppRGBA1010102: return 4;
    case UHDR_IMG_FMT_64bppRGBAHalfFloat: return 8;
    default: return 1;
  }
}
// rows and row-width (in stride units) of plane `pl`; returns false if the plane does not exist
bool plane_geom(const uhdr_raw_image_t* im, int pl, size_t* rows, size_t* width) {
  const size_t w = im->w, h = im->h;
  switch (im->fmt) {
    case UHDR_IMG_FMT_24bppYCbCrP010:
      if (pl == 0) { *rows = h; *width = w; return true; }
      if (pl == 1) { *rows = (h + 1) / 2; *width = ((w + 1) / 2) * 2; return true; }
      return false;
    case UHDR_IMG_FMT_12bppYCbCr420:
      if (pl == 0) { *rows = h; *width = w; } else { *rows = (h + 1) / 2; *width = (w + 1) / 2; }
      return true;
    case UHDR_IMG_FMT_16bppYCbCr422:
      if (pl == 0) { *rows = h; *width = w; } else { *rows = h; *width = (w + 1) / 2; }
      return true;
    case UHDR_IMG_FMT_24bppYCbCr444:
    case UHDR_IMG_FMT_30bppYCbCr444:
      *rows = h; *width = w; return true;
    default:
      if (pl == 0) { *rows = h; *width = w; return true; }
      return false;
  }
}
size_t plane_bytes(const uhdr_raw_image_t* im, int pl) {
  size_t rows, width;
  if (!plane_geom(im, pl, &rows, &width) || rows == 0) return 0;
  return ((rows - 1) * (size_t)im->stride[pl] + width) * bytes_per_sample(im->fmt);
}

// Shared descriptor check of the *_dev entry points (the reference allocates these images itself, a C ABI caller
// fills them by hand): every plane the format has must be non-null and every stride must cover the plane's row.
uhdr_error_info_t validate_image(const uhdr_raw_image_t* im, const char* what) {
  if (!im) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for %s image descriptor", what);
  if (im->w == 0 || im->h == 0)
    return err_status(UHDR_CODEC_INVALID_PARAM, "%s image dimensions cannot be zero, received %ux%u", what, im->w, im->h);
  for (int pl = 0; pl < 3; pl++) {
    size_t rows, width;
    if (!plane_geom(im, pl, &rows, &width)) continue;
    if (!im->planes[pl])
      return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for plane %d of %s image (format %d)", pl, what, im->fmt);
    if ((size_t)im->stride[pl] < width)
      return err_status(UHDR_CODEC_INVALID_PARAM, "%s image: stride %u of plane %d is less than its row width %zu", what,
                        im->stride[pl], pl, width);
  }
  return ok_status();
}

ImageView view_of(const uhdr_raw_image_t* im) {
  ImageView v;
  for (int i = 0; i < 3; i++) { v.p[i] = im->planes[i]; v.stride[i] = im->stride[i]; }
  v.w = im->w; v.h = im->h; v.fmt = im->fmt; v.range = im->range;
  return v;
}
ImageViewMut view_mut_of(const uhdr_raw_image_t* im) {
  ImageViewMut v;
  for (int i = 0; i < 3; i++) { v.p[i] = im->planes[i]; v.stride[i] = im->stride[i]; }
  v.w = im->w; v.h = im->h; v.fmt = im->fmt; v.range = im->range;
  return v;
}

// Stage a host image into device scratch `slot` (all planes packed back to back, 256-B aligned);
// *dev gets device plane pointers.  upload=false only reserves space (outputs).
// a host buffer is about to be (re)written by the library: whatever device copy was kept for it is stale
// Lazy downloads.  write_back: bring the host planes of a kept image up to date (they were left unwritten) and perform the
// copy the caller left to the library, both from the device copy; the entry stays valid.
// device image -> host destination of an adopted copy; expand: RGB888 -> RGBA8888 with alpha 255 on the way
uhdr_error_info_t adopted_copy_out(uhdr_hip_ctx* c, const void* src, size_t src_pitch, unsigned int bps, bool expand, unsigned int w, unsigned int h,
                                   void* dst, size_t dst_pitch) {
  if (expand) {
    UHDR_TRY(ensure(c->pending.tmp, (size_t)w * 4 * h));
    HIP_TRY(launch_repack(0, src, src_pitch, c->pending.tmp.p, (size_t)w * 4, w, h, c->stream));
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, c->pending.tmp.p, (size_t)w * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost, c->stream));
  } else {
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, (size_t)w * bps, h, hipMemcpyDeviceToHost, c->stream));
  }
  return ok_status();
}
uhdr_error_info_t resident_write_back(uhdr_hip_ctx* c, uhdr_hip_ctx::Resident& r) {
  if (!r.valid || (!r.host_unwritten && !r.adopted)) return ok_status();
  const size_t bps = bytes_per_sample(r.fmt);
  if (r.host_unwritten)
    for (int pl = 0; pl < 3; pl++) {
      if (!r.host[pl]) continue;
      HIP_TRY(hipMemcpy2DAsync((void*)r.host[pl], (size_t)r.host_stride[pl] * bps, (const char*)r.buf.p + r.off[pl], (size_t)r.dev_stride[pl] * bps,
                               (size_t)r.pcols[pl] * bps, r.prows[pl], hipMemcpyDeviceToHost, c->stream));
    }
  if (r.adopted)
    UHDR_TRY(adopted_copy_out(c, (const char*)r.buf.p + r.off[0], (size_t)r.dev_stride[0] * bps, (unsigned int)bps, r.adopt_expand, r.adopt_w, r.adopt_h,
                              r.adopt_dst, (size_t)r.adopt_stride * (r.adopt_expand ? 4 : bps)));
  HIP_TRY(hipStreamSynchronize(c->stream));
  r.host_unwritten = false;
  r.adopted = false;
  c->stats.lazy_downloads_done++;
  return ok_status();
}
uhdr_error_info_t resident_write_back_all(uhdr_hip_ctx* c) {
  for (auto& r : c->resident) UHDR_TRY(resident_write_back(c, r));
  return ok_status();
}
// a kept image is given up (its slot is needed, or its host buffer is about to be rewritten by the library)
void resident_retire(uhdr_hip_ctx* c, uhdr_hip_ctx::Resident& r, bool host_is_rewritten) {
  if (r.valid && (r.adopted || (r.host_unwritten && !host_is_rewritten))) {
    if (host_is_rewritten) r.host_unwritten = false;
    const uhdr_error_info_t wb = resident_write_back(c, r);
    if (wb.error_code != UHDR_CODEC_OK && c->sticky.error_code == UHDR_CODEC_OK) {
      c->sticky = wb;
      fprintf(stderr, "uhdr_hip: write-back of a device-resident image failed (%s): the next call on this context reports it\n", wb.has_detail ? wb.detail : "");
    }
  }
  r.valid = false;
  r.host_unwritten = false;
  r.adopted = false;
}
void resident_drop(uhdr_hip_ctx* c, const void* host_plane) {  // any plane of a kept image
  if (!host_plane) return;
  for (auto& r : c->resident)
    if (r.valid && (r.host[0] == host_plane || r.host[1] == host_plane || r.host[2] == host_plane)) resident_retire(c, r, true);
}

// Host -> device copy of a large caller-owned buffer.  hipMemcpyAsync from pageable memory is staged by the runtime on the
// calling thread: one core's memcpy into its bounce buffers, ~7 GB/s end to end (37 MB of P010 + 4:2:0 planes: 5.3 of the 6.1 ms
// of a 4K uhdr_encode, profiles/r04_api_trace.txt).  Here a few threads copy 1 MiB pieces into a pinned ring while the calling
// thread hands every finished run of pieces to the DMA engine: the link, not a core, sets the pace.  The reference itself runs
// its per-pixel loops on up to four threads (JobQueue users, jpegr.cpp:845-864); so does this.  UHDR_HIP_UPLOAD_THREADS=0: the
// runtime's path.
uhdr_error_info_t fast_h2d(uhdr_hip_ctx* c, void* dst, const void* src, size_t bytes) {
  static const int nthreads = [] {
    const char* e = getenv("UHDR_HIP_UPLOAD_THREADS");
    int v = e ? atoi(e) : 4;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && (unsigned)v > hw) v = (int)hw;
    return v < 0 ? 0 : (v > 16 ? 16 : v);
  }();
  constexpr size_t kPiece = (size_t)1 << 20;
  if (nthreads == 0 || bytes < 4 * kPiece) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return ok_status();
  }
  uhdr_hip_ctx::PinArena& pa = c->pin;
  const size_t need = (bytes + 4095) & ~(size_t)4095;
  if (!pa.ev) HIP_TRY(hipEventCreateWithFlags(&pa.ev, hipEventDisableTiming));
  if (pa.cap < need) {
    if (pa.ev_pending) { HIP_TRY(hipEventSynchronize(pa.ev)); pa.ev_pending = false; }
    if (pa.p) (void)hipHostFree(pa.p);
    pa.p = nullptr;
    pa.cap = pa.off = 0;
    size_t want = need + need / 2;
    if (want < ((size_t)64 << 20)) want = (size_t)64 << 20;
    HIP_TRY(hipHostMalloc(&pa.p, want, hipHostMallocDefault));
    pa.cap = want;
  }
  if (pa.off + need > pa.cap) {  // wrap: everything copied out of the ring so far must have left it
    if (pa.ev_pending) { HIP_TRY(hipEventSynchronize(pa.ev)); pa.ev_pending = false; }
    pa.off = 0;
  }
  uint8_t* stage = (uint8_t*)pa.p + pa.off;
  const size_t npieces = (bytes + kPiece - 1) / kPiece;
  std::unique_ptr<std::atomic<unsigned char>[]> done(new std::atomic<unsigned char>[npieces]);
  for (size_t i = 0; i < npieces; i++) done[i].store(0, std::memory_order_relaxed);
  std::atomic<size_t> next{0};
  auto work = [&]() {
    for (;;) {
      const size_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= npieces) return;
      const size_t o = i * kPiece, n = o + kPiece <= bytes ? kPiece : bytes - o;
      memcpy(stage + o, (const uint8_t*)src + o, n);
      done[i].store(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  const int nt = (size_t)nthreads < npieces ? nthreads : (int)npieces;
  pool.reserve((size_t)nt);
  for (int t = 0; t < nt; t++) pool.emplace_back(work);
  hipError_t err = hipSuccess;
  size_t i = 0;
  while (i < npieces) {
    while (!done[i].load(std::memory_order_acquire)) std::this_thread::yield();
    size_t j = i + 1;
    while (j < npieces && j - i < 8 && done[j].load(std::memory_order_acquire)) j++;
    const size_t o = i * kPiece, n = (j * kPiece <= bytes ? j * kPiece : bytes) - o;
    if (err == hipSuccess) err = hipMemcpyAsync((uint8_t*)dst + o, stage + o, n, hipMemcpyHostToDevice, c->stream);
    i = j;
  }
  for (auto& t : pool) t.join();
  HIP_TRY(err);
  HIP_TRY(hipEventRecord(pa.ev, c->stream));
  pa.ev_pending = true;
  pa.off += need;
  return ok_status();
}

uhdr_error_info_t stage_in(uhdr_hip_ctx* c, int slot, const uhdr_raw_image_t* host, uhdr_raw_image_t* dev,
                           bool upload) {
  if (c->sticky.error_code != UHDR_CODEC_OK) {  // see uhdr_hip_ctx::sticky
    const uhdr_error_info_t e = c->sticky;
    c->sticky = ok_status();
    return e;
  }
  if (!upload && c->resident_on) resident_drop(c, host->planes[0]);  // an output: stage_out will overwrite the host planes
  if (upload && c->resident_on) {  // an image uhdr_hip_jpeg_decode_scan wrote in this session is still on the device
    for (auto& r : c->resident) {
      if (!r.valid || r.fmt != host->fmt || host->w > r.w || host->h > r.h) continue;
      bool same = true;
      for (int pl = 0; pl < 3; pl++) {
        const bool has = plane_bytes(host, pl) != 0;
        same = same && (has ? host->planes[pl] == r.host[pl] && host->stride[pl] == r.host_stride[pl] : r.host[pl] == nullptr);
      }
      if (!same) continue;
      *dev = *host;
      for (int pl = 0; pl < 3; pl++) {
        dev->planes[pl] = r.host[pl] ? (char*)r.buf.p + r.off[pl] : nullptr;
        dev->stride[pl] = r.dev_stride[pl];
      }
      c->stats.resident_hits++;
      return ok_status();
    }
  }
  if (upload && c->resident_on) UHDR_TRY(resident_write_back_all(c));  // host planes are read below: none may be a lazily kept image's
  size_t off[3] = {0, 0, 0}, total = 0;
  for (int pl = 0; pl < 3; pl++) {
    size_t b = plane_bytes(host, pl);
    if (b && !host->planes[pl])
      return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for plane %d of image format %d", pl, host->fmt);
    off[pl] = total;
    total += (b + 255) & ~(size_t)255;
  }
  UHDR_TRY(ensure(c->scratch[slot], total ? total : 256));
  *dev = *host;
  for (int pl = 0; pl < 3; pl++) {
    size_t b = plane_bytes(host, pl);
    dev->planes[pl] = b ? (char*)c->scratch[slot].p + off[pl] : nullptr;
    if (b && upload) UHDR_TRY(fast_h2d(c, dev->planes[pl], host->planes[pl], b));
  }
  return ok_status();
}
// Inside a resident session an 8-bit image the library has just produced (gain map, tone-mapped / converted base image)
// stays on the device as well, keyed by the host planes it is being copied to: JpegR::encodeJPEGR hands exactly those
// planes to JpegEncoderHelper::compressImage next (jpegr.cpp:253-316), and uhdr_hip_jpeg_encode_scan then reads the
// device copy instead of uploading what was downloaded a moment ago.  One device-to-device copy (25 MB: ~10 us).
bool resident_keeps(int fmt) {
  switch (fmt) {
    case UHDR_IMG_FMT_12bppYCbCr420: case UHDR_IMG_FMT_16bppYCbCr422: case UHDR_IMG_FMT_24bppYCbCr444: case UHDR_IMG_FMT_8bppYCbCr400:
    case UHDR_IMG_FMT_24bppRGB888: case UHDR_IMG_FMT_32bppRGBA8888: return true;
    default: return false;
  }
}
// host_unwritten: the caller did NOT copy the image to the host planes (lazy downloads): the device copy is the image
uhdr_error_info_t resident_keep(uhdr_hip_ctx* c, const uhdr_raw_image_t* dev, const uhdr_raw_image_t* host, bool host_unwritten = false) {
  if (!resident_keeps(host->fmt)) return ok_status();
  resident_drop(c, host->planes[0]);
  unsigned int slot = c->resident_next++ % 2;
  {  // an in-place operator may have worked ON a resident copy: that buffer is the source of the copy below, take the other one
    const DeviceBuf& b = c->resident[slot].buf;
    const char* d0 = (const char*)dev->planes[0];
    if (b.p && d0 >= (const char*)b.p && d0 < (const char*)b.p + b.cap) slot ^= 1u;
  }
  uhdr_hip_ctx::Resident& r = c->resident[slot];
  resident_retire(c, r, false);
  const DeviceBuf keep = r.buf;
  r = uhdr_hip_ctx::Resident();
  r.buf = keep;
  size_t total = 0, bytes[3] = {0, 0, 0};
  for (int pl = 0; pl < 3; pl++) {
    bytes[pl] = host->planes[pl] ? plane_bytes(host, pl) : 0;
    r.off[pl] = total;
    total += (bytes[pl] + 255) & ~(size_t)255;
  }
  if (total == 0) return ok_status();
  UHDR_TRY(ensure(r.buf, total));
  const size_t bps = bytes_per_sample(host->fmt);
  for (int pl = 0; pl < 3; pl++) {
    size_t rows = 0, width = 0;
    if (!bytes[pl] || !plane_geom(host, pl, &rows, &width)) continue;
    const size_t pitch = (size_t)host->stride[pl] * bps, dpitch = (size_t)dev->stride[pl] * bps;
    HIP_TRY(hipMemcpy2DAsync((char*)r.buf.p + r.off[pl], pitch, dev->planes[pl], dpitch, width * bps, rows, hipMemcpyDeviceToDevice, c->stream));
    r.host[pl] = host->planes[pl];
    r.host_stride[pl] = host->stride[pl];
    r.dev_stride[pl] = host->stride[pl];
    r.prows[pl] = (unsigned int)rows;
    r.pcols[pl] = (unsigned int)width;
  }
  r.fmt = host->fmt;
  r.w = host->w;
  r.h = host->h;
  r.valid = true;
  r.host_unwritten = host_unwritten;
  return ok_status();
}
// Copies back only the w samples of every row, so the caller's stride padding stays untouched
// (the reference never writes there either).  Lazy downloads (uhdr_hip_resident_lazy): an image the handoff keeps is not copied
// back at all -- the generated gain map of an encode, whose only reader is the compressImage that follows (jpegr.cpp:253-257)
// and finds it on the device; whoever else would read the host planes gets them written first (resident_write_back).
uhdr_error_info_t stage_out(uhdr_hip_ctx* c, const uhdr_raw_image_t* dev, uhdr_raw_image_t* host) {
  const bool lazy = c->resident_on && c->resident_lazy && resident_keeps(host->fmt) && host->planes[0];
  if (lazy) {
    UHDR_TRY(resident_keep(c, dev, host, true));
    c->stats.lazy_downloads_skipped++;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return ok_status();
  }
  for (int pl = 0; pl < 3; pl++) {
    size_t rows, width;
    if (!plane_geom(host, pl, &rows, &width) || rows == 0 || !host->planes[pl]) continue;
    const size_t bps = bytes_per_sample(host->fmt);
    const size_t pitch = (size_t)host->stride[pl] * bps;
    const size_t dpitch = (size_t)dev->stride[pl] * bps;  // differs from the host's only for a device-resident copy (stage_in)
    if (dpitch == pitch && (rows == 1 || pitch == width * bps)) {
      HIP_TRY(hipMemcpyAsync(host->planes[pl], dev->planes[pl], ((rows - 1) * (size_t)host->stride[pl] + width) * bps,
                             hipMemcpyDeviceToHost, c->stream));
    } else {
      HIP_TRY(hipMemcpy2DAsync(host->planes[pl], pitch, dev->planes[pl], dpitch, width * bps, rows,
                               hipMemcpyDeviceToHost, c->stream));
    }
  }
  if (c->resident_on) UHDR_TRY(resident_keep(c, dev, host));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

uhdr_error_info_t upload_lut(float** dst, const std::vector<float>& src, hipStream_t s);
uhdr_error_info_t upload_math(uhdr_hip_ctx* c);
// Linearisation table of an HDR input for the encode kernels: inverse OETF with, for HLG, the
// per-channel OOTF (hlgOotfApprox, gainmapmath.cpp:293-295) folded in node by node.
uhdr_error_info_t select_hdr_lut(uhdr_hip_ctx* c, uhdr_color_transfer_t ct, const float** lut, int* n);

uhdr_error_info_t upload_lut(float** dst, const std::vector<float>& src, hipStream_t s) {
  if (*dst) return ok_status();
  HIP_TRY(hipMalloc((void**)dst, src.size() * sizeof(float)));
  HIP_TRY(hipMemcpyAsync(*dst, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice, s));
  HIP_TRY(hipStreamSynchronize(s));
  return ok_status();
}
uhdr_error_info_t upload_math(uhdr_hip_ctx* c) {
  if (c->d_math) return ok_status();
  const std::vector<double>& t = host::math_tables();
  HIP_TRY(hipMalloc((void**)&c->d_math, t.size() * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(c->d_math, t.data(), t.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}
// a verified host step table -> device copy + the kernel-side descriptor (tab == nullptr when the table is not exact)
static uhdr_error_info_t upload_step_table(const host::OetfBuckets& b, float** slot, StepTab* meta, hipStream_t s) {
  memset(meta, 0, sizeof *meta);
  if (!b.exact || b.n == 0 || b.n > (uint32_t)kStepTabMax) return ok_status();
  if (!*slot) {
    std::vector<float> raw(b.entries.size());
    memcpy(raw.data(), b.entries.data(), raw.size() * sizeof(float));
    UHDR_TRY(upload_lut(slot, raw, s));
  }
  meta->tab = (const uint2*)*slot;
  meta->n = b.n;
  meta->base8 = b.base * 8;
  meta->shm3 = b.shift - 3;
  meta->lo_bits = b.clamp_lo_bits;
  meta->hi_bits = b.hi_bits;
  return ok_status();
}
// encodeGain's byte for the clamped gain (one pass, gamma 1), cached per boost range
static uhdr_error_info_t gain_step_table(uhdr_hip_ctx* c, const GenParams& p, StepTab* out) {
  memset(out, 0, sizeof *out);
  if (p.gamma != 1.0f) return ok_status();
  for (auto& g : c->gain_tabs)
    if (g.mn == p.min_boost && g.mx == p.max_boost) { *out = g.meta; return ok_status(); }
  if (c->gain_tabs.size() >= 16) return ok_status();  // a caller cycling through boost ranges keeps the float64 evaluation
  uhdr_hip_ctx::GainTab g;
  g.mn = p.min_boost; g.mx = p.max_boost; g.d = nullptr;
  const host::OetfBuckets b = host::gain_code8_buckets(p.min_boost, p.max_boost, p.log2min, p.log2_range, p.log2_range_rcp);
  UHDR_TRY(upload_step_table(b, &g.d, &g.meta, c->stream));
  c->gain_tabs.push_back(g);
  *out = g.meta;
  return ok_status();
}

uhdr_error_info_t select_hdr_lut(uhdr_hip_ctx* c, uhdr_color_transfer_t ct, const float** lut, int* n) {
  *lut = nullptr;
  *n = 0;
  if (ct == UHDR_CT_HLG) {
    UHDR_TRY(upload_lut(&c->d_hlg_inv_ootf, host::hlg_inv_oetf_ootf_lut(), c->stream));
    *lut = c->d_hlg_inv_ootf; *n = kInvOetfN;
  } else if (ct == UHDR_CT_PQ) {
    UHDR_TRY(upload_lut(&c->d_pq_inv, host::pq_inv_oetf_lut(), c->stream));
    *lut = c->d_pq_inv; *n = kInvOetfN;
  } else if (ct == UHDR_CT_SRGB) {
    *lut = c->d_srgb; *n = kSrgbN;
  }
  return ok_status();
}

// uhdr_validate_gainmap_metadata_descriptor (ultrahdr_api.cpp:431-503)
uhdr_error_info_t validate_metadata(const uhdr_gainmap_metadata_t* m) {
  if (!m) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for gainmap metadata descriptor");
  uhdr_error_info_t st = ok_status();
  for (int i = 0; i < 3; i++) {
    if (!std::isfinite(m->min_content_boost[i]) || !std::isfinite(m->max_content_boost[i]) ||
        !std::isfinite(m->offset_sdr[i]) || !std::isfinite(m->offset_hdr[i]) ||
        !std::isfinite(m->hdr_capacity_min) || !std::isfinite(m->hdr_capacity_max) || !std::isfinite(m->gamma[i])) {
      st = err_status(UHDR_CODEC_INVALID_PARAM, "Field(s) of gainmap metadata descriptor are either NaN or infinite");
    } else if (m->max_content_boost[i] < m->min_content_boost[i]) {
      st = err_status(UHDR_CODEC_INVALID_PARAM, "received bad value for content boost max %f, expects to be >= content boost min %f",
                      m->max_content_boost[i], m->min_content_boost[i]);
    } else if (m->min_content_boost[i] <= 0.0f) {
      return err_status(UHDR_CODEC_INVALID_PARAM, "received bad value for min boost %f, expects > 0.0f", m->min_content_boost[i]);
    } else if (m->gamma[i] <= 0.0f) {
      st = err_status(UHDR_CODEC_INVALID_PARAM, "received bad value for gamma %f, expects > 0.0f", m->gamma[i]);
    } else if (m->offset_sdr[i] < 0.0f) {
      st = err_status(UHDR_CODEC_INVALID_PARAM, "received bad value for offset sdr %f, expects to be >= 0.0f", m->offset_sdr[i]);
    } else if (m->offset_hdr[i] < 0.0f) {
      st = err_status(UHDR_CODEC_INVALID_PARAM, "received bad value for offset hdr %f, expects to be >= 0.0f", m->offset_hdr[i]);
    } else if (m->hdr_capacity_max <= m->hdr_capacity_min) {
      st = err_status(UHDR_CODEC_INVALID_PARAM, "received bad value for hdr capacity max %f, expects to be > hdr capacity min %f",
                      m->hdr_capacity_max, m->hdr_capacity_min);
    } else if (m->hdr_capacity_min < 1.0f) {
      st = err_status(UHDR_CODEC_INVALID_PARAM, "received bad value for hdr capacity min %f, expects to be >= 1.0f", m->hdr_capacity_min);
    }
  }
  return st;
}

bool is_rgb_fmt_host(int fmt) {
  return fmt == UHDR_IMG_FMT_64bppRGBAHalfFloat || fmt == UHDR_IMG_FMT_32bppRGBA8888 ||
         fmt == UHDR_IMG_FMT_32bppRGBA1010102;
}

// argument checks of UltraHdr::applyGainMap (jpegr.cpp:1538-1614), in the reference's order
uhdr_error_info_t validate_apply(const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* gm,
                                 const uhdr_gainmap_metadata_t* md, uhdr_color_transfer_t out_ct,
                                 const uhdr_raw_image_t* dest) {
  if (dest == nullptr || dest->planes[UHDR_PLANE_PACKED] == nullptr)
    return err_status(UHDR_CODEC_INVALID_PARAM, "apply gainmap method received nullptr for destination image or plane pointer");
  if (dest->stride[UHDR_PLANE_PACKED] < dest->w)
    return err_status(UHDR_CODEC_INVALID_PARAM, "destination stride (%u) cannot be less than image width (%u)",
                      dest->stride[UHDR_PLANE_PACKED], dest->w);
  if (out_ct != UHDR_CT_LINEAR && out_ct != UHDR_CT_HLG && out_ct != UHDR_CT_PQ)
    return err_status(UHDR_CODEC_INVALID_PARAM,
                      "apply gainmap method expects output color transfer to be one of {UHDR_CT_LINEAR, UHDR_CT_HLG, UHDR_CT_PQ}. Received %d", out_ct);
  if ((out_ct == UHDR_CT_LINEAR && dest->fmt != UHDR_IMG_FMT_64bppRGBAHalfFloat) ||
      ((out_ct == UHDR_CT_HLG || out_ct == UHDR_CT_PQ) && dest->fmt != UHDR_IMG_FMT_32bppRGBA1010102))
    return err_status(UHDR_CODEC_INVALID_PARAM, "unsupported destination pixel format %d for output color transfer %d", dest->fmt, out_ct);
  UHDR_TRY(validate_metadata(md));
  if (!sdr || !gm) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for base image or gainmap image");
  if (sdr->fmt != UHDR_IMG_FMT_24bppYCbCr444 && sdr->fmt != UHDR_IMG_FMT_16bppYCbCr422 &&
      sdr->fmt != UHDR_IMG_FMT_12bppYCbCr420 && sdr->fmt != UHDR_IMG_FMT_24bppRGB888 &&
      sdr->fmt != UHDR_IMG_FMT_32bppRGBA8888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "apply gainmap method expects base image color format to be one of "
                      "{UHDR_IMG_FMT_24bppYCbCr444, UHDR_IMG_FMT_16bppYCbCr422, UHDR_IMG_FMT_12bppYCbCr420, "
                      "UHDR_IMG_FMT_24bppRGB888, UHDR_IMG_FMT_32bppRGBA8888}. Received %d", sdr->fmt);
  if (gm->fmt != UHDR_IMG_FMT_8bppYCbCr400 && gm->fmt != UHDR_IMG_FMT_24bppRGB888 && gm->fmt != UHDR_IMG_FMT_32bppRGBA8888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "apply gainmap method expects gainmap image color format to be one of "
                      "{UHDR_IMG_FMT_8bppYCbCr400, UHDR_IMG_FMT_24bppRGB888, UHDR_IMG_FMT_32bppRGBA8888}. Received %d", gm->fmt);
  return ok_status();
}

// acquire a table slot holding the ApplyTables block for (metadata, weight, scale)
uhdr_error_info_t get_apply_tables(uhdr_hip_ctx* c, const uhdr_gainmap_metadata_t& md, float weight,
                                   int idw_scale, const float** d_out) {
  std::string key((const char*)&md, sizeof md);
  key.append((const char*)&weight, sizeof weight);
  key.append((const char*)&idw_scale, sizeof idw_scale);
  for (int i = 0; i < kTableSlots; i++)
    if (c->tab_cap[i] && c->tab_key[i] == key) { *d_out = c->d_tab[i]; return ok_status(); }
  const int slot = c->tab_next;
  c->tab_next = (c->tab_next + 1) % kTableSlots;
  std::vector<float> t;
  host::build_apply_tables(md, weight, idw_scale, &t);
  const size_t bytes = t.size() * sizeof(float);
  if (c->tab_cap[slot]) HIP_TRY(hipEventSynchronize(c->tab_ev[slot]));  // previous upload from this slot done
  if (c->tab_cap[slot] < bytes) {
    // a kernel may still be reading the old device block: drain the stream before freeing it
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->h_tab[slot]) (void)hipHostFree(c->h_tab[slot]);
    if (c->d_tab[slot]) (void)hipFree(c->d_tab[slot]);
    c->h_tab[slot] = nullptr; c->d_tab[slot] = nullptr; c->tab_cap[slot] = 0;
    HIP_TRY(hipHostMalloc((void**)&c->h_tab[slot], bytes, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&c->d_tab[slot], bytes));
    if (!c->tab_ev[slot]) HIP_TRY(hipEventCreateWithFlags(&c->tab_ev[slot], hipEventDisableTiming));
    c->tab_cap[slot] = bytes;
  }
  memcpy(c->h_tab[slot], t.data(), bytes);
  HIP_TRY(hipMemcpyAsync(c->d_tab[slot], c->h_tab[slot], bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipEventRecord(c->tab_ev[slot], c->stream));
  c->tab_key[slot] = key;
  *d_out = c->d_tab[slot];
  return ok_status();
}

}  // namespace

// -------------------------------------------------------------------------------------------------
// context
// -------------------------------------------------------------------------------------------------
#pragma GCC visibility push(default)
extern "C" {

const char* uhdr_hip_version(void) { return "libuhdr_hip 0.3 (gfx950; reference libultrahdr 2.0.2 hot path)"; }

int uhdr_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

uhdr_hip_ctx_t* uhdr_hip_create(int device, uhdr_error_info_t* err) {
  auto fail = [&](const char* what, hipError_t e) -> uhdr_hip_ctx_t* {
    if (err) *err = err_status(UHDR_CODEC_ERROR, "uhdr_hip_create: %s failed: %s (no CPU fallback exists behind this library)",
                               what, hipGetErrorString(e));
    return nullptr;
  };
  if (err) *err = ok_status();
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) return fail("hipGetDeviceCount", e == hipSuccess ? hipErrorNoDevice : e);
  if (device < 0) {
    e = hipGetDevice(&device);
    if (e != hipSuccess) return fail("hipGetDevice", e);
  }
  if (device >= n) return fail("device index", hipErrorInvalidDevice);
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail("hipSetDevice", e);
  uhdr_hip_ctx* c = new uhdr_hip_ctx();
  c->device = device;
  e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) { delete c; return fail("hipStreamCreate", e); }
  c->stream = c->own_stream;
  uhdr_error_info_t st = upload_lut(&c->d_srgb, host::srgb_inv_oetf_lut(), c->stream);
  if (st.error_code != UHDR_CODEC_OK) {
    if (err) *err = st;
    (void)hipStreamDestroy(c->own_stream);
    delete c;
    return nullptr;
  }
  return c;
}

void uhdr_hip_destroy(uhdr_hip_ctx_t* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (auto& e : c->prof_entries) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  float* luts[] = {c->d_srgb, c->d_hlg_inv, c->d_pq_inv, c->d_hlg_oetf, c->d_pq_oetf, c->d_hlg_inv_ootf, (float*)c->d_math,
                   c->d_hlg_buckets, c->d_pq_buckets, c->d_hlg_buckets_pre, c->d_pq_buckets_pre};
  for (float* p : luts) if (p) (void)hipFree(p);
  for (int i = 0; i < kTableSlots; i++) {
    if (c->h_tab[i]) (void)hipHostFree(c->h_tab[i]);
    if (c->d_tab[i]) (void)hipFree(c->d_tab[i]);
    if (c->tab_ev[i]) (void)hipEventDestroy(c->tab_ev[i]);
  }
  for (auto& b : c->scratch) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->jpg) if (b.p) (void)hipFree(b.p);
  for (auto& b : c->enc) if (b.p) (void)hipFree(b.p);
  for (auto& r : c->resident) if (r.buf.p) (void)hipFree(r.buf.p);
  if (c->pending.buf.p) (void)hipFree(c->pending.buf.p);
  if (c->pending.tmp.p) (void)hipFree(c->pending.tmp.p);
  if (c->minmax.p) (void)hipFree(c->minmax.p);
  if (c->d_coef_src) (void)hipFree(c->d_coef_src);
  if (c->d_huff) (void)hipFree(c->d_huff);
  if (c->d_srgb8) (void)hipFree(c->d_srgb8);
  for (float* q : c->d_lin10) if (q) (void)hipFree(q);
  for (auto& g : c->gain_tabs) if (g.d) (void)hipFree(g.d);
  uhdr_hip_comm_destroy(c);
  if (c->exchange.p) (void)hipFree(c->exchange.p);
  if (c->affine.p) (void)hipFree(c->affine.p);
  if (c->d_srgb_of_byte) (void)hipFree(c->d_srgb_of_byte);
  if (c->h_mm) (void)hipHostFree(c->h_mm);
  if (c->h_flags) (void)hipHostFree(c->h_flags);
  if (c->pin.p) (void)hipHostFree(c->pin.p);
  if (c->pin.ev) (void)hipEventDestroy(c->pin.ev);
  if (c->huff_tabs.dev.p) (void)hipFree(c->huff_tabs.dev.p);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

uhdr_error_info_t uhdr_hip_set_stream(uhdr_hip_ctx_t* c, void* hip_stream) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return ok_status();
}

void* uhdr_hip_get_stream(uhdr_hip_ctx_t* c) { return c ? (void*)c->stream : nullptr; }

uhdr_error_info_t uhdr_hip_synchronize(uhdr_hip_ctx_t* c) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

void uhdr_hip_get_stats(uhdr_hip_ctx_t* c, uhdr_hip_stats_t* out) {
  if (!out) return;
  if (!c) { memset(out, 0, sizeof *out); return; }
  *out = c->stats;
}

void uhdr_hip_profile_enable(uhdr_hip_ctx_t* c, int enable) {
  if (c) c->prof = enable != 0;
}

int uhdr_hip_profile_read(uhdr_hip_ctx_t* c, const char* family, double* total_ms, int reset) {
  if (!c) return 0;
  (void)hipStreamSynchronize(c->stream);
  int n = 0;
  double tot = 0.0;
  std::vector<ProfEntry> keep;
  for (auto& e : c->prof_entries) {
    const bool match = !family || e.family == family;
    if (match) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { tot += ms; n++; }
    }
    if (match && reset) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    else keep.push_back(e);
  }
  if (reset) c->prof_entries.swap(keep);
  if (total_ms) *total_ms = tot;
  return n;
}

void uhdr_hip_profile_mark(uhdr_hip_ctx_t* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  launch_profile_mark(c->stream);
}

int uhdr_hip_profile_read_list(uhdr_hip_ctx_t* c, const char* family, double* ms_out, int capacity, int reset) {
  if (!c) return 0;
  (void)hipStreamSynchronize(c->stream);
  int n = 0;
  std::vector<ProfEntry> keep;
  for (auto& e : c->prof_entries) {
    const bool match = !family || e.family == family;
    if (match) {
      float ms = 0.0f;
      if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
        if (ms_out && n < capacity) ms_out[n] = ms;
        n++;
      }
    }
    if (match && reset) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    else keep.push_back(e);
  }
  if (reset) c->prof_entries.swap(keep);
  return n;
}

// -------------------------------------------------------------------------------------------------
// applyGainMap
// -------------------------------------------------------------------------------------------------
static uhdr_error_info_t build_apply_params(uhdr_hip_ctx* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* gm,
                                            const uhdr_gainmap_metadata_t* md, uhdr_color_transfer_t out_ct,
                                            float max_display_boost, uhdr_raw_image_t* dest, unsigned int y0,
                                            unsigned int full_height, ApplyParams* out) {
  UHDR_TRY(validate_apply(sdr, gm, md, out_ct, dest));
  // colour-space bookkeeping (jpegr.cpp:1616-1631)
  const int sdr_cg = sdr->cg == UHDR_CG_UNSPECIFIED ? UHDR_CG_BT_709 : sdr->cg;
  const int hdr_cg = gm->cg == UHDR_CG_UNSPECIFIED ? sdr_cg : gm->cg;
  dest->cg = (uhdr_color_gamut_t)hdr_cg;
  ApplyParams& p = *out;
  memset(&p, 0, sizeof p);
  bool identity = false;
  if (!host::gamut_matrix(hdr_cg, sdr_cg, &p.gamut, &identity))
    return err_status(UHDR_CODEC_ERROR, "No implementation available for converting from gamut %d to %d", sdr_cg, hdr_cg);
  p.hdr_gamut_on = (md->use_base_cg && !identity) ? 1 : 0;
  p.sdr_gamut_on = (!md->use_base_cg && !identity) ? 1 : 0;

  if (gm->w == 0 || gm->h == 0 || sdr->w == 0 || sdr->h == 0)
    return err_status(UHDR_CODEC_INVALID_PARAM, "received image with zero width or height");
  // aspect-ratio guard (jpegr.cpp:1651-1671) on the WHOLE image's height
  const unsigned int whole_h = full_height ? full_height : sdr->h;
  if (full_height == 0 && y0 != 0)
    return err_status(UHDR_CODEC_INVALID_PARAM, "stripe offset y0=%u given without the full image height", y0);
  if ((uint64_t)y0 + sdr->h > whole_h)
    return err_status(UHDR_CODEC_INVALID_PARAM, "stripe rows [%u, %u) exceed the full image height %u", y0, y0 + sdr->h, whole_h);
  {
    const float pa = (float)sdr->w / whole_h, ga = (float)gm->w / gm->h;
    if (fabsf(pa - ga) / pa > 0.01f)
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE,
                        "gain map aspect ratio differs from the base image (%ux%u vs %ux%u): the reference's "
                        "resize_image fallback (jpegr.cpp:1659) is outside the HIP hot path",
                        gm->w, gm->h, sdr->w, whole_h);
  }
  const float msf = (float)sdr->w / gm->w;
  int msf_rnd = (int)roundf(msf);
  if (msf_rnd < 1) msf_rnd = 1;
  const bool use_table = (msf == floorf(msf));
  p.scale = use_table ? (uint32_t)msf : 0u;
  p.scale_magic = p.scale > 1 ? (uint32_t)((0x100000000ull + p.scale - 1) / p.scale) : 0u;
  p.scale_f = msf;
  if (use_table && (sdr->w >= 65536 || (uint64_t)sdr->h + y0 >= 65536))
    return err_status(UHDR_CODEC_INVALID_PARAM, "image dimensions beyond 65535 are not supported");

  const float weight = host::gainmap_weight(*md, max_display_boost);
  UHDR_TRY(get_apply_tables(c, *md, weight, use_table ? (int)p.scale : msf_rnd, &p.tables));
  if (out_ct == UHDR_CT_HLG) {
    UHDR_TRY(upload_lut(&c->d_hlg_oetf, host::oetf_code_thresholds(UHDR_CT_HLG), c->stream));
    p.oetf_thr = c->d_hlg_oetf;
  } else if (out_ct == UHDR_CT_PQ) {
    UHDR_TRY(upload_lut(&c->d_pq_oetf, host::pq_oetf_code_lut(), c->stream));
    p.oetf_thr = c->d_pq_oetf;
  }
  if (out_ct == UHDR_CT_HLG || out_ct == UHDR_CT_PQ) {
    // no HDR-side gamut conversion between the nit scaling and the OETF: the table absorbs (x * 203) / peak as well
    const bool pre = !p.hdr_gamut_on && host::oetf_code_buckets(out_ct, true).exact;
    const host::OetfBuckets& b = host::oetf_code_buckets(out_ct, pre);
    if (b.exact) {  // otherwise the quad kernel is not offered this transfer (apply_quad_mode) and the generic kernel runs
      float** slot = out_ct == UHDR_CT_HLG ? (pre ? &c->d_hlg_buckets_pre : &c->d_hlg_buckets) : (pre ? &c->d_pq_buckets_pre : &c->d_pq_buckets);
      if (!*slot) {
        std::vector<float> raw(b.entries.size());
        memcpy(raw.data(), b.entries.data(), raw.size() * sizeof(float));
        UHDR_TRY(upload_lut(slot, raw, c->stream));
      }
      p.oetf_buckets = (const uint2*)*slot;
      p.oetf_n = b.n;
      p.oetf_base8 = b.base * 8;
      p.oetf_lo_bits = b.clamp_lo_bits;
      p.oetf_hi_bits = b.hi_bits;
      p.oetf_prescaled = pre ? 1 : 0;
    }
  }
  p.sdr = view_of(sdr);
  p.gm = view_of(gm);
  p.dst = view_mut_of(dest);
  p.y0 = y0;
  p.map_ch = gm->fmt == UHDR_IMG_FMT_8bppYCbCr400 ? 1 : 3;
  p.map_bpp = gm->fmt == UHDR_IMG_FMT_8bppYCbCr400 ? 1 : (gm->fmt == UHDR_IMG_FMT_32bppRGBA8888 ? 4 : 3);
  p.out_ct = out_ct;
  p.sdr_is_rgb = is_rgb_fmt_host(sdr->fmt) ? 1 : 0;  // RGB888 is NOT in isPixelFormatRgb (gainmapmath.cpp:1274)
  const bool single = host::metadata_channels_identical(*md);
  for (int i = 0; i < 3; i++) {
    const int k = single ? 0 : i;
    p.gamma_inv[i] = 1.0f / md->gamma[k];
    p.gamma_is_one[i] = p.gamma_inv[i] == 1.0f ? 1 : 0;
    p.offset_sdr[i] = md->offset_sdr[i];
    p.offset_hdr[i] = md->offset_hdr[i];
  }
  p.yuv = host::yuv2rgb_coeffs(UHDR_CG_DISPLAY_P3);
  return ok_status();
}

namespace {
constexpr uint64_t kMallHotBytes = 160ull << 20;
size_t input_bytes(const uhdr_raw_image_t* im) {
  size_t n = 0;
  const ImageView v = view_of(im);
  if (im->fmt == UHDR_IMG_FMT_12bppYCbCr420) n = (size_t)v.stride[0] * v.h + (size_t)v.stride[1] * (v.h / 2) + (size_t)v.stride[2] * (v.h / 2);
  else if (im->fmt == UHDR_IMG_FMT_8bppYCbCr400) n = (size_t)v.stride[0] * v.h;
  else if (im->fmt == UHDR_IMG_FMT_24bppRGB888) n = (size_t)v.stride[0] * v.h * 3;
  else n = (size_t)v.stride[0] * v.h * 4;
  return n;
}
// true: `key` (a frame's luma plane stands for all its planes) was read so recently that it should still be cached; records the read
bool mall_touch(uhdr_hip_ctx* c, const void* key, size_t bytes) {
  bool hot = false;
  for (auto& e : c->mall)
    if (e.p == key) {
      hot = c->mall_clock - e.stamp < kMallHotBytes;
      e.stamp = c->mall_clock + bytes;
      c->mall_clock += bytes;
      return hot;
    }
  if (c->mall.size() >= 64) c->mall.erase(c->mall.begin());
  c->mall_clock += bytes;
  c->mall.push_back({key, c->mall_clock});
  return false;
}
}  // namespace

uhdr_error_info_t uhdr_hip_apply_gainmap_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr,
                                             const uhdr_raw_image_t* gm, const uhdr_gainmap_metadata_t* md,
                                             uhdr_color_transfer_t out_ct, uhdr_img_fmt_t out_fmt,
                                             float max_display_boost, uhdr_raw_image_t* dest, unsigned int y0,
                                             unsigned int full_height) {
  (void)out_fmt;
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipSetDevice(c->device));
  ApplyParams p;
  UHDR_TRY(build_apply_params(c, sdr, gm, md, out_ct, max_display_boost, dest, y0, full_height, &p));
  p.inputs_hot = mall_touch(c, sdr->planes[0], input_bytes(sdr) + input_bytes(gm)) ? 1u : 0u;
  {
    ProfScope ps(c, "apply_gainmap");
    HIP_TRY(launch_apply_gainmap(p, c->stream));
  }
  return ok_status();
}

// Batch of n frames with identical geometry, formats, colour aspects and metadata (burst / video
// style decode, BASELINE config 5): ONE launch walks all frames, so table staging, launch latency
// and the pipeline ramp are paid once.
uhdr_error_info_t uhdr_hip_apply_gainmap_batch_dev(uhdr_hip_ctx_t* c, unsigned int n, const uhdr_raw_image_t* sdr,
                                                   const uhdr_raw_image_t* gm, const uhdr_gainmap_metadata_t* md,
                                                   uhdr_color_transfer_t out_ct, uhdr_img_fmt_t out_fmt,
                                                   float max_display_boost, uhdr_raw_image_t* dest) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (n == 0 || !sdr || !gm || !dest) return err_status(UHDR_CODEC_INVALID_PARAM, "received empty batch or nullptr array");
  HIP_TRY(hipSetDevice(c->device));
  ApplyParams p;
  UHDR_TRY(build_apply_params(c, &sdr[0], &gm[0], md, out_ct, max_display_boost, &dest[0], 0, 0, &p));
  bool uniform = true;
  for (unsigned int i = 1; i < n && uniform; i++) {
    uniform = sdr[i].fmt == sdr[0].fmt && sdr[i].w == sdr[0].w && sdr[i].h == sdr[0].h && sdr[i].cg == sdr[0].cg &&
              gm[i].fmt == gm[0].fmt && gm[i].w == gm[0].w && gm[i].h == gm[0].h && gm[i].cg == gm[0].cg &&
              dest[i].fmt == dest[0].fmt && dest[i].w == dest[0].w && dest[i].h == dest[0].h &&
              !memcmp(sdr[i].stride, sdr[0].stride, sizeof sdr[0].stride) && gm[i].stride[0] == gm[0].stride[0] &&
              dest[i].stride[0] == dest[0].stride[0] && sdr[i].planes[0] && sdr[i].planes[1] && sdr[i].planes[2] &&
              gm[i].planes[0] && dest[i].planes[0] && ((uintptr_t)sdr[i].planes[0] % 2 == 0) &&
              ((uintptr_t)dest[i].planes[0] % 16 == 0) && ((uintptr_t)gm[i].planes[0] % 8 == 0);
  }
  if (n == 1 || !uniform || apply_quad_mode(p) < 0) {  // no batch kernel for this combination: frame by frame
    for (unsigned int i = 0; i < n; i++)
      UHDR_TRY(uhdr_hip_apply_gainmap_dev(c, &sdr[i], &gm[i], md, out_ct, out_fmt, max_display_boost, &dest[i], 0, 0));
    return ok_status();
  }
  // The frame pointers travel in the kernel arguments (ApplyParams::frame_tab, <= kMaxBatchFrames per launch): no
  // table upload, nothing for an in-flight launch to lose, and the call records into a HIP graph as kernel nodes only.
  std::vector<FramePtrs> tab(n);
  for (unsigned int i = 0; i < n; i++) {
    tab[i].y = (const uint8_t*)sdr[i].planes[0];
    tab[i].u = (const uint8_t*)sdr[i].planes[1];
    tab[i].v = (const uint8_t*)sdr[i].planes[2];
    tab[i].map = (const uint8_t*)gm[i].planes[0];
    tab[i].dst = (uint8_t*)dest[i].planes[0];
    dest[i].cg = dest[0].cg;
  }
  // Launch in chunks of at most 16 frames: the waves of one launch are spread over all of its frames, and beyond
  // ~16 separate frame allocations the concurrent access streams lose DRAM locality (measured: 16 frames 5.7 TB/s,
  // 32 frames 5.3 TB/s in one launch); back-to-back launches cost ~3 us each.
  constexpr unsigned int kBatchChunk = kMaxBatchFrames;
  for (unsigned int f0 = 0; f0 < n; f0 += kBatchChunk) {
    const unsigned int nf = (n - f0 < kBatchChunk) ? (n - f0) : kBatchChunk;
    ApplyParams q = p;
    q.n_frames = nf;
    for (unsigned int i = 0; i < nf; i++) q.frame_tab[i] = tab[f0 + i];
    if (nf == 1) {  // a single frame goes through the kernel's direct-pointer path
      q.sdr.p[0] = tab[f0].y; q.sdr.p[1] = tab[f0].u; q.sdr.p[2] = tab[f0].v;
      q.gm.p[0] = tab[f0].map;
      q.dst.p[0] = tab[f0].dst;
    }
    ProfScope ps(c, "apply_gainmap");
    HIP_TRY(launch_apply_gainmap(q, c->stream));
  }
  return ok_status();
}

// applyGainMap with the base image still in coefficient form: JpegDecoderHelper's dequantize + IDCT stage
// (jpegdecoderhelper.cpp:468-535) runs inside the applyGainMap kernel, the 8-bit planes never exist in memory.
uhdr_error_info_t uhdr_hip_apply_gainmap_coef_dev(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_coefficients_t* base, unsigned int w,
                                                  unsigned int h, uhdr_color_gamut_t base_cg, const uhdr_raw_image_t* gm,
                                                  const uhdr_gainmap_metadata_t* md, uhdr_color_transfer_t out_ct,
                                                  uhdr_img_fmt_t out_fmt, float max_display_boost, uhdr_raw_image_t* dest) {
  (void)out_fmt;
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!base) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for the base image coefficients");
  if (w == 0 || h == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "image dimensions cannot be zero, received %ux%u", w, h);
  // the block grid of a 4:2:0 frame (jpeg_component_info::width_in_blocks / height_in_blocks)
  const unsigned int cw = (w + 1) / 2, ch = (h + 1) / 2;
  const unsigned int want_w[3] = {(w + 7) / 8, (cw + 7) / 8, (cw + 7) / 8}, want_h[3] = {(h + 7) / 8, (ch + 7) / 8, (ch + 7) / 8};
  CoefSrc cs;
  for (int i = 0; i < 3; i++) {
    if (!base->coef[i] || ((uintptr_t)base->coef[i] & 15))
      return err_status(UHDR_CODEC_INVALID_PARAM, "coefficient buffer %d is null or not 16-byte aligned", i);
    if (base->blocks_w[i] != (int)want_w[i] || base->blocks_h[i] != (int)want_h[i])
      return err_status(UHDR_CODEC_INVALID_PARAM, "component %d: a %dx%d block grid does not match a 4:2:0 image of %ux%u (expected %ux%u)", i,
                        base->blocks_w[i], base->blocks_h[i], w, h, want_w[i], want_h[i]);
    cs.coef[i] = base->coef[i];
    cs.bw[i] = base->blocks_w[i];
    cs.bh[i] = base->blocks_h[i];
    for (int k = 0; k < 64; k++) {
      if (base->qtable[i][k] == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "component %d: quantization table entry %d is zero", i, k);
      cs.q[i][k] = base->qtable[i][k];
    }
  }
  HIP_TRY(hipSetDevice(c->device));
  // geometry-only view of the image the coefficients decode to (the kernel never dereferences these planes)
  uhdr_raw_image_t sdr;
  memset(&sdr, 0, sizeof sdr);
  sdr.fmt = UHDR_IMG_FMT_12bppYCbCr420;
  sdr.cg = base_cg; sdr.ct = UHDR_CT_SRGB; sdr.range = UHDR_CR_FULL_RANGE;
  sdr.w = w; sdr.h = h;
  for (int i = 0; i < 3; i++) { sdr.planes[i] = (void*)base->coef[i]; sdr.stride[i] = (unsigned int)base->blocks_w[i] * 8; }
  ApplyParams p;
  UHDR_TRY(build_apply_params(c, &sdr, gm, md, out_ct, max_display_boost, dest, 0, 0, &p));
  if (apply_quad_mode(p) < 0)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "apply_gainmap_coef covers the 2x2-quad kernel's cases (even dimensions, width >= 128, 16-byte "
                      "aligned destination rows, gain map at scale 1 or an even scale <= 8 with gamma 1); decode with idct_dequant and call apply_gainmap");
  constexpr unsigned int kSlots = 8;
  if (!c->d_coef_src) HIP_TRY(hipMalloc((void**)&c->d_coef_src, sizeof(CoefSrc) * kSlots));
  CoefSrc* slot = c->d_coef_src + (c->coef_src_next++ % kSlots);
  HIP_TRY(hipMemcpyAsync(slot, &cs, sizeof cs, hipMemcpyHostToDevice, c->stream));  // pageable source: staged before return
  p.coef_src = slot;
  {
    ProfScope ps(c, "apply_gainmap");
    HIP_TRY(launch_apply_gainmap_coef(p, c->stream));
  }
  return ok_status();
}

uhdr_error_info_t uhdr_hip_apply_gainmap(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr,
                                         const uhdr_raw_image_t* gm, const uhdr_gainmap_metadata_t* md,
                                         uhdr_color_transfer_t out_ct, uhdr_img_fmt_t out_fmt,
                                         float max_display_boost, uhdr_raw_image_t* dest) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  UHDR_TRY(validate_apply(sdr, gm, md, out_ct, dest));
  HIP_TRY(hipSetDevice(c->device));
  uhdr_raw_image_t dsdr, dgm, ddst;
  UHDR_TRY(stage_in(c, 0, sdr, &dsdr, true));
  UHDR_TRY(stage_in(c, 1, gm, &dgm, true));
  UHDR_TRY(stage_in(c, 2, dest, &ddst, false));
  uhdr_error_info_t st = uhdr_hip_apply_gainmap_dev(c, &dsdr, &dgm, md, out_ct, out_fmt, max_display_boost, &ddst, 0, 0);
  if (st.error_code != UHDR_CODEC_OK) return st;
  dest->cg = ddst.cg;
  return stage_out(c, &ddst, dest);
}

// -------------------------------------------------------------------------------------------------
// generateGainMap
// -------------------------------------------------------------------------------------------------
static uhdr_error_info_t fill_gen_params(uhdr_hip_ctx* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                         const uhdr_hip_encode_cfg_t* cfg, GenParams* p, int* use_base_cg,
                                         float* hdr_white_nits_out, bool sdr_in_registers = false) {
  // sdr_in_registers: the fused API-0 front end renders the SDR pixel itself and never reads SDR planes
  if (!sdr || !hdr || !cfg) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  // format checks: jpegr.cpp:537-562
  if (sdr->fmt != UHDR_IMG_FMT_24bppYCbCr444 && sdr->fmt != UHDR_IMG_FMT_16bppYCbCr422 &&
      sdr->fmt != UHDR_IMG_FMT_12bppYCbCr420 && sdr->fmt != UHDR_IMG_FMT_32bppRGBA8888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "generate gainmap method expects sdr intent color format to be one of "
                      "{UHDR_IMG_FMT_24bppYCbCr444, UHDR_IMG_FMT_16bppYCbCr422, UHDR_IMG_FMT_12bppYCbCr420, "
                      "UHDR_IMG_FMT_32bppRGBA8888}. Received %d", sdr->fmt);
  if (hdr->fmt != UHDR_IMG_FMT_24bppYCbCrP010 && hdr->fmt != UHDR_IMG_FMT_30bppYCbCr444 &&
      hdr->fmt != UHDR_IMG_FMT_32bppRGBA1010102 && hdr->fmt != UHDR_IMG_FMT_64bppRGBAHalfFloat)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "generate gainmap method expects hdr intent color format to be one of "
                      "{UHDR_IMG_FMT_24bppYCbCrP010, UHDR_IMG_FMT_30bppYCbCr444, UHDR_IMG_FMT_32bppRGBA1010102, "
                      "UHDR_IMG_FMT_64bppRGBAHalfFloat}. Received %d", hdr->fmt);
  if (hdr->ct < UHDR_CT_LINEAR || hdr->ct > UHDR_CT_SRGB)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for converting transfer characteristics %d to linear", hdr->ct);
  if (hdr->cg < UHDR_CG_BT_709 || hdr->cg > UHDR_CG_BT_2100)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for calculating luminance for color gamut %d", hdr->cg);
  if (sdr->cg < UHDR_CG_BT_709 || sdr->cg > UHDR_CG_BT_2100)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for converting yuv to rgb for color gamut %d", sdr->cg);
  if (sdr->w != hdr->w || sdr->h != hdr->h)
    return err_status(UHDR_CODEC_INVALID_PARAM, "sdr intent resolution %ux%u and hdr intent resolution %ux%u do not match",
                      sdr->w, sdr->h, hdr->w, hdr->h);
  if (cfg->map_dimension_scale_factor < 1)
    return err_status(UHDR_CODEC_INVALID_PARAM, "gainmap scale factor %d is not positive", cfg->map_dimension_scale_factor);
  if (!sdr_in_registers) UHDR_TRY(validate_image(sdr, "sdr intent"));
  UHDR_TRY(validate_image(hdr, "hdr intent"));
  memset(p, 0, sizeof *p);
  const float hdr_white_nits = host::reference_peak_nits(hdr->ct);
  *hdr_white_nits_out = hdr_white_nits;
  // gamut handling: jpegr.cpp:605-638 with kWriteXmpMetadata == false (ISO-only build, the default)
  int use_sdr_cg = 1;
  bool identity;
  if (sdr->cg != hdr->cg) {
    use_sdr_cg = !(hdr->cg == UHDR_CG_BT_2100 || (hdr->cg == UHDR_CG_DISPLAY_P3 && sdr->cg != UHDR_CG_BT_2100));
    if (use_sdr_cg) {
      host::gamut_matrix(sdr->cg, hdr->cg, &p->hdr_gamut, &identity);
      p->hdr_gamut_on = 1;
    } else {
      host::gamut_matrix(hdr->cg, sdr->cg, &p->sdr_gamut, &identity);
      p->sdr_gamut_on = 1;
    }
  }
  *use_base_cg = use_sdr_cg;
  p->sdr_yuv = host::yuv2rgb_coeffs(cfg->sdr_is_601 ? UHDR_CG_DISPLAY_P3 : sdr->cg);
  p->hdr_yuv = host::yuv2rgb_coeffs(hdr->cg);
  host::luminance_coeffs(sdr->cg, p->lum);
  p->sdr = view_of(sdr);
  p->hdr = view_of(hdr);
  uint32_t scale = (uint32_t)cfg->map_dimension_scale_factor;
  uint32_t mw = sdr->w / scale, mh = sdr->h / scale;
  if (mw == 0 || mh == 0) {  // jpegr.cpp:696-706
    uint32_t s = sdr->w < sdr->h ? sdr->w : sdr->h;
    s = (s >= 8) ? (s / 8) : 1;
    scale = s;
    mw = sdr->w / scale;
    mh = sdr->h / scale;
  }
  p->scale = scale; p->map_w = mw; p->map_h = mh;
  p->srgb_lut = c->d_srgb;
  UHDR_TRY(select_hdr_lut(c, hdr->ct, &p->hdr_inv_lut, &p->hdr_inv_n));
  UHDR_TRY(upload_math(c));
  p->math_tab = c->d_math;
  p->sdr_is_rgb = is_rgb_fmt_host(sdr->fmt);
  p->hdr_is_rgb = is_rgb_fmt_host(hdr->fmt);
  p->multichannel = cfg->use_multi_channel_gainmap != 0;
  p->use_luminance = cfg->use_luminance != 0;
  p->hdr_nits = hdr->ct == UHDR_CT_LINEAR ? 203.0f : hdr_white_nits;
  p->gamma = cfg->gamma;
  p->gain_cap = host::gain_cap_ratio();
  if (!(p->gain_cap > 0.0f)) return err_status(UHDR_CODEC_ERROR, "internal: the dark-pixel gain cap has no exact form in the ratio domain");
  return ok_status();
}

// Everything between the two passes in one launch (generate_gainmap.hip: minmax_table_kernel).  `merged_in` non-null:
// the striped path's second half (finalize from the all-reduced extrema + table).
static void fill_finalize(MinmaxTableParams* t, const uhdr_hip_encode_cfg_t* cfg) {
  t->nch = (cfg && cfg->use_multi_channel_gainmap) ? 3 : 1;
  t->has_max_hint = cfg && cfg->max_content_boost != FLT_MAX;
  t->has_min_hint = cfg && cfg->min_content_boost != FLT_MIN;
  t->log2_max_hint = t->has_max_hint ? log2f(cfg->max_content_boost) : 0.0f;
  t->log2_min_hint = t->has_min_hint ? log2f(cfg->min_content_boost) : 0.0f;
  t->gamma = cfg ? cfg->gamma : 1.0f;
}

static void fill_gainmap_desc(const uhdr_raw_image_t* hdr, const GenParams& p, uhdr_raw_image_t* gm) {
  gm->fmt = p.multichannel ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_8bppYCbCr400;
  gm->cg = hdr->cg; gm->ct = hdr->ct; gm->range = hdr->range;
  gm->w = p.map_w; gm->h = p.map_h;
}

static uhdr_error_info_t uhdr_hip_generate_gainmap_finalize_md(const uhdr_hip_encode_cfg_t* cfg, uhdr_color_transfer_t hdr_ct,
                                                               int use_base_cg, const float mm[6], uhdr_gainmap_metadata_t* md);
uhdr_error_info_t uhdr_hip_generate_gainmap_finalize(const uhdr_hip_encode_cfg_t* cfg, uhdr_color_transfer_t hdr_ct,
                                                     int use_base_cg, float mm[6], uhdr_gainmap_metadata_t* md) {
  if (!cfg || !mm || !md) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  const int nch = cfg->use_multi_channel_gainmap ? 3 : 1;
  float* gmin = mm;
  float* gmax = mm + 3;
  for (int i = 0; i < nch; i++) {  // jpegr.cpp:969-986
    gmin[i] = gmin[i] < -14.3f ? -14.3f : (gmin[i] > 15.6f ? 15.6f : gmin[i]);
    gmax[i] = gmax[i] < -14.3f ? -14.3f : (gmax[i] > 15.6f ? 15.6f : gmax[i]);
    if (cfg->max_content_boost != FLT_MAX) {
      const float s = log2f(cfg->max_content_boost);
      gmax[i] = gmax[i] < s ? gmax[i] : s;
    }
    if (cfg->min_content_boost != FLT_MIN) {
      const float s = log2f(cfg->min_content_boost);
      gmin[i] = gmin[i] < s ? s : gmin[i];
    }
    if (fabsf(gmax[i] - gmin[i]) < FLT_EPSILON) gmax[i] += 0.1f;
  }
  return uhdr_hip_generate_gainmap_finalize_md(cfg, hdr_ct, use_base_cg, mm, md);
}

// the metadata fill of jpegr.cpp:1031-1048 from a FINAL per-channel range
static uhdr_error_info_t uhdr_hip_generate_gainmap_finalize_md(const uhdr_hip_encode_cfg_t* cfg, uhdr_color_transfer_t hdr_ct,
                                                               int use_base_cg, const float mm[6], uhdr_gainmap_metadata_t* md) {
  const int nch = cfg->use_multi_channel_gainmap ? 3 : 1;
  const float* gmin = mm;
  const float* gmax = mm + 3;
  for (int i = 0; i < 3; i++) {  // jpegr.cpp:1031-1048
    const int k = nch == 3 ? i : 0;
    md->max_content_boost[i] = exp2f(gmax[k]);
    md->min_content_boost[i] = exp2f(gmin[k]);
    md->gamma[i] = cfg->gamma;
    md->offset_sdr[i] = 1e-7f;
    md->offset_hdr[i] = 1e-7f;
  }
  const float hdr_white_nits = host::reference_peak_nits(hdr_ct);
  md->hdr_capacity_min = 1.0f;
  md->hdr_capacity_max = cfg->target_disp_peak_nits != -1.0f ? cfg->target_disp_peak_nits / 203.0f : hdr_white_nits / 203.0f;
  md->use_base_cg = use_base_cg;
  return ok_status();
}

uhdr_error_info_t uhdr_hip_generate_gainmap_pass1_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr,
                                                      const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                                      float* gain_log2_dev, float* minmax_dev, int* use_base_cg) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!gain_log2_dev || !minmax_dev || !use_base_cg) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  HIP_TRY(hipSetDevice(c->device));
  GenParams p;
  float white;
  UHDR_TRY(fill_gen_params(c, sdr, hdr, cfg, &p, use_base_cg, &white));
  // The small-image fallback of jpegr.cpp:696-706 belongs to the WHOLE image; a stripe must keep the configured scale
  // factor (its caller sizes gain_log2_dev as (w / scale) * (h / scale) samples), so a stripe too short for one map
  // row is an error here -- such a rank launches nothing and contributes the identity {127, -128} to the merge.
  if (p.scale != (uint32_t)cfg->map_dimension_scale_factor)
    return err_status(UHDR_CODEC_INVALID_PARAM, "stripe %ux%u holds no map sample at scale factor %d (pass1 takes stripes of at least "
                      "scale rows / columns; the reference's small-image fallback applies to whole images only)", sdr->w, sdr->h,
                      cfg->map_dimension_scale_factor);
  UHDR_TRY(ensure(c->minmax, (6 + 2048 * 6) * sizeof(float)));
  p.gain_log2 = gain_log2_dev;
  p.minmax = (float*)c->minmax.p;
  {
    ProfScope ps(c, "generate_gainmap");
    HIP_TRY(launch_generate_gainmap(p, true, c->stream));
    MinmaxTableParams t;  // ratio extrema of the stripe -> the reference's six log2 extrema
    memset(&t, 0, sizeof t);
    t.do_reduce = 1;
    t.partials = p.minmax + 6;
    t.n_partials = gen_partials_count(p);
    t.mm6 = minmax_dev;
    t.math_tab = c->d_math;
    HIP_TRY(launch_minmax_table(t, c->stream));
  }
  return ok_status();
}

uhdr_error_info_t uhdr_hip_generate_gainmap_pass2_dev(uhdr_hip_ctx_t* c, const float* gain_log2_dev, const float mm[6],
                                                      const uhdr_hip_encode_cfg_t* cfg, uhdr_raw_image_t* gm) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!gain_log2_dev || !mm || !cfg || !gm || !gm->planes[0]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  if (gm->stride[0] < gm->w) return err_status(UHDR_CODEC_INVALID_PARAM, "gainmap stride (%u) cannot be less than its width (%u)", gm->stride[0], gm->w);
  HIP_TRY(hipSetDevice(c->device));
  UHDR_TRY(upload_math(c));
  UHDR_TRY(ensure(c->affine, kAffineDevBytes));
  MinmaxTableParams t;  // the final range arrives from the host: only the step tables are left to build
  memset(&t, 0, sizeof t);
  t.do_table = 1;
  fill_finalize(&t, cfg);
  for (int i = 0; i < 6; i++) t.final_mm[i] = mm[i];
  t.dev = (AffineDev*)c->affine.p;
  t.math_tab = c->d_math;
  AffineParams a;
  memset(&a, 0, sizeof a);
  a.dev = (const AffineDev*)c->affine.p;
  a.math_tab = c->d_math;
  a.gain_log2 = gain_log2_dev;
  a.out = (uint8_t*)gm->planes[0];
  a.map_w = gm->w; a.map_h = gm->h; a.out_stride = gm->stride[0];
  a.nch = cfg->use_multi_channel_gainmap ? 3 : 1;
  a.gamma = cfg->gamma;
  ProfScope ps(c, "generate_gainmap");
  HIP_TRY(launch_minmax_table(t, c->stream));
  HIP_TRY(launch_affine_map(a, c->stream));
  return ok_status();
}

static void note_table_stats(uhdr_hip_ctx* c, const uhdr_hip_encode_cfg_t* cfg) {  // after the synchronisation that landed c->h_mm
  const int nch = cfg->use_multi_channel_gainmap ? 3 : 1;
  for (int i = 0; i < nch; i++) {
    if (c->h_mm[6 + i] != 0.0f) c->stats.generate_channels_tabled++;
    else c->stats.generate_channels_per_sample++;
  }
}

// pass 1's partials -> extrema -> final range -> step tables -> pass 2, all stream ordered; the final range is copied to
// the pinned c->h_mm for the caller's metadata fill (after ITS synchronisation)
static uhdr_error_info_t two_pass_tail(uhdr_hip_ctx* c, const GenParams& p, int n_partials, const uhdr_hip_encode_cfg_t* cfg, uhdr_raw_image_t* gm) {
  UHDR_TRY(ensure(c->affine, kAffineDevBytes));
  UHDR_TRY(ensure(c->exchange, 256));
  if (!c->h_mm) HIP_TRY(hipHostMalloc((void**)&c->h_mm, 9 * sizeof(float), hipHostMallocDefault));
  float* final_mm = (float*)((char*)c->exchange.p + 192);
  MinmaxTableParams t;
  memset(&t, 0, sizeof t);
  t.do_reduce = t.do_finalize = t.do_table = 1;
  t.partials = p.minmax + 6;
  t.n_partials = n_partials;
  t.mm6 = p.minmax;
  fill_finalize(&t, cfg);
  t.out_mm = final_mm;
  t.dev = (AffineDev*)c->affine.p;
  t.math_tab = c->d_math;
  AffineParams a;
  memset(&a, 0, sizeof a);
  a.dev = (const AffineDev*)c->affine.p;
  a.math_tab = c->d_math;
  a.gain_log2 = p.gain_log2;
  a.out = (uint8_t*)gm->planes[0];
  a.map_w = gm->w; a.map_h = gm->h; a.out_stride = gm->stride[0];
  a.nch = cfg->use_multi_channel_gainmap ? 3 : 1;
  a.gamma = cfg->gamma;
  {
    ProfScope ps(c, "generate_gainmap");
    HIP_TRY(launch_minmax_table(t, c->stream));
    HIP_TRY(launch_affine_map(a, c->stream));
  }
  HIP_TRY(hipMemcpyAsync(c->h_mm, final_mm, 9 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_generate_gainmap_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                                const uhdr_hip_encode_cfg_t* cfg, uhdr_gainmap_metadata_t* md,
                                                uhdr_raw_image_t* gm) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!md || !gm || !gm->planes[0]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for gainmap metadata or image");
  HIP_TRY(hipSetDevice(c->device));
  GenParams p;
  int use_base_cg = 1;
  float hdr_white_nits;
  UHDR_TRY(fill_gen_params(c, sdr, hdr, cfg, &p, &use_base_cg, &hdr_white_nits));
  fill_gainmap_desc(hdr, p, gm);
  if (gm->stride[0] < gm->w) return err_status(UHDR_CODEC_INVALID_PARAM, "gainmap stride (%u) cannot be less than its width (%u)", gm->stride[0], gm->w);
  if (cfg->preset == UHDR_USAGE_REALTIME) {  // one pass: jpegr.cpp:724-737
    for (int i = 0; i < 3; i++) {
      md->max_content_boost[i] = hdr_white_nits / 203.0f;
      md->min_content_boost[i] = 1.0f;
      md->gamma[i] = cfg->gamma;
      md->offset_sdr[i] = 0.0f;
      md->offset_hdr[i] = 0.0f;
    }
    md->hdr_capacity_min = 1.0f;
    md->hdr_capacity_max = cfg->target_disp_peak_nits != -1.0f ? cfg->target_disp_peak_nits / 203.0f : md->max_content_boost[0];
    md->use_base_cg = use_base_cg;
    p.min_boost = md->min_content_boost[0];
    p.max_boost = md->max_content_boost[0];
    p.log2min = log2f(md->min_content_boost[0]);
    p.log2max = log2f(md->max_content_boost[0]);
    p.log2_range = (double)(p.log2max - p.log2min);
    p.log2_range_rcp = 1.0 / p.log2_range;
    UHDR_TRY(gain_step_table(c, p, &p.gain8));
    p.out = (uint8_t*)gm->planes[0];
    p.out_stride = gm->stride[0];
    ProfScope ps(c, "generate_gainmap");
    HIP_TRY(launch_generate_gainmap(p, false, c->stream));
    return ok_status();
  }
  // two pass on one device
  const size_t nfl = (size_t)p.map_w * p.map_h * (p.multichannel ? 3 : 1);
  UHDR_TRY(ensure(c->scratch[7], nfl * sizeof(float)));
  UHDR_TRY(ensure(c->minmax, (6 + 2048 * 6) * sizeof(float)));
  p.gain_log2 = (float*)c->scratch[7].p;
  p.minmax = (float*)c->minmax.p;
  {
    ProfScope ps(c, "generate_gainmap");
    HIP_TRY(launch_generate_gainmap(p, true, c->stream));
  }
  UHDR_TRY(two_pass_tail(c, p, gen_partials_count(p), cfg, gm));
  HIP_TRY(hipStreamSynchronize(c->stream));  // the only host synchronisation: the metadata needs the final range
  float mm[6];
  memcpy(mm, c->h_mm, sizeof mm);
  note_table_stats(c, cfg);
  return uhdr_hip_generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
}

// -------------------------------------------------------------------------------------------------
// Row-striped two-pass generation across GPUs (one process per GPU): the one exchange step of the whole hot path
// (jpegr.cpp:932-938, the per-channel min / max merge) as ONE ncclAllReduce(min) over {min0..2, -max0..2}, issued
// from here on the context's stream, with the finalisation (jpegr.cpp:969-986) on the device: pass 1 -> all-reduce ->
// finalize -> pass 2 is one stream-ordered sequence, the host synchronises once at the end for the metadata.
// RCCL is bound at run time (rccl_bind.cpp): no link-time dependency, single-GPU users never load it.
// -------------------------------------------------------------------------------------------------
namespace {
#define RCCL_TRY(expr)                                                                                             \
  do {                                                                                                             \
    ncclResult_t r_ = (expr);                                                                                      \
    if (r_ != ncclSuccess) return err_status(UHDR_CODEC_ERROR, "RCCL: %s failed: %s", #expr, rccl().GetErrorString(r_)); \
  } while (0)
}  // namespace

int uhdr_hip_comm_unique_id(unsigned char id[UHDR_HIP_COMM_ID_BYTES]) {
  static_assert(sizeof(ncclUniqueId) == UHDR_HIP_COMM_ID_BYTES, "ncclUniqueId size");
  if (!id || !rccl().ok) return -1;
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess) return -1;
  memcpy(id, &u, sizeof u);
  return 0;
}

uhdr_error_info_t uhdr_hip_comm_init(uhdr_hip_ctx_t* c, const unsigned char id[UHDR_HIP_COMM_ID_BYTES], int rank, int nranks) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!id || nranks < 1 || rank < 0 || rank >= nranks) return err_status(UHDR_CODEC_INVALID_PARAM, "bad communicator arguments (rank %d of %d)", rank, nranks);
  if (!rccl().ok) return err_status(UHDR_CODEC_ERROR, "RCCL is not available in this process (librccl.so.1 could not be loaded)");
  if (c->comm || c->comm_custom) return err_status(UHDR_CODEC_INVALID_OPERATION, "this context already has a communicator");
  HIP_TRY(hipSetDevice(c->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  ncclComm_t comm = nullptr;
  RCCL_TRY(rccl().CommInitRank(&comm, nranks, u, rank));
  c->comm = comm;
  c->comm_rank = rank;
  RCCL_TRY(rccl().CommCount(comm, &c->comm_size));
  return ok_status();
}

// The same exchange steps over a caller-provided transport (an MPI / gloo / shared-memory relay, or a test double): the
// library calls the functions in stream order with device pointers and its own stream; RCCL stays the default.
uhdr_error_info_t uhdr_hip_comm_init_custom(uhdr_hip_ctx_t* c, const uhdr_hip_comm_ops_t* ops, int rank, int nranks) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!ops || !ops->all_reduce_min_f32 || nranks < 1 || rank < 0 || rank >= nranks)
    return err_status(UHDR_CODEC_INVALID_PARAM, "bad communicator arguments (rank %d of %d, all_reduce_min_f32 is required)", rank, nranks);
  if (c->comm || c->comm_custom) return err_status(UHDR_CODEC_INVALID_OPERATION, "this context already has a communicator");
  c->comm_ops = *ops;
  c->comm_custom = true;
  c->comm_rank = rank;
  c->comm_size = nranks;
  return ok_status();
}

void uhdr_hip_comm_destroy(uhdr_hip_ctx_t* c) {
  if (!c || (!c->comm && !c->comm_custom)) return;
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy((ncclComm_t)c->comm);
  c->comm = nullptr;
  c->comm_custom = false;
  memset(&c->comm_ops, 0, sizeof c->comm_ops);
  c->comm_size = 0;
}

int uhdr_hip_comm_size(uhdr_hip_ctx_t* c) { return c ? c->comm_size : 0; }
int uhdr_hip_comm_rank(uhdr_hip_ctx_t* c) { return c ? c->comm_rank : 0; }

namespace {
// THE collective of the hot path: MIN over a handful of floats, in place, on the library's own stream
uhdr_error_info_t comm_all_reduce_min(uhdr_hip_ctx* c, float* buf, size_t n) {
  if (c->comm_custom) {
    const int rc = c->comm_ops.all_reduce_min_f32(c->comm_ops.user, buf, n, (void*)c->stream);
    if (rc != 0) return err_status(UHDR_CODEC_ERROR, "custom transport: all_reduce_min_f32 failed (%d)", rc);
  } else if (c->comm) {
    RCCL_TRY(rccl().AllReduce(buf, buf, n, ncclFloat, ncclMin, (ncclComm_t)c->comm, c->stream));
  }
  return ok_status();
}
}  // namespace

// The hot path's one collective on its own (device pointer, in place, enqueued on the context's stream): what
// uhdr_hip_generate_gainmap_striped_dev / uhdr_hip_encode_api1_fused_dev issue between their passes.  Without a communicator: nothing.
uhdr_error_info_t uhdr_hip_comm_all_reduce_min_dev(uhdr_hip_ctx_t* c, float* buf, size_t n) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!buf || n == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "all_reduce: nullptr buffer or zero count");
  HIP_TRY(hipSetDevice(c->device));
  ProfScope ps(c, "stripe_exchange");
  return comm_all_reduce_min(c, buf, n);
}

// Every rank contributes `bytes` bytes; recv (nranks * bytes) holds them in rank order on every rank.  Device pointers,
// enqueued on the context's stream.  Without a communicator: a copy.
uhdr_error_info_t uhdr_hip_comm_all_gather_dev(uhdr_hip_ctx_t* c, const void* send, void* recv, size_t bytes) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (bytes == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "all_gather: zero size");  // (the same on every rank: nobody enters the collective)
  HIP_TRY(hipSetDevice(c->device));
  // a rank that was handed a nullptr still joins the collective (zeros from / into scratch) and reports afterwards: its peers
  // must not wait for it forever
  uhdr_error_info_t local = ok_status();
  if (!send || !recv) {
    local = err_status(UHDR_CODEC_INVALID_PARAM, "all_gather: nullptr buffer");
    const size_t nr = (size_t)(c->comm_size > 0 ? c->comm_size : 1);
    UHDR_TRY(ensure(c->scratch[6], bytes * (nr + 1)));
    HIP_TRY(hipMemsetAsync(c->scratch[6].p, 0, bytes * (nr + 1), c->stream));
    recv = c->scratch[6].p;
    send = (const char*)c->scratch[6].p + bytes * nr;
  }
  if (c->comm_custom) {
    if (!c->comm_ops.all_gather) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "custom transport without all_gather");
    const int rc = c->comm_ops.all_gather(c->comm_ops.user, send, recv, bytes, (void*)c->stream);
    if (rc != 0) return err_status(UHDR_CODEC_ERROR, "custom transport: all_gather failed (%d)", rc);
  } else if (c->comm) {
    RCCL_TRY(rccl().AllGather(send, recv, bytes, ncclUint8, (ncclComm_t)c->comm, c->stream));
  } else {
    if (send != recv) HIP_TRY(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  return local;
}

// Stripes of unequal size to one rank -- the merge of the stripes' outputs into one image (the reference's threads write
// into one buffer, jpegr.cpp:845-864; gain-map stripes and per-stripe entropy-coded streams here): rank r's send_bytes ==
// counts[r] bytes land at recv + sum(counts[0..r)) on `root`; recv is ignored elsewhere.  counts is a host array of nranks
// entries, identical on every rank.  RCCL: one group of ncclSend / ncclRecv over xGMI, no host staging.
uhdr_error_info_t uhdr_hip_comm_gather_dev(uhdr_hip_ctx_t* c, const void* send, size_t send_bytes, void* recv, const size_t* counts, int root) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  const int n = c->comm_size > 0 ? c->comm_size : 1, rank = c->comm_size > 0 ? c->comm_rank : 0;
  // Without `counts` or a valid root nobody knows what to exchange: that is the one failure that needs the communicator aborted.
  if (!counts || root < 0 || root >= n) return err_status(UHDR_CODEC_INVALID_PARAM, "gather: nullptr counts or root %d outside 0..%d", root, n - 1);
  HIP_TRY(hipSetDevice(c->device));
  // Every other local failure is RECORDED and the rank still takes part in the exchange exactly as `counts` says -- with a
  // scratch buffer in place of the one it cannot use -- so that its peers do not wait in ncclRecv / ncclSend forever (the
  // pattern of uhdr_hip_generate_gainmap_striped_dev).  The error is returned afterwards.
  uhdr_error_info_t local = ok_status();
  if (counts[rank] != send_bytes) local = err_status(UHDR_CODEC_INVALID_PARAM, "gather: counts[%d] = %zu but this rank sends %zu bytes", rank, counts[rank], send_bytes);
  else if ((send_bytes && !send) || (rank == root && !recv)) local = err_status(UHDR_CODEC_INVALID_PARAM, "gather: nullptr buffer");
  if (local.error_code != UHDR_CODEC_OK) {
    size_t total = 0;
    for (int r = 0; r < n; r++) total += counts[r];
    UHDR_TRY(ensure(c->scratch[6], (total ? total : 1) + counts[rank]));
    HIP_TRY(hipMemsetAsync(c->scratch[6].p, 0, (total ? total : 1) + counts[rank], c->stream));
    recv = c->scratch[6].p;                          // (root) somewhere to receive
    send = (const char*)c->scratch[6].p + total;      // counts[rank] zero bytes to send
    send_bytes = counts[rank];
  }
  if (c->comm_custom) {
    if (!c->comm_ops.gather_v) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "custom transport without gather_v");  // (the same on every rank)
    const int rc = c->comm_ops.gather_v(c->comm_ops.user, send, send_bytes, recv, counts, root, (void*)c->stream);
    if (rc != 0) return err_status(UHDR_CODEC_ERROR, "custom transport: gather_v failed (%d)", rc);
    return local;
  }
  size_t off = 0;
  if (!c->comm) {
    if (local.error_code == UHDR_CODEC_OK && send_bytes && send != recv) HIP_TRY(hipMemcpyAsync(recv, send, send_bytes, hipMemcpyDeviceToDevice, c->stream));
    return local;
  }
  RCCL_TRY(rccl().GroupStart());
  ncclResult_t r1 = ncclSuccess;
  if (rank == root) {
    for (int r = 0; r < n && r1 == ncclSuccess; r++) {
      if (r != root && counts[r]) r1 = rccl().Recv((char*)recv + off, counts[r], ncclUint8, r, (ncclComm_t)c->comm, c->stream);
      off += counts[r];
    }
  } else if (send_bytes) {
    r1 = rccl().Send(send, send_bytes, ncclUint8, root, (ncclComm_t)c->comm, c->stream);
  }
  const ncclResult_t r2 = rccl().GroupEnd();
  if (r1 != ncclSuccess) return err_status(UHDR_CODEC_ERROR, "RCCL: send / recv failed: %s", rccl().GetErrorString(r1));
  if (r2 != ncclSuccess) return err_status(UHDR_CODEC_ERROR, "RCCL: ncclGroupEnd failed: %s", rccl().GetErrorString(r2));
  if (rank == root && send_bytes) {
    size_t mine = 0;
    for (int r = 0; r < root; r++) mine += counts[r];
    if ((char*)recv + mine != (const char*)send) HIP_TRY(hipMemcpyAsync((char*)recv + mine, send, send_bytes, hipMemcpyDeviceToDevice, c->stream));
  }
  return local;
}

// A rank must never leave this function without having taken part in the collective: the other ranks would wait in it
// forever.  So everything that can fail locally -- argument and geometry checks, allocation, the launch of pass 1 -- is
// recorded in `local`, the rank then contributes the merge's identity {127, -128} exactly like an empty stripe, runs
// the exchange and the finalisation, and only then returns its error.
uhdr_error_info_t uhdr_hip_generate_gainmap_striped_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                                        const uhdr_hip_encode_cfg_t* cfg, uhdr_gainmap_metadata_t* md,
                                                        uhdr_raw_image_t* gm) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  const bool args_ok = sdr && hdr && cfg && md && gm;
  if (args_ok && cfg->preset == UHDR_USAGE_REALTIME) {  // one pass has no exchange step: every stripe is an independent image
    if (cfg->map_dimension_scale_factor < 1) return err_status(UHDR_CODEC_INVALID_PARAM, "gainmap scale factor %d is not positive", cfg->map_dimension_scale_factor);
    const uint32_t s1 = (uint32_t)cfg->map_dimension_scale_factor;
    if (sdr->h < s1 || sdr->w < s1) {
      // a stripe too short for one map row launches nothing (the whole image's map has H / scale rows); its metadata is the
      // one every other stripe computes (jpegr.cpp:724-737 depends on the transfer function and the gamuts only)
      if (sdr->w != hdr->w || sdr->h != hdr->h) return err_status(UHDR_CODEC_INVALID_PARAM, "sdr intent resolution %ux%u and hdr intent resolution %ux%u do not match", sdr->w, sdr->h, hdr->w, hdr->h);
      if (hdr->ct < UHDR_CT_LINEAR || hdr->ct > UHDR_CT_SRGB) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for converting transfer characteristics %d to linear", hdr->ct);
      const float white = host::reference_peak_nits(hdr->ct);
      for (int i = 0; i < 3; i++) {
        md->max_content_boost[i] = white / 203.0f;
        md->min_content_boost[i] = 1.0f;
        md->gamma[i] = cfg->gamma;
        md->offset_sdr[i] = 0.0f;
        md->offset_hdr[i] = 0.0f;
      }
      md->hdr_capacity_min = 1.0f;
      md->hdr_capacity_max = cfg->target_disp_peak_nits != -1.0f ? cfg->target_disp_peak_nits / 203.0f : md->max_content_boost[0];
      md->use_base_cg = sdr->cg == hdr->cg || !(hdr->cg == UHDR_CG_BT_2100 || (hdr->cg == UHDR_CG_DISPLAY_P3 && sdr->cg != UHDR_CG_BT_2100));
      gm->w = sdr->w / s1;
      gm->h = 0;
      return ok_status();
    }
    // (with a row and a column of map samples the whole-image small-image rule of jpegr.cpp:696-706 cannot re-scale the stripe)
    return uhdr_hip_generate_gainmap_dev(c, sdr, hdr, cfg, md, gm);
  }
  HIP_TRY(hipSetDevice(c->device));
  uhdr_error_info_t local = ok_status();
  auto note = [&](const uhdr_error_info_t& e) { if (local.error_code == UHDR_CODEC_OK && e.error_code != UHDR_CODEC_OK) local = e; };
  auto note_hip = [&](hipError_t e, const char* what) {
    if (e != hipSuccess) note(err_status(UHDR_CODEC_ERROR, "%s: %s", what, hipGetErrorString(e)));
  };
  if (!args_ok) note(err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument"));
  if (args_ok && cfg->map_dimension_scale_factor < 1) note(err_status(UHDR_CODEC_INVALID_PARAM, "gainmap scale factor %d is not positive", cfg->map_dimension_scale_factor));
  // ---- phase 0: validation and allocation, no device work yet --------------------------------------------------------
  // the exchange buffers first: without them this rank cannot even contribute the identity (then, and only then, the
  // function returns early -- the caller has to abort the communicator)
  UHDR_TRY(ensure(c->exchange, 256));
  UHDR_TRY(ensure(c->affine, kAffineDevBytes));
  UHDR_TRY(ensure(c->minmax, (6 + 2048 * 6) * sizeof(float)));
  UHDR_TRY(upload_math(c));
  if (!c->h_mm) HIP_TRY(hipHostMalloc((void**)&c->h_mm, 9 * sizeof(float), hipHostMallocDefault));
  float* merged = (float*)c->exchange.p;                       // 6 floats
  AffineDev* adev = (AffineDev*)c->affine.p;
  float* final_mm = (float*)((char*)c->exchange.p + 192);      // 6 floats
  const uint32_t scale = local.error_code == UHDR_CODEC_OK ? (uint32_t)cfg->map_dimension_scale_factor : 1u;
  // a stripe shorter than one map row (the last rank of an uneven split) launches nothing and contributes the identity
  const bool empty = local.error_code == UHDR_CODEC_OK && (sdr->h < scale || sdr->w < scale);
  GenParams p;
  int use_base_cg = 1;
  float hdr_white_nits = 0;
  bool run = false;
  if (local.error_code == UHDR_CODEC_OK && !empty) {
    if (!gm->planes[0]) note(err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for the gainmap stripe"));
    if (local.error_code == UHDR_CODEC_OK) note(fill_gen_params(c, sdr, hdr, cfg, &p, &use_base_cg, &hdr_white_nits));
    if (local.error_code == UHDR_CODEC_OK && p.scale != scale)
      note(err_status(UHDR_CODEC_INVALID_PARAM, "stripe %ux%u holds no map sample at scale factor %d", sdr->w, sdr->h, cfg->map_dimension_scale_factor));
    if (local.error_code == UHDR_CODEC_OK) {
      fill_gainmap_desc(hdr, p, gm);
      if (gm->stride[0] < gm->w) note(err_status(UHDR_CODEC_INVALID_PARAM, "gainmap stride (%u) cannot be less than its width (%u)", gm->stride[0], gm->w));
    }
    if (local.error_code == UHDR_CODEC_OK) {
      const size_t nfl = (size_t)p.map_w * p.map_h * (p.multichannel ? 3 : 1);
      note(ensure(c->scratch[7], nfl * sizeof(float)));
    }
    run = local.error_code == UHDR_CODEC_OK;
  } else if (empty) {
    use_base_cg = !(hdr->cg == UHDR_CG_BT_2100 || (hdr->cg == UHDR_CG_DISPLAY_P3 && sdr->cg != UHDR_CG_BT_2100)) || sdr->cg == hdr->cg;
    gm->w = sdr->w / scale;
    gm->h = 0;
  }
  // ---- phase 1: pass 1 of this stripe ----------------------------------------------------------------------------------------
  if (run) {
    p.gain_log2 = (float*)c->scratch[7].p;
    p.minmax = (float*)c->minmax.p;
    ProfScope ps(c, "generate_gainmap");
    const hipError_t e = launch_generate_gainmap(p, true, c->stream);  // float gain ratios + this stripe's ratio extrema
    note_hip(e, "generate_gainmap pass 1");
    if (e != hipSuccess) run = false;
  }
  // ---- phase 2: the exchange -- every rank gets here ----------------------------------------------------------------------
  uhdr_error_info_t xchg = ok_status();
  {
    ProfScope ps(c, "stripe_exchange");
    MinmaxTableParams t;  // this stripe's ratio extrema -> log2 extrema in the {min, -max} form of the single min-all-reduce
    memset(&t, 0, sizeof t);
    t.do_reduce = 1;
    t.partials = (const float*)c->minmax.p + 6;
    t.n_partials = run ? gen_partials_count(p) : 0;
    t.empty = run ? 0 : 1;
    t.mm6 = (float*)c->minmax.p;
    t.merged6 = merged;
    t.math_tab = c->d_math;
    note_hip(launch_minmax_table(t, c->stream), "minmax reduce");
    xchg = comm_all_reduce_min(c, merged, 6);
    MinmaxTableParams f;  // the merged range -> final range (jpegr.cpp:969-986) -> pass 2's step tables, on the device
    memset(&f, 0, sizeof f);
    f.do_finalize = f.do_table = 1;
    f.merged_in = merged;
    fill_finalize(&f, args_ok ? cfg : nullptr);
    f.out_mm = final_mm;
    f.dev = adev;
    f.math_tab = c->d_math;
    note_hip(launch_minmax_table(f, c->stream), "minmax finalize");
  }
  if (run && xchg.error_code == UHDR_CODEC_OK) {
    AffineParams a;
    memset(&a, 0, sizeof a);
    a.dev = adev;
    a.math_tab = c->d_math;
    a.gain_log2 = p.gain_log2;
    a.out = (uint8_t*)gm->planes[0];
    a.map_w = gm->w; a.map_h = gm->h; a.out_stride = gm->stride[0];
    a.nch = cfg->use_multi_channel_gainmap ? 3 : 1;
    a.gamma = cfg->gamma;
    ProfScope ps(c, "generate_gainmap");
    note_hip(launch_affine_map(a, c->stream), "generate_gainmap pass 2");
  }
  note_hip(hipMemcpyAsync(c->h_mm, final_mm, 9 * sizeof(float), hipMemcpyDeviceToHost, c->stream), "metadata copy");
  note_hip(hipStreamSynchronize(c->stream), "synchronize");  // the only host synchronisation: the metadata needs the merged range
  if (xchg.error_code != UHDR_CODEC_OK) return xchg;
  if (local.error_code != UHDR_CODEC_OK) return local;
  float mm[6];
  memcpy(mm, c->h_mm, sizeof mm);
  note_table_stats(c, cfg);
  // metadata from the already-final range (the clamp / hint / epsilon steps are idempotent on it)
  return uhdr_hip_generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
}

uhdr_error_info_t uhdr_hip_generate_gainmap(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                            const uhdr_hip_encode_cfg_t* cfg, uhdr_gainmap_metadata_t* md,
                                            uhdr_raw_image_t* gm) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!sdr || !hdr || !cfg || !md || !gm || !gm->planes[0]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  HIP_TRY(hipSetDevice(c->device));
  uhdr_raw_image_t dsdr, dhdr;
  UHDR_TRY(stage_in(c, 0, sdr, &dsdr, true));
  UHDR_TRY(stage_in(c, 1, hdr, &dhdr, true));
  // size the device gain map from the same rule the kernels use
  uint32_t scale = (uint32_t)(cfg->map_dimension_scale_factor < 1 ? 1 : cfg->map_dimension_scale_factor);
  uint32_t mw = sdr->w / scale, mh = sdr->h / scale;
  if (mw == 0 || mh == 0) {
    uint32_t s = sdr->w < sdr->h ? sdr->w : sdr->h;
    s = (s >= 8) ? (s / 8) : 1;
    mw = sdr->w / s; mh = sdr->h / s;
  }
  uhdr_raw_image_t tmp = *gm;
  tmp.fmt = cfg->use_multi_channel_gainmap ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_8bppYCbCr400;
  tmp.w = mw; tmp.h = mh;
  if (tmp.stride[0] < mw) return err_status(UHDR_CODEC_INVALID_PARAM, "gainmap stride (%u) cannot be less than its width (%u)", tmp.stride[0], mw);
  uhdr_raw_image_t dgm;
  UHDR_TRY(stage_in(c, 2, &tmp, &dgm, false));
  UHDR_TRY(uhdr_hip_generate_gainmap_dev(c, &dsdr, &dhdr, cfg, md, &dgm));
  void* host_plane = gm->planes[0];
  const unsigned host_stride = gm->stride[0];
  *gm = dgm;
  gm->planes[0] = host_plane; gm->planes[1] = gm->planes[2] = nullptr;
  gm->stride[0] = host_stride; gm->stride[1] = gm->stride[2] = 0;
  return stage_out(c, &dgm, gm);
}

// -------------------------------------------------------------------------------------------------
// toneMap
// -------------------------------------------------------------------------------------------------
// everything of ToneMapParams that depends on the HDR image only (the caller sets p->sdr)
static uhdr_error_info_t fill_tone_map_params(uhdr_hip_ctx* c, const uhdr_raw_image_t* hdr, ToneMapParams* pp) {
  ToneMapParams& p = *pp;
  memset(&p, 0, sizeof p);
  p.hdr = view_of(hdr);
  UHDR_TRY(select_hdr_lut(c, hdr->ct, &p.hdr_inv_lut, &p.hdr_inv_n));
  UHDR_TRY(upload_math(c));
  p.math_tab = c->d_math;
  if (!c->srgb8_meta.tab && !c->d_srgb8) UHDR_TRY(upload_step_table(host::srgb_code8_buckets(), &c->d_srgb8, &c->srgb8_meta, c->stream));
  p.srgb8 = c->srgb8_meta;
  if (hdr->fmt == UHDR_IMG_FMT_32bppRGBA1010102 && (int)hdr->ct >= 0 && (int)hdr->ct < 5) {
    if (!c->d_lin10[hdr->ct]) {
      const std::vector<float>* src = nullptr;
      if (hdr->ct == UHDR_CT_HLG) src = &host::hlg_inv_oetf_ootf_lut();
      else if (hdr->ct == UHDR_CT_PQ) src = &host::pq_inv_oetf_lut();
      else if (hdr->ct == UHDR_CT_SRGB) src = &host::srgb_inv_oetf_lut();
      UHDR_TRY(upload_lut(&c->d_lin10[hdr->ct], host::lin10_table(src ? src->data() : nullptr, src ? (int)src->size() : 0), c->stream));
    }
    p.lin10 = c->d_lin10[hdr->ct];
  }
  p.hdr_is_rgb = is_rgb_fmt_host(hdr->fmt);
  p.is_normalized = hdr->ct != UHDR_CT_LINEAR;
  p.headroom = host::reference_peak_nits(hdr->ct) / 203.0f;
  p.headroom_sq = p.headroom * p.headroom;
  p.headroom_sq_rcp = 1.0f / p.headroom_sq;
  bool identity;
  host::gamut_matrix(UHDR_CG_DISPLAY_P3, hdr->cg, &p.gamut, &identity);
  p.gamut_on = identity ? 0 : 1;
  p.hdr_yuv = host::yuv2rgb_coeffs(hdr->cg);
  p.p3 = host::rgb2yuv_coeffs(UHDR_CG_DISPLAY_P3);
  return ok_status();
}

uhdr_error_info_t uhdr_hip_tone_map_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* hdr, uhdr_raw_image_t* sdr) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!hdr || !sdr) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  // checks: jpegr.cpp:1986-2103
  if (hdr->fmt != UHDR_IMG_FMT_24bppYCbCrP010 && hdr->fmt != UHDR_IMG_FMT_30bppYCbCr444 &&
      hdr->fmt != UHDR_IMG_FMT_32bppRGBA1010102 && hdr->fmt != UHDR_IMG_FMT_64bppRGBAHalfFloat)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "tonemap method expects hdr intent color format to be one of "
                      "{UHDR_IMG_FMT_24bppYCbCrP010, UHDR_IMG_FMT_30bppYCbCr444, UHDR_IMG_FMT_32bppRGBA1010102, "
                      "UHDR_IMG_FMT_64bppRGBAHalfFloat}. Received %d", hdr->fmt);
  if (hdr->fmt == UHDR_IMG_FMT_24bppYCbCrP010 && sdr->fmt != UHDR_IMG_FMT_12bppYCbCr420)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "tonemap method expects sdr intent color format to be UHDR_IMG_FMT_12bppYCbCr420, if "
                      "hdr intent color format is UHDR_IMG_FMT_24bppYCbCrP010. Received %d", sdr->fmt);
  if (hdr->fmt == UHDR_IMG_FMT_30bppYCbCr444 && sdr->fmt != UHDR_IMG_FMT_24bppYCbCr444)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "tonemap method expects sdr intent color format to be UHDR_IMG_FMT_24bppYCbCr444, if "
                      "hdr intent color format is UHDR_IMG_FMT_30bppYCbCr444. Received %d", sdr->fmt);
  if ((hdr->fmt == UHDR_IMG_FMT_32bppRGBA1010102 || hdr->fmt == UHDR_IMG_FMT_64bppRGBAHalfFloat) &&
      sdr->fmt != UHDR_IMG_FMT_32bppRGBA8888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "tonemap method expects sdr intent color format to be UHDR_IMG_FMT_32bppRGBA8888, if "
                      "hdr intent color format is UHDR_IMG_FMT_32bppRGBA1010102 or UHDR_IMG_FMT_64bppRGBAHalfFloat. Received %d", sdr->fmt);
  if (hdr->cg < UHDR_CG_BT_709 || hdr->cg > UHDR_CG_BT_2100)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for converting yuv to rgb for color gamut %d", hdr->cg);
  if (hdr->ct < UHDR_CT_LINEAR || hdr->ct > UHDR_CT_SRGB)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for calculating Ootf for color transfer %d", hdr->ct);
  if (sdr->w != hdr->w || sdr->h != hdr->h)
    return err_status(UHDR_CODEC_INVALID_PARAM, "sdr intent resolution %ux%u and hdr intent resolution %ux%u do not match",
                      sdr->w, sdr->h, hdr->w, hdr->h);
  UHDR_TRY(validate_image(hdr, "hdr intent"));
  UHDR_TRY(validate_image(sdr, "sdr intent"));
  HIP_TRY(hipSetDevice(c->device));
  sdr->cg = UHDR_CG_DISPLAY_P3;
  sdr->ct = UHDR_CT_SRGB;
  sdr->range = UHDR_CR_FULL_RANGE;
  ToneMapParams p;
  UHDR_TRY(fill_tone_map_params(c, hdr, &p));
  p.sdr = view_mut_of(sdr);
  ProfScope ps(c, "tone_map");
  HIP_TRY(launch_tone_map(p, c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_tone_map(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* hdr, uhdr_raw_image_t* sdr) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!hdr || !sdr) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  HIP_TRY(hipSetDevice(c->device));
  uhdr_raw_image_t dh, ds;
  UHDR_TRY(stage_in(c, 0, hdr, &dh, true));
  UHDR_TRY(stage_in(c, 1, sdr, &ds, false));
  UHDR_TRY(uhdr_hip_tone_map_dev(c, &dh, &ds));
  sdr->cg = ds.cg; sdr->ct = ds.ct; sdr->range = ds.range;
  return stage_out(c, &ds, sdr);
}

// -------------------------------------------------------------------------------------------------
// convertYuv / convert_raw_input_to_ycbcr
// -------------------------------------------------------------------------------------------------
uhdr_error_info_t uhdr_hip_convert_yuv_dev(uhdr_hip_ctx_t* c, uhdr_raw_image_t* img, uhdr_color_gamut_t src, uhdr_color_gamut_t dst) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!img) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  YuvXformParams p;
  const int r = host::yuv_encoding_matrix(src, dst, &p.c);
  if (r == -1) return err_status(UHDR_CODEC_INVALID_PARAM, "Unrecognized src color gamut %d", src);
  if (r == -2) return err_status(UHDR_CODEC_INVALID_PARAM, "Unrecognized dest color gamut %d", dst);
  if (r == 1) return ok_status();
  if (img->fmt != UHDR_IMG_FMT_12bppYCbCr420 && img->fmt != UHDR_IMG_FMT_24bppYCbCr444)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for performing gamut conversion for color format %d", img->fmt);
  UHDR_TRY(validate_image(img, "yuv"));
  HIP_TRY(hipSetDevice(c->device));
  p.img = view_mut_of(img);
  ProfScope ps(c, "convert_yuv");
  HIP_TRY(launch_transform_yuv(p, c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_convert_yuv(uhdr_hip_ctx_t* c, uhdr_raw_image_t* img, uhdr_color_gamut_t src, uhdr_color_gamut_t dst) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!img) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  HIP_TRY(hipSetDevice(c->device));
  uhdr_raw_image_t d;
  UHDR_TRY(stage_in(c, 0, img, &d, true));
  UHDR_TRY(uhdr_hip_convert_yuv_dev(c, &d, src, dst));
  return stage_out(c, &d, img);
}

uhdr_error_info_t uhdr_hip_convert_raw_input_to_ycbcr_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* src, int chroma, uhdr_raw_image_t* dst) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!src || !dst) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  if (src->fmt != UHDR_IMG_FMT_32bppRGBA1010102 && src->fmt != UHDR_IMG_FMT_32bppRGBA8888 && src->fmt != UHDR_IMG_FMT_24bppRGB888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "convert_raw_input_to_ycbcr on the device handles RGB inputs; format %d is a plain copy in the reference", src->fmt);
  if (src->cg < UHDR_CG_BT_709 || src->cg > UHDR_CG_BT_2100)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "unrecognized color gamut %d", src->cg);
  HIP_TRY(hipSetDevice(c->device));
  const bool ten = src->fmt == UHDR_IMG_FMT_32bppRGBA1010102;
  dst->fmt = ten ? (chroma ? UHDR_IMG_FMT_24bppYCbCrP010 : UHDR_IMG_FMT_30bppYCbCr444)
                 : (chroma ? UHDR_IMG_FMT_12bppYCbCr420 : UHDR_IMG_FMT_24bppYCbCr444);
  dst->cg = src->cg; dst->ct = src->ct; dst->range = UHDR_CR_FULL_RANGE;
  dst->w = src->w; dst->h = src->h;
  UHDR_TRY(validate_image(src, "source"));
  UHDR_TRY(validate_image(dst, "destination"));
  RgbToYcbcrParams p;
  p.src = view_of(src);
  p.dst = view_mut_of(dst);
  p.k = host::rgb2yuv_coeffs(src->cg);
  ProfScope ps(c, "convert_raw_input_to_ycbcr");
  HIP_TRY(launch_rgb_to_ycbcr(p, c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_convert_raw_input_to_ycbcr(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* src, int chroma, uhdr_raw_image_t* dst) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!src || !dst) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  HIP_TRY(hipSetDevice(c->device));
  const bool ten = src->fmt == UHDR_IMG_FMT_32bppRGBA1010102;
  uhdr_raw_image_t tmp = *dst;
  tmp.fmt = ten ? (chroma ? UHDR_IMG_FMT_24bppYCbCrP010 : UHDR_IMG_FMT_30bppYCbCr444)
                : (chroma ? UHDR_IMG_FMT_12bppYCbCr420 : UHDR_IMG_FMT_24bppYCbCr444);
  tmp.w = src->w; tmp.h = src->h;
  uhdr_raw_image_t ds, dd;
  UHDR_TRY(stage_in(c, 0, src, &ds, true));
  UHDR_TRY(stage_in(c, 1, &tmp, &dd, false));
  UHDR_TRY(uhdr_hip_convert_raw_input_to_ycbcr_dev(c, &ds, chroma, &dd));
  dst->fmt = dd.fmt; dst->cg = dd.cg; dst->ct = dd.ct; dst->range = dd.range; dst->w = dd.w; dst->h = dd.h;
  return stage_out(c, &dd, dst);
}

// -------------------------------------------------------------------------------------------------
// JPEG FDCT + quantize
// -------------------------------------------------------------------------------------------------
void uhdr_hip_jpeg_quant_table(int quality, int is_chroma, uint16_t qt[64]) { host::jpeg_quant_table(quality, is_chroma, qt); }

int uhdr_hip_oetf_code_thresholds(uhdr_color_transfer_t ct, float thresholds[1024]) {
  if ((ct != UHDR_CT_HLG && ct != UHDR_CT_PQ) || !thresholds) return -1;
  const std::vector<float>& t = host::oetf_code_thresholds(ct);
  for (int i = 0; i < 1024; i++) thresholds[i] = t[(size_t)i];
  return 0;
}

uhdr_error_info_t uhdr_hip_selftest(uhdr_hip_ctx_t* c, int which, unsigned int arg0, unsigned int arg1, unsigned int seed, const float mm[6],
                                    unsigned long long out[8]) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!out || which < 0 || which > 4 || (which == 4 && (!mm || arg0 > 2 || arg1 < 1 || arg1 > 3 || arg0 >= arg1)) || (which == 2 && (arg0 < 1 || arg1 > 254 || arg0 > arg1)))
    return err_status(UHDR_CODEC_INVALID_PARAM, "bad self-test arguments");
  HIP_TRY(hipSetDevice(c->device));
  UHDR_TRY(upload_math(c));
  UHDR_TRY(ensure(c->affine, kAffineDevBytes));
  UHDR_TRY(ensure(c->exchange, 256));
  unsigned long long* d_out = (unsigned long long*)c->exchange.p;
  HIP_TRY(hipMemsetAsync(d_out, 0, 64, c->stream));
  if (which == 4) {
    MinmaxTableParams t;
    memset(&t, 0, sizeof t);
    t.do_table = 1;
    t.nch = (int)arg1;
    t.gamma = 1.0f;
    for (int i = 0; i < 6; i++) t.final_mm[i] = mm[i];
    t.dev = (AffineDev*)c->affine.p;
    t.math_tab = c->d_math;
    HIP_TRY(launch_minmax_table(t, c->stream));
  }
  HIP_TRY(launch_selftest(which, d_out, arg0, arg1, seed, c->d_math, (const AffineDev*)c->affine.p, c->stream));
  HIP_TRY(hipMemcpyAsync(out, d_out, 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

int uhdr_hip_exact_math_eval(int fn, const float* in, float* out, size_t n) {
  if (!in || !out || fn < 0 || fn > 5) return -1;
  const double* T = host::math_tables().data();
  if (fn == 2) {  // in[0] = the constant divisor b; out[i] = div_const(in[i], b, 1/b) for i >= 1
    if (n < 1) return -1;
    const float b = in[0], rb = 1.0f / b;
    out[0] = rb;
    for (size_t i = 1; i < n; i++) out[i] = div_const(in[i], b, rb);
    return 0;
  }
  if (fn == 4) {  // pairs (a, b): out[2i] = a / b through rcp64_of_f32 seeded with a 2-ulp-off float reciprocal
    for (size_t i = 0; i + 1 < n; i += 2) {
      const float b = in[i + 1];
      const float seed = nextafterf(nextafterf(1.0f / b, INFINITY), INFINITY);  // worse than v_rcp_f32's 1 ulp
      out[i] = div_by_rcp64(in[i], rcp64_of_f32(b, seed));
      out[i + 1] = seed;
    }
    return 0;
  }
  if (fn == 3) {  // in[0] = any divisor b; out[i] = div_by_rcp64(in[i], 1.0 / (double)b) for i >= 1
    if (n < 1) return -1;
    const double rbd = 1.0 / (double)in[0];
    out[0] = (float)rbd;
    for (size_t i = 1; i < n; i++) out[i] = div_by_rcp64(in[i], rbd);
    return 0;
  }
  if (fn == 5) {  // the round-4 form of srgbOetf: direct pow table (exact_math.h: srgb_oetf_direct)
    for (size_t i = 0; i < n; i++) out[i] = srgb_oetf_direct(in[i], T + kPowDirOff);
    return 0;
  }
  for (size_t i = 0; i < n; i++) out[i] = fn == 0 ? srgb_oetf_table(in[i], T) : (float)log2_table_f64(in[i], T);
  return 0;
}

// -------------------------------------------------------------------------------------------------
// image effects (editorhelper.cpp:210-520)
// -------------------------------------------------------------------------------------------------
uhdr_error_info_t uhdr_hip_apply_effect_dev(uhdr_hip_ctx_t* c, int effect, int p0, int p1, const uhdr_raw_image_t* src, uhdr_raw_image_t* dst) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!src || !dst) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  UHDR_TRY(validate_image(src, "source"));
  if (dst->fmt != src->fmt) return err_status(UHDR_CODEC_INVALID_PARAM, "effect destination format %d differs from the source format %d", dst->fmt, src->fmt);
  UHDR_TRY(validate_image(dst, "destination"));
  uint32_t mode = 0, a0 = 0, a1 = 0;
  const uint32_t sw = src->w, sh = src->h, dw = dst->w, dh = dst->h;
  switch (effect) {
    case 0:
      if (p0 == 90 || p0 == 270) {
        if (dw != sh || dh != sw) return err_status(UHDR_CODEC_INVALID_PARAM, "rotation by %d degrees of a %ux%u image needs a %ux%u destination", p0, sw, sh, sh, sw);
        mode = p0 == 90 ? 0u : 2u;
      } else if (p0 == 180) {
        if (dw != sw || dh != sh) return err_status(UHDR_CODEC_INVALID_PARAM, "rotation by 180 degrees keeps the image size");
        mode = 1;
      } else {
        return err_status(UHDR_CODEC_INVALID_PARAM, "unsupported degrees, expects one of {90, 180, 270}");  // ultrahdr_api.cpp uhdr_add_effect_rotate
      }
      break;
    case 1:
      if (p0 != 0 && p0 != 1) return err_status(UHDR_CODEC_INVALID_PARAM, "unsupported direction, expects one of {UHDR_MIRROR_HORIZONTAL, UHDR_MIRROR_VERTICAL}");
      if (dw != sw || dh != sh) return err_status(UHDR_CODEC_INVALID_PARAM, "mirroring keeps the image size");
      mode = p0 == 0 ? 3u : 4u;
      break;
    case 2:
      if (p0 < 0 || p1 < 0 || (uint64_t)p0 + dw > sw || (uint64_t)p1 + dh > sh)
        return err_status(UHDR_CODEC_INVALID_PARAM, "crop window %ux%u at (%d, %d) leaves the %ux%u image", dw, dh, p0, p1, sw, sh);
      mode = 5; a0 = (uint32_t)p0; a1 = (uint32_t)p1;
      break;
    case 3:
      if (dw == 0 || dh == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "resize to an empty image");
      mode = 6; a0 = sw / dw; a1 = sh / dh;
      break;
    default:
      return err_status(UHDR_CODEC_INVALID_PARAM, "unknown effect %d", effect);
  }
  HIP_TRY(hipSetDevice(c->device));
  // the planes as the reference walks them: element size, geometry in elements (P010 chroma: one 4-byte element per U, V pair)
  struct Pl { int idx; uint32_t elem, div, stride_div; };
  Pl pls[3];
  int npl = 0;
  switch (src->fmt) {
    case UHDR_IMG_FMT_24bppYCbCrP010: pls[0] = {0, 2, 1, 1}; pls[1] = {1, 4, 2, 2}; npl = 2; break;
    case UHDR_IMG_FMT_12bppYCbCr420: pls[0] = {0, 1, 1, 1}; pls[1] = {1, 1, 2, 1}; pls[2] = {2, 1, 2, 1}; npl = 3; break;
    case UHDR_IMG_FMT_8bppYCbCr400: pls[0] = {0, 1, 1, 1}; npl = 1; break;
    case UHDR_IMG_FMT_24bppYCbCr444: for (int i = 0; i < 3; i++) pls[i] = {i, 1, 1, 1}; npl = 3; break;
    case UHDR_IMG_FMT_30bppYCbCr444: for (int i = 0; i < 3; i++) pls[i] = {i, 2, 1, 1}; npl = 3; break;
    case UHDR_IMG_FMT_32bppRGBA8888:
    case UHDR_IMG_FMT_32bppRGBA1010102: pls[0] = {0, 4, 1, 1}; npl = 1; break;
    case UHDR_IMG_FMT_64bppRGBAHalfFloat: pls[0] = {0, 8, 1, 1}; npl = 1; break;
    default:
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "image effects are not implemented for color format %d", src->fmt);
  }
  ProfScope ps(c, "effect");
  for (int k = 0; k < npl; k++) {
    EffectPlane e;
    e.src = src->planes[pls[k].idx];
    e.dst = dst->planes[pls[k].idx];
    e.elem = pls[k].elem;
    e.src_w = sw / pls[k].div; e.src_h = sh / pls[k].div; e.src_stride = src->stride[pls[k].idx] / pls[k].stride_div;
    e.dst_w = dw / pls[k].div; e.dst_h = dh / pls[k].div; e.dst_stride = dst->stride[pls[k].idx] / pls[k].stride_div;
    e.mode = mode;
    e.a0 = mode == 5 ? a0 / pls[k].div : (mode == 6 ? e.src_w / (e.dst_w ? e.dst_w : 1) : 0);
    e.a1 = mode == 5 ? a1 / pls[k].div : (mode == 6 ? e.src_h / (e.dst_h ? e.dst_h : 1) : 0);
    HIP_TRY(launch_effect_plane(e, c->stream));
  }
  return ok_status();
}

uhdr_error_info_t uhdr_hip_apply_effect(uhdr_hip_ctx_t* c, int effect, int p0, int p1, const uhdr_raw_image_t* src, uhdr_raw_image_t* dst) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!src || !dst) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  HIP_TRY(hipSetDevice(c->device));
  uhdr_raw_image_t dsrc, ddst;
  UHDR_TRY(stage_in(c, 0, src, &dsrc, true));
  UHDR_TRY(stage_in(c, 1, dst, &ddst, false));
  UHDR_TRY(uhdr_hip_apply_effect_dev(c, effect, p0, p1, &dsrc, &ddst));
  return stage_out(c, &ddst, dst);
}

int uhdr_hip_step_table_eval(int which, float a, float b, const float* in, uint32_t* out, size_t n, uint32_t info[4]) {
  host::OetfBuckets tmp;
  const host::OetfBuckets* t = nullptr;
  if (which == 0) t = &host::srgb_code8_buckets();
  else if (which == 1) {
    const float l2min = log2f(a), l2max = log2f(b);
    const double range = (double)(l2max - l2min);
    tmp = host::gain_code8_buckets(a, b, l2min, range, 1.0 / range);
    t = &tmp;
  } else if (which == 2 || which == 3) t = &host::oetf_code_buckets(which == 2 ? UHDR_CT_HLG : UHDR_CT_PQ);
  else if (which == 4 || which == 5) t = &host::oetf_code_buckets(which == 4 ? UHDR_CT_HLG : UHDR_CT_PQ, true);
  else if (which == 6) {
    // a synthetic staircase over [0, 1] whose steps sit EXACTLY on bucket starts (bucket = 2^15 bit patterns): the corner the
    // builder answers with an empty bucket in front of the first threshold (clamp_lo_bits, host_tables.cpp)
    uint32_t first, step;
    memcpy(&first, &a, 4);
    memcpy(&step, &b, 4);
    first &= ~0x7fffu;
    step = (step >> 15) ? (step & ~0x7fffu) : (1u << 15);
    tmp = host::build_step_table([=](uint32_t u) { return u < first ? 0u : 1u + (u - first) / step; }, 0u, 0x3f800000u, 15, 65536);
    t = &tmp;
  }
  if (!t) return -1;
  if (info) { info[0] = t->exact ? 1u : 0u; info[1] = t->n; info[2] = t->shift; info[3] = t->base; }
  if (!t->exact) return 1;
  for (size_t i = 0; i < n; i++) {  // the kernels' step_code / oetf_code_bucket, on the host
    uint32_t bits;
    memcpy(&bits, &in[i], 4);
    int ib = (int)bits;
    ib = ib < (int)t->clamp_lo_bits ? (int)t->clamp_lo_bits : (ib > (int)t->hi_bits ? (int)t->hi_bits : ib);
    const uint32_t u = (uint32_t)ib;
    const uint32_t off = ((u >> t->shift) - t->base) * 8;  // never negative: clamp_lo_bits >= base << shift
    const uint32_t thr = t->entries[off / 4], cc = t->entries[off / 4 + 1];
    out[i] = u >= thr ? cc >> 16 : cc & 0xffffu;
  }
  return 0;
}

uhdr_error_info_t uhdr_hip_fdct_quant_dev(uhdr_hip_ctx_t* c, const uint8_t* plane, size_t stride, int bw, int bh,
                                          const uint16_t qt[64], int16_t* coef) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!plane || !qt || !coef || bw <= 0 || bh <= 0) return err_status(UHDR_CODEC_INVALID_PARAM, "received bad argument for fdct_quant");
  if (((uintptr_t)coef & 15) != 0) return err_status(UHDR_CODEC_INVALID_PARAM, "coefficient buffer must be 16-byte aligned");
  for (int i = 0; i < 64; i++)
    if (qt[i] == 0 || qt[i] > 255) return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d out of baseline range", i);
  HIP_TRY(hipSetDevice(c->device));
  ProfScope ps(c, "fdct_quant");
  HIP_TRY(launch_fdct_quant(plane, stride, bw, bh, qt, coef, c->stream));  // the table travels in the kernel arguments
  return ok_status();
}

uhdr_error_info_t uhdr_hip_fdct_quant(uhdr_hip_ctx_t* c, const uint8_t* plane, size_t stride, int bw, int bh,
                                      const uint16_t qt[64], int16_t* coef) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!plane || !qt || !coef || bw <= 0 || bh <= 0) return err_status(UHDR_CODEC_INVALID_PARAM, "received bad argument for fdct_quant");
  HIP_TRY(hipSetDevice(c->device));
  const size_t in_bytes = ((size_t)bh * 8 - 1) * stride + (size_t)bw * 8;
  const size_t out_bytes = (size_t)bw * bh * 64 * sizeof(int16_t);
  UHDR_TRY(ensure(c->scratch[0], in_bytes));
  UHDR_TRY(ensure(c->scratch[1], out_bytes));
  HIP_TRY(hipMemcpyAsync(c->scratch[0].p, plane, in_bytes, hipMemcpyHostToDevice, c->stream));
  UHDR_TRY(uhdr_hip_fdct_quant_dev(c, (const uint8_t*)c->scratch[0].p, stride, bw, bh, qt, (int16_t*)c->scratch[1].p));
  HIP_TRY(hipMemcpyAsync(coef, c->scratch[1].p, out_bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

// -------------------------------------------------------------------------------------------------
// 3-channel gain map: libjpeg's RGB -> YCbCr + FDCT + quantize of all three components in one pass
// -------------------------------------------------------------------------------------------------
uhdr_error_info_t uhdr_hip_fdct_quant_rgb_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* rgb, const uint16_t qt_luma[64],
                                              const uint16_t qt_chroma[64], int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!rgb || !rgb->planes[0] || !qt_luma || !qt_chroma || !coef_y || !coef_cb || !coef_cr)
    return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  if (rgb->fmt != UHDR_IMG_FMT_24bppRGB888 && rgb->fmt != UHDR_IMG_FMT_32bppRGBA8888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "fdct_quant_rgb expects UHDR_IMG_FMT_24bppRGB888 or UHDR_IMG_FMT_32bppRGBA8888. Received %d", rgb->fmt);
  const int bpp = rgb->fmt == UHDR_IMG_FMT_32bppRGBA8888 ? 4 : 3;
  const size_t pitch = (size_t)rgb->stride[0] * bpp, al = bpp == 4 ? 16 : 8;
  if (rgb->w == 0 || rgb->h == 0 || rgb->w % 8 || rgb->h % 8)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "fdct_quant_rgb needs dimensions that are multiples of 8 (received %ux%u); "
                      "pad to the MCU grid as jpegencoderhelper.cpp:246-309 does, or use jpeg_rgb_to_ycc + fdct_quant", rgb->w, rgb->h);
  if (rgb->stride[0] < rgb->w) return err_status(UHDR_CODEC_INVALID_PARAM, "stride (%u) cannot be less than width (%u)", rgb->stride[0], rgb->w);
  if (pitch % al || ((uintptr_t)rgb->planes[0] % al))
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "fdct_quant_rgb needs %zu-byte aligned rows; use jpeg_rgb_to_ycc + fdct_quant", al);
  if (((uintptr_t)coef_y | (uintptr_t)coef_cb | (uintptr_t)coef_cr) & 15)
    return err_status(UHDR_CODEC_INVALID_PARAM, "coefficient buffers must be 16-byte aligned");
  for (int i = 0; i < 64; i++)
    if (qt_luma[i] == 0 || qt_luma[i] > 255 || qt_chroma[i] == 0 || qt_chroma[i] > 255)
      return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d out of baseline range", i);
  HIP_TRY(hipSetDevice(c->device));
  ProfScope ps(c, "fdct_quant");
  HIP_TRY(launch_fdct_quant_rgb((const uint8_t*)rgb->planes[0], pitch, bpp, (int)(rgb->w / 8), (int)(rgb->h / 8), qt_luma, qt_chroma,
                                coef_y, coef_cb, coef_cr, c->stream));
  return ok_status();
}

// -------------------------------------------------------------------------------------------------
// API-0 front end fused: toneMap + generateGainMap + convert_raw_input_to_ycbcr(4:4:4) in one pass
// -------------------------------------------------------------------------------------------------
uhdr_error_info_t uhdr_hip_encode_api0_fused_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                                 uhdr_raw_image_t* sdr_rgba, uhdr_raw_image_t* base_ycc, uhdr_gainmap_metadata_t* md,
                                                 uhdr_raw_image_t* gm) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!hdr || !cfg || !base_ycc || !md || !gm || !gm->planes[0]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  if (hdr->fmt != UHDR_IMG_FMT_32bppRGBA1010102 && hdr->fmt != UHDR_IMG_FMT_64bppRGBAHalfFloat)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-0 front end takes UHDR_IMG_FMT_32bppRGBA1010102 or UHDR_IMG_FMT_64bppRGBAHalfFloat "
                      "(the inputs toneMap renders to RGBA8888). Received %d", hdr->fmt);
  if (cfg->map_dimension_scale_factor != 1)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-0 front end needs a full-resolution gain map (scale factor 1), received %d; "
                      "use tone_map + generate_gainmap + convert_raw_input_to_ycbcr", cfg->map_dimension_scale_factor);
  if (hdr->cg < UHDR_CG_BT_709 || hdr->cg > UHDR_CG_BT_2100)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for color gamut %d", hdr->cg);
  if (hdr->ct < UHDR_CT_LINEAR || hdr->ct > UHDR_CT_SRGB)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "No implementation available for color transfer %d", hdr->ct);
  for (int i = 0; i < 3; i++) {
    if (!base_ycc->planes[i]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for base image plane %d", i);
    if (base_ycc->stride[i] < hdr->w) return err_status(UHDR_CODEC_INVALID_PARAM, "base image stride (%u) cannot be less than width (%u)", base_ycc->stride[i], hdr->w);
  }
  if (sdr_rgba && sdr_rgba->planes[0] && sdr_rgba->stride[0] < hdr->w)
    return err_status(UHDR_CODEC_INVALID_PARAM, "sdr stride (%u) cannot be less than width (%u)", sdr_rgba->stride[0], hdr->w);
  HIP_TRY(hipSetDevice(c->device));
  // the SDR rendition toneMap would hand to generateGainMap: RGBA8888, Display-P3, sRGB, full range
  uhdr_raw_image_t sdr_desc;
  memset(&sdr_desc, 0, sizeof sdr_desc);
  if (sdr_rgba) sdr_desc = *sdr_rgba;
  sdr_desc.fmt = UHDR_IMG_FMT_32bppRGBA8888; sdr_desc.cg = UHDR_CG_DISPLAY_P3; sdr_desc.ct = UHDR_CT_SRGB; sdr_desc.range = UHDR_CR_FULL_RANGE;
  sdr_desc.w = hdr->w; sdr_desc.h = hdr->h;
  if (!sdr_desc.planes[0]) sdr_desc.stride[0] = hdr->w;
  if (sdr_rgba) { sdr_rgba->fmt = sdr_desc.fmt; sdr_rgba->cg = sdr_desc.cg; sdr_rgba->ct = sdr_desc.ct; sdr_rgba->range = sdr_desc.range; sdr_rgba->w = hdr->w; sdr_rgba->h = hdr->h; }
  FusedParams p;
  UHDR_TRY(fill_tone_map_params(c, hdr, &p.tm));
  p.tm.sdr = view_mut_of(&sdr_desc);
  int use_base_cg = 1;
  float hdr_white_nits;
  UHDR_TRY(fill_gen_params(c, &sdr_desc, hdr, cfg, &p.gen, &use_base_cg, &hdr_white_nits, /*sdr_in_registers=*/true));
  UHDR_TRY(upload_lut(&c->d_srgb_of_byte, host::srgb_inv_oetf_of_byte(), c->stream));
  p.gen.srgb_of_byte = c->d_srgb_of_byte;
  fill_gainmap_desc(hdr, p.gen, gm);
  if (gm->stride[0] < gm->w) return err_status(UHDR_CODEC_INVALID_PARAM, "gainmap stride (%u) cannot be less than its width (%u)", gm->stride[0], gm->w);
  base_ycc->fmt = UHDR_IMG_FMT_24bppYCbCr444; base_ycc->cg = UHDR_CG_DISPLAY_P3; base_ycc->ct = UHDR_CT_SRGB; base_ycc->range = UHDR_CR_FULL_RANGE;
  base_ycc->w = hdr->w; base_ycc->h = hdr->h;
  p.ycc = view_mut_of(base_ycc);
  p.base_k = host::rgb2yuv_coeffs(UHDR_CG_DISPLAY_P3);
  if (cfg->preset == UHDR_USAGE_REALTIME) {  // one pass: jpegr.cpp:724-737
    for (int i = 0; i < 3; i++) {
      md->max_content_boost[i] = hdr_white_nits / 203.0f;
      md->min_content_boost[i] = 1.0f;
      md->gamma[i] = cfg->gamma;
      md->offset_sdr[i] = 0.0f;
      md->offset_hdr[i] = 0.0f;
    }
    md->hdr_capacity_min = 1.0f;
    md->hdr_capacity_max = cfg->target_disp_peak_nits != -1.0f ? cfg->target_disp_peak_nits / 203.0f : md->max_content_boost[0];
    md->use_base_cg = use_base_cg;
    p.gen.min_boost = md->min_content_boost[0];
    p.gen.max_boost = md->max_content_boost[0];
    p.gen.log2min = log2f(md->min_content_boost[0]);
    p.gen.log2max = log2f(md->max_content_boost[0]);
    p.gen.log2_range = (double)(p.gen.log2max - p.gen.log2min);
    p.gen.log2_range_rcp = 1.0 / p.gen.log2_range;
    UHDR_TRY(gain_step_table(c, p.gen, &p.gen.gain8));
    p.gen.out = (uint8_t*)gm->planes[0];
    p.gen.out_stride = gm->stride[0];
    ProfScope ps(c, "encode_api0_fused");
    HIP_TRY(launch_encode_api0_fused(p, false, nullptr, c->stream));
    return ok_status();
  }
  const size_t nfl = (size_t)p.gen.map_w * p.gen.map_h * (p.gen.multichannel ? 3 : 1);
  UHDR_TRY(ensure(c->scratch[7], nfl * sizeof(float)));
  UHDR_TRY(ensure(c->minmax, (6 + 2048 * 6) * sizeof(float)));
  p.gen.gain_log2 = (float*)c->scratch[7].p;
  p.gen.minmax = (float*)c->minmax.p;
  int grid = 0;
  {
    ProfScope ps(c, "encode_api0_fused");
    HIP_TRY(launch_encode_api0_fused(p, true, &grid, c->stream));
  }
  UHDR_TRY(two_pass_tail(c, p.gen, grid, cfg, gm));
  HIP_TRY(hipStreamSynchronize(c->stream));
  float mm[6];
  memcpy(mm, c->h_mm, sizeof mm);
  note_table_stats(c, cfg);
  return uhdr_hip_generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
}

// -------------------------------------------------------------------------------------------------
// API-1 encode chain fused (encode_api1_fused.hip): pass 1 -> range + tables -> map blocks; base blocks
// -------------------------------------------------------------------------------------------------
// With a communicator on the context (uhdr_hip_comm_init / _init_custom) the images are this rank's ROW STRIPE and the extrema
// are merged across ranks between the passes, exactly as in uhdr_hip_generate_gainmap_striped_dev: reduce -> ONE all-reduce(min)
// over {min, -max} -> finalize + tables.  Every rank takes part in that exchange whatever happens locally (a rank that
// failed validation, or whose stripe is empty -- h == 0 --, contributes the identity), and reports its error afterwards.
uhdr_error_info_t uhdr_hip_encode_api1_fused_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                                 const uhdr_hip_encode_cfg_t* cfg, uhdr_color_gamut_t base_encoding,
                                                 const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                                 const uhdr_hip_api1_blocks_t* blocks, uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gm) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipSetDevice(c->device));
  const bool striped = c->comm != nullptr || c->comm_custom;
  GenParams p;
  int use_base_cg = 1, nch = 1;
  float hdr_white_nits = 0;
  uint8_t* map_out = nullptr;
  uint32_t map_stride = 0;
  Mat3 conv;
  bool convert = false, empty = false;
  auto prepare = [&]() -> uhdr_error_info_t {
    if (!sdr || !hdr || !cfg || !qt_base || !qt_map || !blocks || !md) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
    if (sdr->fmt != UHDR_IMG_FMT_12bppYCbCr420 || sdr->w % 16 || sdr->h % 16 || sdr->w == 0)
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain takes a UHDR_IMG_FMT_12bppYCbCr420 base image whose dimensions are multiples of 16 "
                        "(received format %d, %ux%u); use the operators", sdr->fmt, sdr->w, sdr->h);
    if (cfg->preset == UHDR_USAGE_REALTIME) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain is the two-pass (best quality) encode; one pass: generate_gainmap + fdct_quant");
    if (cfg->gamma != 1.0f) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain needs gain-map gamma 1 (received %f); use the operators", cfg->gamma);
    for (int t = 0; t < 2; t++)
      for (int i = 0; i < 64; i++)
        if (qt_base[t][i] == 0 || qt_base[t][i] > 255 || qt_map[t][i] == 0 || qt_map[t][i] > 255)
          return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d out of baseline range", i);
    if (striped && sdr->h == 0 && hdr->h == 0) {  // a rank without rows: nothing to launch, the identity to contribute
      empty = true;
      use_base_cg = !(hdr->cg == UHDR_CG_BT_2100 || (hdr->cg == UHDR_CG_DISPLAY_P3 && sdr->cg != UHDR_CG_BT_2100)) || sdr->cg == hdr->cg;
      return ok_status();
    }
    if (sdr->h == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "empty base image");
    UHDR_TRY(fill_gen_params(c, sdr, hdr, cfg, &p, &use_base_cg, &hdr_white_nits));
    if (p.scale != (uint32_t)cfg->map_dimension_scale_factor || p.map_w % 8 || p.map_h % 8)
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain needs map dimensions that are multiples of 8 (%ux%u at scale factor %u); use the operators",
                        p.map_w, p.map_h, p.scale);
    nch = p.multichannel ? 3 : 1;
    if (((uintptr_t)sdr->planes[0] | sdr->stride[0]) & 1) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain reads luma in 16-bit pairs: even base address and stride");
    for (int i = 0; i < 3; i++)
      if (!blocks->base_coef[i] || ((uintptr_t)blocks->base_coef[i] & 15)) return err_status(UHDR_CODEC_INVALID_PARAM, "base coefficient buffer %d is null or not 16-byte aligned", i);
    for (int i = 0; i < nch; i++)
      if (!blocks->map_coef[i] || ((uintptr_t)blocks->map_coef[i] & 15)) return err_status(UHDR_CODEC_INVALID_PARAM, "map coefficient buffer %d is null or not 16-byte aligned", i);
    if (gm) {
      fill_gainmap_desc(hdr, p, gm);
      if (!gm->planes[0]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for the gainmap image's plane");
      if (gm->stride[0] < gm->w) return err_status(UHDR_CODEC_INVALID_PARAM, "gainmap stride (%u) cannot be less than its width (%u)", gm->stride[0], gm->w);
      if (((uintptr_t)gm->planes[0] | ((size_t)gm->stride[0] * nch)) & 7) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain stores the map in 8-byte pieces: aligned rows");
      map_out = (uint8_t*)gm->planes[0];
      map_stride = gm->stride[0];
    }
    if (base_encoding != UHDR_CG_UNSPECIFIED) {
      const int r = host::yuv_encoding_matrix(sdr->cg, base_encoding, &conv);
      if (r == -1) return err_status(UHDR_CODEC_INVALID_PARAM, "Unrecognized src color gamut %d", sdr->cg);
      if (r == -2) return err_status(UHDR_CODEC_INVALID_PARAM, "Unrecognized dest color gamut %d", base_encoding);
      convert = r == 0;
    }
    UHDR_TRY(ensure(c->scratch[7], (size_t)p.map_w * p.map_h * nch * sizeof(float)));
    return ok_status();
  };
  // the exchange buffers first: without them a rank cannot even contribute the identity
  UHDR_TRY(upload_math(c));
  UHDR_TRY(ensure(c->minmax, (6 + 2048 * 6) * sizeof(float)));
  UHDR_TRY(ensure(c->affine, kAffineDevBytes));
  UHDR_TRY(ensure(c->exchange, 256));
  if (!c->h_mm) HIP_TRY(hipHostMalloc((void**)&c->h_mm, 9 * sizeof(float), hipHostMallocDefault));
  uhdr_error_info_t local = prepare();
  if (local.error_code != UHDR_CODEC_OK && !striped) return local;
  bool run = local.error_code == UHDR_CODEC_OK && !empty;
  float* merged = (float*)c->exchange.p;
  float* final_mm = (float*)((char*)c->exchange.p + 192);
  uhdr_error_info_t xchg = ok_status();
  auto note_hip = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && local.error_code == UHDR_CODEC_OK) local = err_status(UHDR_CODEC_ERROR, "%s: %s", what, hipGetErrorString(e));
  };
  // profiling: the per-stage families below, and ONE event pair around the whole chain ("encode_api1_chain": first launch's start to
  // last launch's end, the gaps between the four launches included) -- destroyed, i.e. recorded, before the metadata copy
  std::unique_ptr<ProfScope> chain(new ProfScope(c, "encode_api1_chain"));
  {
    ProfScope ps(c, "generate_gainmap");
    if (run) {
      p.gain_log2 = (float*)c->scratch[7].p;
      p.minmax = (float*)c->minmax.p;
      const hipError_t e = launch_generate_gainmap(p, true, c->stream);
      note_hip(e, "generate_gainmap pass 1");
      if (e != hipSuccess) run = false;
    }
    MinmaxTableParams t;
    memset(&t, 0, sizeof t);
    t.partials = (const float*)c->minmax.p + 6;
    t.n_partials = run ? gen_partials_count(p) : 0;
    t.empty = run ? 0 : 1;
    t.mm6 = (float*)c->minmax.p;
    fill_finalize(&t, cfg);
    if (local.error_code != UHDR_CODEC_OK || empty) t.nch = (cfg && cfg->use_multi_channel_gainmap) ? 3 : 1;
    t.out_mm = final_mm;
    t.dev = (AffineDev*)c->affine.p;
    t.math_tab = c->d_math;
    if (!striped) {  // one launch for everything between the passes
      t.do_reduce = t.do_finalize = t.do_table = 1;
      note_hip(launch_minmax_table(t, c->stream), "minmax / tables");
    } else {
      MinmaxTableParams r = t;
      r.do_reduce = 1;
      r.merged6 = merged;
      note_hip(launch_minmax_table(r, c->stream), "minmax reduce");
      {
        ProfScope px(c, "stripe_exchange");
        xchg = comm_all_reduce_min(c, merged, 6);
      }
      t.do_finalize = t.do_table = 1;
      t.merged_in = merged;
      note_hip(launch_minmax_table(t, c->stream), "minmax finalize / tables");
    }
  }
  if (run && xchg.error_code == UHDR_CODEC_OK) {
    ProfScope ps(c, "fdct_quant");
    note_hip(launch_map_blocks(p.gain_log2, (const AffineDev*)c->affine.p, c->d_math, nch, (int)(p.map_w / 8), (int)(p.map_h / 8), qt_map[0], qt_map[1],
                               blocks->map_coef, map_out, map_stride, c->stream), "map blocks");
    note_hip(launch_base_blocks(view_of(sdr), convert ? &conv : nullptr, qt_base[0], qt_base[1], blocks->base_coef, c->stream), "base blocks");
  }
  chain.reset();
  note_hip(hipMemcpyAsync(c->h_mm, final_mm, 9 * sizeof(float), hipMemcpyDeviceToHost, c->stream), "metadata copy");
  note_hip(hipStreamSynchronize(c->stream), "synchronize");  // the only host synchronisation: the metadata needs the (merged) range
  if (xchg.error_code != UHDR_CODEC_OK) return xchg;
  if (local.error_code != UHDR_CODEC_OK) return local;
  float mm[6];
  memcpy(mm, c->h_mm, sizeof mm);
  if (run) note_table_stats(c, cfg);
  return uhdr_hip_generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
}

// JpegR::encodeJPEGR API-1 (jpegr.cpp:253-316) from its two raw intents to its two entropy-coded scans in ONE entry point: the
// intents go up once (fast_h2d), the fused chain leaves coefficient blocks in HBM, the marker-less Huffman coder turns them into
// the reference's bytes, and only those come down.  What the facade's seam at encodeJPEGR calls.
static uhdr_error_info_t encode_api1_scans_impl(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                                uhdr_color_gamut_t base_encoding, const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                                uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc, uint8_t* base_scan, size_t base_capacity,
                                                size_t* base_bytes, uint8_t* map_scan, size_t map_capacity, size_t* map_bytes);
uhdr_error_info_t uhdr_hip_encode_api1_scans(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                             uhdr_color_gamut_t base_encoding, const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                             uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc, uint8_t* base_scan, size_t base_capacity,
                                             size_t* base_bytes, uint8_t* map_scan, size_t map_capacity, size_t* map_bytes) {
  const auto t0 = std::chrono::steady_clock::now();
  const uhdr_error_info_t r = encode_api1_scans_impl(c, sdr, hdr, cfg, base_encoding, qt_base, qt_map, md, gainmap_desc, base_scan, base_capacity, base_bytes,
                                                     map_scan, map_capacity, map_bytes);
  if (c) c->stats.last_encode_api1_scans_ns = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return r;
}
static uhdr_error_info_t encode_api1_scans_impl(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                                uhdr_color_gamut_t base_encoding, const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                                uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc, uint8_t* base_scan, size_t base_capacity,
                                                size_t* base_bytes, uint8_t* map_scan, size_t map_capacity, size_t* map_bytes) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!sdr || !hdr || !cfg || !qt_base || !qt_map || !md || !base_scan || !map_scan || !base_bytes || !map_bytes)
    return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  // what the fused chain would decline is declined before 37 MB go up for nothing (the same conditions, uhdr_hip_encode_api1_fused_dev)
  if (sdr->fmt != UHDR_IMG_FMT_12bppYCbCr420 || sdr->w % 16 || sdr->h % 16 || sdr->w == 0 || sdr->h == 0)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain takes a UHDR_IMG_FMT_12bppYCbCr420 base image whose dimensions are multiples of 16 "
                      "(received format %d, %ux%u); use the operators", sdr->fmt, sdr->w, sdr->h);
  if (cfg->preset == UHDR_USAGE_REALTIME) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain is the two-pass (best quality) encode");
  if (cfg->gamma != 1.0f) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain needs gain-map gamma 1 (received %f)", cfg->gamma);
  const int scale = cfg->map_dimension_scale_factor;
  if (scale < 1 || sdr->w / (unsigned)scale == 0 || sdr->h / (unsigned)scale == 0 || (sdr->w / (unsigned)scale) % 8 || (sdr->h / (unsigned)scale) % 8)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-1 chain needs map dimensions that are multiples of 8 (scale factor %d on %ux%u)", scale, sdr->w, sdr->h);
  if (hdr->w != sdr->w || hdr->h != sdr->h)
    return err_status(UHDR_CODEC_INVALID_PARAM, "sdr intent resolution %ux%u and hdr intent resolution %ux%u do not match", sdr->w, sdr->h, hdr->w, hdr->h);
  if (c->comm != nullptr || c->comm_custom) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "a context with a communicator encodes stripes (uhdr_hip_encode_api1_fused_dev)");
  UHDR_TRY(validate_image(sdr, "sdr intent"));
  UHDR_TRY(validate_image(hdr, "hdr intent"));
  HIP_TRY(hipSetDevice(c->device));
  const unsigned w = sdr->w, h = sdr->h, mw = w / (unsigned)scale, mh = h / (unsigned)scale;
  const int nch = cfg->use_multi_channel_gainmap ? 3 : 1;
  uhdr_raw_image_t ds, dh;
  if (c->resident_on) UHDR_TRY(resident_write_back_all(c));
  UHDR_TRY(stage_in(c, 0, sdr, &ds, true));
  UHDR_TRY(stage_in(c, 1, hdr, &dh, true));
  // coefficient arrays: base Y, Cb, Cr, then the map's 1 or 3 components; 256-byte aligned
  const size_t nb[3] = {(size_t)(w / 8) * (h / 8), (size_t)(w / 16) * (h / 16), (size_t)(w / 16) * (h / 16)};
  const size_t nm = (size_t)(mw / 8) * (mh / 8);
  size_t off = 0, o_base[3], o_map[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++) { o_base[i] = off; off += (nb[i] * 128 + 255) & ~(size_t)255; }
  for (int i = 0; i < nch; i++) { o_map[i] = off; off += (nm * 128 + 255) & ~(size_t)255; }
  UHDR_TRY(ensure(c->enc[0], off));
  uhdr_hip_api1_blocks_t blocks;
  memset(&blocks, 0, sizeof blocks);
  for (int i = 0; i < 3; i++) blocks.base_coef[i] = (int16_t*)((uint8_t*)c->enc[0].p + o_base[i]);
  for (int i = 0; i < nch; i++) blocks.map_coef[i] = (int16_t*)((uint8_t*)c->enc[0].p + o_map[i]);
  uhdr_raw_image_t gm;
  memset(&gm, 0, sizeof gm);
  UHDR_TRY(uhdr_hip_encode_api1_fused_dev(c, &ds, &dh, cfg, base_encoding, qt_base, qt_map, &blocks, md, nullptr));
  if (gainmap_desc) {  // what generateGainMap's freshly allocated image would say (jpegr.cpp:714-716); planes untouched
    gainmap_desc->fmt = nch == 3 ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_8bppYCbCr400;
    gainmap_desc->cg = hdr->cg; gainmap_desc->ct = hdr->ct; gainmap_desc->range = hdr->range;
    gainmap_desc->w = mw; gainmap_desc->h = mh;
  }
  // the two scans: device buffers as large as the caller's, then one copy each
  uhdr_hip_jpeg_scan_t sb, sm;
  memset(&sb, 0, sizeof sb);
  memset(&sm, 0, sizeof sm);
  sb.num_components = 3;
  sb.w = w; sb.h = h;
  for (int i = 0; i < 3; i++) {
    sb.coef[i] = blocks.base_coef[i];
    sb.blocks_w[i] = (int)(i ? w / 16 : w / 8);
    sb.blocks_h[i] = (int)(i ? h / 16 : h / 8);
    sb.h_samp[i] = sb.v_samp[i] = i ? 1 : 2;
  }
  sm.num_components = nch;
  sm.w = mw; sm.h = mh;
  for (int i = 0; i < nch; i++) {
    sm.coef[i] = blocks.map_coef[i];
    sm.blocks_w[i] = (int)(mw / 8);
    sm.blocks_h[i] = (int)(mh / 8);
    sm.h_samp[i] = sm.v_samp[i] = 1;
  }
  if (base_capacity > 0xFFFFFFF0u) base_capacity = 0xFFFFFFF0u;
  if (map_capacity > 0xFFFFFFF0u) map_capacity = 0xFFFFFFF0u;
  UHDR_TRY(ensure(c->enc[1], base_capacity + 64));
  UHDR_TRY(ensure(c->enc[2], map_capacity + 64));
  size_t nbs = 0, nms = 0;
  const uhdr_error_info_t eb = uhdr_hip_huffman_encode_dev(c, &sb, (uint8_t*)c->enc[1].p, base_capacity, &nbs);
  *base_bytes = nbs;
  if (eb.error_code != UHDR_CODEC_OK) { *map_bytes = 0; return eb; }
  HIP_TRY(hipMemcpyAsync(base_scan, c->enc[1].p, nbs, hipMemcpyDeviceToHost, c->stream));  // overlaps the map's entropy coding
  const uhdr_error_info_t em = uhdr_hip_huffman_encode_dev(c, &sm, (uint8_t*)c->enc[2].p, map_capacity, &nms);
  *map_bytes = nms;
  if (em.error_code != UHDR_CODEC_OK) { (void)hipStreamSynchronize(c->stream); return em; }
  HIP_TRY(hipMemcpyAsync(map_scan, c->enc[2].p, nms, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

// -------------------------------------------------------------------------------------------------
// copy_raw_image (gainmapmath.cpp:1492-1613), device to device
// -------------------------------------------------------------------------------------------------
uhdr_error_info_t uhdr_hip_copy_raw_image_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* src, uhdr_raw_image_t* dst) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!src || !dst) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  if (dst->w != src->w || dst->h != src->h)
    return err_status(UHDR_CODEC_MEM_ERROR, "destination image dimensions %dx%d and source image dimensions %dx%d are not identical for copy_raw_image",
                      dst->w, dst->h, src->w, src->h);
  UHDR_TRY(validate_image(src, "source"));
  UHDR_TRY(validate_image(dst, "destination"));
  HIP_TRY(hipSetDevice(c->device));
  dst->cg = src->cg; dst->ct = src->ct; dst->range = src->range;
  const size_t w = src->w, h = src->h;
  auto copy2d = [&](int pl, size_t bps, size_t width, size_t rows) -> hipError_t {
    if (!width || !rows) return hipSuccess;
    return hipMemcpy2DAsync(dst->planes[pl], (size_t)dst->stride[pl] * bps, src->planes[pl], (size_t)src->stride[pl] * bps,
                            width * bps, rows, hipMemcpyDeviceToDevice, c->stream);
  };
  if (dst->fmt == src->fmt) {
    switch (src->fmt) {
      case UHDR_IMG_FMT_24bppYCbCrP010:  // h / 2 chroma rows of w samples, as the reference copies them
        HIP_TRY(copy2d(0, 2, w, h));
        HIP_TRY(copy2d(1, 2, w, h / 2));
        return ok_status();
      case UHDR_IMG_FMT_12bppYCbCr420:
        HIP_TRY(copy2d(0, 1, w, h));
        HIP_TRY(copy2d(1, 1, w / 2, h / 2));
        HIP_TRY(copy2d(2, 1, w / 2, h / 2));
        return ok_status();
      case UHDR_IMG_FMT_8bppYCbCr400: HIP_TRY(copy2d(0, 1, w, h)); return ok_status();
      case UHDR_IMG_FMT_32bppRGBA8888:
      case UHDR_IMG_FMT_32bppRGBA1010102: HIP_TRY(copy2d(0, 4, w, h)); return ok_status();
      case UHDR_IMG_FMT_64bppRGBAHalfFloat: HIP_TRY(copy2d(0, 8, w, h)); return ok_status();
      case UHDR_IMG_FMT_24bppRGB888: HIP_TRY(copy2d(0, 3, w, h)); return ok_status();
      default: break;
    }
  } else if (src->fmt == UHDR_IMG_FMT_24bppRGB888 && dst->fmt == UHDR_IMG_FMT_32bppRGBA8888) {
    HIP_TRY(launch_repack(0, src->planes[0], (size_t)src->stride[0] * 3, dst->planes[0], (size_t)dst->stride[0] * 4, src->w, src->h, c->stream));
    return ok_status();
  } else if (src->fmt == UHDR_IMG_FMT_32bppRGBA8888 && dst->fmt == UHDR_IMG_FMT_8bppYCbCr400) {
    HIP_TRY(launch_repack(1, src->planes[0], (size_t)src->stride[0] * 4, dst->planes[0], (size_t)dst->stride[0], src->w, src->h, c->stream));
    return ok_status();
  }
  return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "unsupported source / destinations color formats in copy_raw_image, src fmt %d, dst fmt %d",
                    src->fmt, dst->fmt);
}

// -------------------------------------------------------------------------------------------------
// JPEG decode stage: dequant + IDCT, libjpeg colour conversions
// -------------------------------------------------------------------------------------------------
uhdr_error_info_t uhdr_hip_idct_dequant_dev(uhdr_hip_ctx_t* c, const int16_t* coef, int bw, int bh, const uint16_t qt[64],
                                            uint8_t* plane, size_t stride) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!plane || !qt || !coef || bw <= 0 || bh <= 0) return err_status(UHDR_CODEC_INVALID_PARAM, "received bad argument for idct_dequant");
  if (((uintptr_t)coef & 15) != 0) return err_status(UHDR_CODEC_INVALID_PARAM, "coefficient buffer must be 16-byte aligned");
  if (stride < (size_t)bw * 8) return err_status(UHDR_CODEC_INVALID_PARAM, "plane stride (%zu) cannot be less than blocks_w * 8 (%d)", stride, bw * 8);
  for (int i = 0; i < 64; i++)
    if (qt[i] == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d is zero", i);
  HIP_TRY(hipSetDevice(c->device));
  ProfScope ps(c, "idct_dequant");
  HIP_TRY(launch_idct_dequant(coef, bw, bh, qt, plane, stride, c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_idct_dequant(uhdr_hip_ctx_t* c, const int16_t* coef, int bw, int bh, const uint16_t qt[64],
                                        uint8_t* plane, size_t stride) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!plane || !qt || !coef || bw <= 0 || bh <= 0) return err_status(UHDR_CODEC_INVALID_PARAM, "received bad argument for idct_dequant");
  if (stride < (size_t)bw * 8) return err_status(UHDR_CODEC_INVALID_PARAM, "plane stride (%zu) cannot be less than blocks_w * 8 (%d)", stride, bw * 8);
  HIP_TRY(hipSetDevice(c->device));
  const size_t in_bytes = (size_t)bw * bh * 64 * sizeof(int16_t);
  const size_t dpitch = ((size_t)bw * 8 + 63) & ~(size_t)63;
  UHDR_TRY(ensure(c->scratch[0], in_bytes));
  UHDR_TRY(ensure(c->scratch[1], dpitch * (size_t)bh * 8));
  HIP_TRY(hipMemcpyAsync(c->scratch[0].p, coef, in_bytes, hipMemcpyHostToDevice, c->stream));
  UHDR_TRY(uhdr_hip_idct_dequant_dev(c, (const int16_t*)c->scratch[0].p, bw, bh, qt, (uint8_t*)c->scratch[1].p, dpitch));
  resident_drop(c, plane);  //  a device copy kept for this host plane is stale from here on
  HIP_TRY(hipMemcpy2DAsync(plane, stride, c->scratch[1].p, dpitch, (size_t)bw * 8, (size_t)bh * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

static uhdr_error_info_t check_jpeg_color(const uhdr_raw_image_t* rgb, const uhdr_raw_image_t* ycc, bool ycc_is_dst) {
  if (!rgb || !ycc) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  if (rgb->fmt != UHDR_IMG_FMT_24bppRGB888 && rgb->fmt != UHDR_IMG_FMT_32bppRGBA8888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "libjpeg colour conversion expects UHDR_IMG_FMT_24bppRGB888 or UHDR_IMG_FMT_32bppRGBA8888 on the RGB side. Received %d", rgb->fmt);
  const uhdr_raw_image_t* src = ycc_is_dst ? rgb : ycc;
  const uhdr_raw_image_t* dst = ycc_is_dst ? ycc : rgb;
  if (!ycc_is_dst && ycc->fmt != UHDR_IMG_FMT_24bppYCbCr444)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "libjpeg colour conversion expects UHDR_IMG_FMT_24bppYCbCr444 on the YCbCr side. Received %d", ycc->fmt);
  if (src->w == 0 || src->h == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "image dimensions cannot be zero, received %ux%u", src->w, src->h);
  const int np_src = ycc_is_dst ? 1 : 3, np_dst = ycc_is_dst ? 3 : 1;
  for (int i = 0; i < np_src; i++) {
    if (!src->planes[i]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for source plane %d", i);
    if (src->stride[i] < src->w) return err_status(UHDR_CODEC_INVALID_PARAM, "source stride (%u) cannot be less than width (%u)", src->stride[i], src->w);
  }
  for (int i = 0; i < np_dst; i++) {
    if (!dst->planes[i]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for destination plane %d", i);
    if (dst->stride[i] < src->w) return err_status(UHDR_CODEC_INVALID_PARAM, "destination stride (%u) cannot be less than width (%u)", dst->stride[i], src->w);
  }
  return ok_status();
}

uhdr_error_info_t uhdr_hip_jpeg_rgb_to_ycc_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* rgb, uhdr_raw_image_t* ycc) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  UHDR_TRY(check_jpeg_color(rgb, ycc, true));
  HIP_TRY(hipSetDevice(c->device));
  ycc->fmt = UHDR_IMG_FMT_24bppYCbCr444;
  ycc->cg = rgb->cg; ycc->ct = rgb->ct; ycc->range = UHDR_CR_FULL_RANGE;
  ycc->w = rgb->w; ycc->h = rgb->h;
  ProfScope ps(c, "jpeg_color");
  HIP_TRY(launch_jpeg_rgb_to_ycc(view_of(rgb), view_mut_of(ycc), c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_jpeg_rgb_to_ycc(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* rgb, uhdr_raw_image_t* ycc) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  UHDR_TRY(check_jpeg_color(rgb, ycc, true));
  HIP_TRY(hipSetDevice(c->device));
  uhdr_raw_image_t tmp = *ycc;
  tmp.fmt = UHDR_IMG_FMT_24bppYCbCr444;
  tmp.w = rgb->w; tmp.h = rgb->h;
  uhdr_raw_image_t ds, dd;
  UHDR_TRY(stage_in(c, 0, rgb, &ds, true));
  UHDR_TRY(stage_in(c, 1, &tmp, &dd, false));
  UHDR_TRY(uhdr_hip_jpeg_rgb_to_ycc_dev(c, &ds, &dd));
  ycc->fmt = dd.fmt; ycc->cg = dd.cg; ycc->ct = dd.ct; ycc->range = dd.range; ycc->w = dd.w; ycc->h = dd.h;
  return stage_out(c, &dd, ycc);
}

uhdr_error_info_t uhdr_hip_jpeg_ycc_to_rgb_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* ycc, int variant, uhdr_raw_image_t* rgb) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  UHDR_TRY(check_jpeg_color(rgb, ycc, false));
  if (variant != 0 && variant != 1) return err_status(UHDR_CODEC_INVALID_PARAM, "unknown libjpeg variant %d", variant);
  HIP_TRY(hipSetDevice(c->device));
  rgb->cg = ycc->cg; rgb->ct = ycc->ct; rgb->range = UHDR_CR_FULL_RANGE;
  rgb->w = ycc->w; rgb->h = ycc->h;
  ProfScope ps(c, "jpeg_color");
  HIP_TRY(launch_jpeg_ycc_to_rgb(view_of(ycc), view_mut_of(rgb), variant, c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_jpeg_ycc_to_rgb(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* ycc, int variant, uhdr_raw_image_t* rgb) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  UHDR_TRY(check_jpeg_color(rgb, ycc, false));
  HIP_TRY(hipSetDevice(c->device));
  uhdr_raw_image_t tmp = *rgb;
  tmp.w = ycc->w; tmp.h = ycc->h;
  uhdr_raw_image_t ds, dd;
  UHDR_TRY(stage_in(c, 0, ycc, &ds, true));
  UHDR_TRY(stage_in(c, 1, &tmp, &dd, false));
  UHDR_TRY(uhdr_hip_jpeg_ycc_to_rgb_dev(c, &ds, variant, &dd));
  rgb->cg = dd.cg; rgb->ct = dd.ct; rgb->range = dd.range; rgb->w = dd.w; rgb->h = dd.h;
  return stage_out(c, &dd, rgb);
}

// 3-channel gain map: dequant + IDCT of the three components + ycc_rgb_convert in one pass
uhdr_error_info_t uhdr_hip_idct_dequant_rgb_dev(uhdr_hip_ctx_t* c, const int16_t* coef_y, const int16_t* coef_cb, const int16_t* coef_cr,
                                                int bw, int bh, const uint16_t qt_luma[64], const uint16_t qt_chroma[64], int variant,
                                                uhdr_raw_image_t* rgb) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!coef_y || !coef_cb || !coef_cr || !qt_luma || !qt_chroma || !rgb || !rgb->planes[0] || bw <= 0 || bh <= 0)
    return err_status(UHDR_CODEC_INVALID_PARAM, "received bad argument for idct_dequant_rgb");
  if (rgb->fmt != UHDR_IMG_FMT_24bppRGB888 && rgb->fmt != UHDR_IMG_FMT_32bppRGBA8888)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "idct_dequant_rgb expects UHDR_IMG_FMT_24bppRGB888 or UHDR_IMG_FMT_32bppRGBA8888. Received %d", rgb->fmt);
  if (variant != 0 && variant != 1) return err_status(UHDR_CODEC_INVALID_PARAM, "unknown libjpeg variant %d", variant);
  if (rgb->w == 0 || rgb->h == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "image dimensions cannot be zero, received %ux%u", rgb->w, rgb->h);
  if ((rgb->w + 7) / 8 != (unsigned)bw || (rgb->h + 7) / 8 != (unsigned)bh)
    return err_status(UHDR_CODEC_INVALID_PARAM, "image %ux%u does not match a %dx%d block grid", rgb->w, rgb->h, bw, bh);
  if (rgb->stride[0] < rgb->w) return err_status(UHDR_CODEC_INVALID_PARAM, "stride (%u) cannot be less than width (%u)", rgb->stride[0], rgb->w);
  if (((uintptr_t)coef_y | (uintptr_t)coef_cb | (uintptr_t)coef_cr) & 15)
    return err_status(UHDR_CODEC_INVALID_PARAM, "coefficient buffers must be 16-byte aligned");
  for (int i = 0; i < 64; i++)
    if (qt_luma[i] == 0 || qt_chroma[i] == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d is zero", i);
  HIP_TRY(hipSetDevice(c->device));
  rgb->range = UHDR_CR_FULL_RANGE;
  ProfScope ps(c, "idct_dequant");
  HIP_TRY(launch_idct_dequant_rgb(coef_y, coef_cb, coef_cr, bw, bh, qt_luma, qt_chroma, variant, view_mut_of(rgb), c->stream));
  return ok_status();
}

// -------------------------------------------------------------------------------------------------
// JPEG entropy stage: baseline Huffman coding of coefficient blocks, one restart interval per wavefront
// -------------------------------------------------------------------------------------------------
static uhdr_error_info_t check_scan(const uhdr_hip_jpeg_scan_t* sc, bool need_coef, int* mcus_per_row, int* mcu_rows, int* blocks_per_mcu) {
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
    uint64_t total = 0;
    uint32_t meta[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(meta, t.meta, sizeof meta, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
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
  HIP_TRY(hipStreamSynchronize(c->stream));
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
static bool make_fast_table(const uint8_t bits[17], const uint8_t vals[256], HuffFastTable* t) {
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
                   kTabVal = (kTabFast + sizeof(HuffFastTable) * 8 + 255) & ~(size_t)255, kTabBytes = kTabVal + (size_t)4 * kHuffValWords * 4;
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
      std::vector<uint32_t> vtabs((size_t)4 * kHuffValWords);
      bool fast_ok = true;
      for (int t = 0; t < 4; t++) {
        const uint8_t* bits = key + (size_t)t * (17 + 256);
        if (!make_dec_table(bits, bits + 17, &tabs[(size_t)t])) return err_status(UHDR_CODEC_INVALID_PARAM, "Huffman table %d is not a valid DHT table", t);
        fast_ok = make_fast_table(bits, bits + 17, &ftabs[(size_t)t]) && fast_ok;
      }
      if (fast_ok)
        for (int t = 0; t < 4; t++) {
          make_track_table(ftabs[(size_t)t], (t & 1) == 0, &ftabs[(size_t)t + 4]);
          make_value_table(ftabs[(size_t)t], (t & 1) == 0, vtabs.data() + (size_t)t * kHuffValWords);
        }
      UHDR_TRY(ensure(hc.dev, kTabBytes));
      HIP_TRY(hipStreamSynchronize(c->stream));  // nothing in flight reads the old tables
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabDec, tabs.data(), sizeof(HuffDecTable) * 4, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabFast, ftabs.data(), sizeof(HuffFastTable) * 8, hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy((uint8_t*)hc.dev.p + kTabVal, vtabs.data(), vtabs.size() * 4, hipMemcpyHostToDevice));
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
  HIP_TRY(hipMemsetAsync(a.status, 0, 16, c->stream));
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
      HIP_TRY(hipMemsetAsync(y.flags, 0, zero_bytes_sync, c->stream));  // flags, nblk, dcd and the restart map
      if (form2) HIP_TRY(hipMemsetAsync(y.coef_scan, 0, scan_bytes, c->stream));
      dbg.mark("huffman_decode_dev: fills enqueued");
      int final_buf = 0;
      uint32_t* fl = c->h_flags;  // pinned; [9]: restart markers the unstuff pass dropped, [16] [17] / [0] [7]: their sequence sums as found / as due
      for (int q = 0; q < 24; q++) fl[q] = 0;
      bool hyp_done = false, unstuffed = false, rounds_ran = false;
      auto start_over = [&]() -> uhdr_error_info_t {  // an attempt failed: everything it wrote goes back to its initial state
        HIP_TRY(hipMemsetAsync(y.flags, 0, 32, c->stream));  // not [8]: the stuffed-byte count stays
        HIP_TRY(hipMemsetAsync(y.nblk, 0, (size_t)nsub * 4 + 4, c->stream));
        if (form2) {
          HIP_TRY(hipMemsetAsync(y.coef_scan, 0, scan_bytes, c->stream));
        } else {
          HIP_TRY(hipMemsetAsync(y.dcd, 0, (size_t)total_blocks * 4, c->stream));
          for (int i = 0; i < a.ncomp; i++) HIP_TRY(hipMemsetAsync(a.coef[i], 0, zero_bytes[i], c->stream));
        }
        return ok_status();
      };
      for (size_t ti = 0; ti < attempts.size() && !hyp_done && !rounds_ran; ti++) {
        const Attempt& t = attempts[ti];
        if (ti > 0) UHDR_TRY(start_over());
        y.sub_bits = t.sub_bits;
        const uint32_t nsub_t = huff_sync_max_subsequences(data_bytes, t.sub_bits);
        if (t.levels > 0) {
          y.hyp_h = bpm;
          y.hyp_levels = t.levels;
          y.hyp_main_levels = rst_sync ? 0 : main_levels_env;
          y.hyp_state = (uint64_t*)(sb + o_hs);
          y.hyp_map = sb + o_hm;
          y.hyp_cnt = (uint16_t*)(sb + o_hc);
          y.hyp_hist = getenv("UHDR_HIP_HUFF_DEBUG") ? 1 : 0;
          // hyp_map <- 0xff (unmapped); state[0] <- 0xff: a start state the write pass skips, should the chain be lost
          HIP_TRY(hipMemsetAsync(y.state[0], 0xff, ff_bytes, c->stream));
          // hyp_cnt needs no initialisation: a slot's count is written together with its map entry, and only mapped slots are read
          size_t tiles_off = 0;
          (void)huff_hyp_chain_bytes(data_bytes, t.sub_bits, &tiles_off);
          {
            ProfScope ps(c, "huffman_decode");
            if (!unstuffed) HIP_TRY(launch_huffman_unstuff(data, (uint32_t)data_bytes, (uint32_t*)(sb + o_cnt), y.flags + 8, sb + o_clean, c->stream, rst_map, y.rst_partial));
            unstuffed = true;
            HIP_TRY(launch_huffman_decode_hyp(y, (int*)(sb + o_dcp), sb + o_ch, sb + o_ch + tiles_off, c->stream));
          }
          dbg.mark("huffman_decode_dev: hypothesis attempt enqueued");
          HIP_TRY(hipMemcpyAsync(fl, y.flags, 24 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(hipStreamSynchronize(c->stream));
          dbg.mark("huffman_decode_dev: hypothesis attempt finished");
          hyp_done = fl[2] == 0;
          if (hyp_done && ti > 0) {  // where the next scan like this one starts
            uhdr_hip_ctx::HuffHint& hint = c->huff_hint[bpm & 15];
            hint.sub_bits = t.sub_bits;
            hint.bits_per_block = (uint32_t)((uint64_t)data_bytes * 8u / ((uint64_t)a.total_mcus * (uint64_t)bpm));
          }
          if (getenv("UHDR_HIP_HUFF_DEBUG")) {
            uint32_t hist[16] = {};
            (void)hipMemcpy(hist, y.flags, sizeof hist, hipMemcpyDeviceToHost);
            fprintf(stderr, "uhdr_hip: hypothesis decode of %zu bytes, %u subsequences of %u bits x %d: merges per level %u %u %u %u %u %u+, %u paths unmerged after %d levels (%d in lockstep, %u paths handed to the straggler waves), true path %s\n",
                    data_bytes, nsub_t, t.sub_bits, bpm, hist[10], hist[11], hist[12], hist[13], hist[14], hist[15], fl[3], t.levels, y.hyp_main_levels, fl[kHuffFlagStragglers], hyp_done ? "resolved" : "LOST (next attempt)");
          }
        } else {
          {
            ProfScope ps(c, "huffman_decode");
            if (!unstuffed) HIP_TRY(launch_huffman_unstuff(data, (uint32_t)data_bytes, (uint32_t*)(sb + o_cnt), y.flags + 8, sb + o_clean, c->stream, rst_map, y.rst_partial));
            unstuffed = true;
            HIP_TRY(launch_huffman_decode_sync(y, max_rounds, (int*)(sb + o_dcp), &final_buf, c->stream));
          }
          HIP_TRY(hipMemcpyAsync(fl, y.flags, 24 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(hipStreamSynchronize(c->stream));
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
  {
    ProfScope ps(c, "huffman_decode");
    HIP_TRY(launch_huffman_decode(a, counts, starts, ends, c->stream));
  }
  if (a.nseg == 1) c->stats.entropy_decode_single_lane++;
  else c->stats.entropy_decode_intervals++;
  uint32_t st[2] = {0, 0};
  HIP_TRY(hipMemcpyAsync(st, a.status, sizeof st, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (st[1] != (uint32_t)(a.nseg - 1))
    return err_status(UHDR_CODEC_INVALID_PARAM, "found %u restart markers, a restart interval of %d MCUs over %d MCUs needs %d", st[1], a.ri,
                      a.total_mcus, a.nseg - 1);
  if (st[0] & 4u) return err_status(UHDR_CODEC_INVALID_PARAM, "restart markers out of sequence");
  if (st[0] & 2u) return err_status(UHDR_CODEC_INVALID_PARAM, "corrupt entropy-coded data (undefined Huffman code or a run past the end of a block)");
  return ok_status();
}

// JpegEncoderHelper::compressImage's sample -> entropy-coded-data part (jpegencoderhelper.cpp:131-309) on the device: FDCT +
// quantization (and rgb_ycc_convert for a packed RGB gain map) feed the restart-interval Huffman encoder without the
// coefficients leaving HBM; only the samples go up and only the compressed bytes come down.
// image_edges: the planes are the IMAGE's planes and partial edge blocks get their missing samples on the device by the
// reference helper's rules (FdctEdge, fdct_quant.hip); else the caller has padded every plane to whole blocks.
static uhdr_error_info_t jpeg_encode_impl(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_scan_t* scan, const uint16_t qtable[3][64],
                                          const uint8_t* const planes[3], const unsigned int strides[3], int rgb_channels, uint8_t* out,
                                          size_t out_capacity, size_t* out_bytes, bool image_edges) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!scan || !qtable || !planes || !strides || !out || !out_bytes) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument for jpeg_encode_scan");
  if (rgb_channels != 0 && rgb_channels != 3 && rgb_channels != 4) return err_status(UHDR_CODEC_INVALID_PARAM, "rgb_channels is 0 (planes), 3 (RGB888) or 4 (RGBA8888), received %d", rgb_channels);
  uhdr_hip_jpeg_scan_t sc = *scan;
  const int nc = sc.num_components;
  int mpr = 0, mrows = 0, bpm = 0;
  UHDR_TRY(check_scan(&sc, false, &mpr, &mrows, &bpm));
  if (sc.restart_interval < 0 || sc.restart_interval * bpm > 64)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "restart_interval must be 0 (no markers) or in 1..%d for %d blocks per MCU, received %d", 64 / bpm, bpm, sc.restart_interval);
  HIP_TRY(hipSetDevice(c->device));
  size_t coef_bytes = 0;
  for (int i = 0; i < nc; i++) {
    if (sc.blocks_w[i] <= 0 || sc.blocks_h[i] <= 0) return err_status(UHDR_CODEC_INVALID_PARAM, "component %d has an empty block grid", i);
    const size_t b = (size_t)sc.blocks_w[i] * sc.blocks_h[i] * 64 * sizeof(int16_t);
    UHDR_TRY(ensure(c->jpg[1 + i], b));
    sc.coef[i] = (const int16_t*)c->jpg[1 + i].p;
    coef_bytes += b;
  }
  if (rgb_channels == 0) {
    // valid samples per plane (JpegEncoderHelper::encode's mPlaneWidth / mPlaneHeight, jpegencoderhelper.cpp:190-195) and what is copied up
    int hmax = 1, vmax = 1;
    for (int i = 0; i < nc; i++) {
      if (nc == 3 && sc.h_samp[i] > hmax) hmax = sc.h_samp[i];
      if (nc == 3 && sc.v_samp[i] > vmax) vmax = sc.v_samp[i];
    }
    unsigned pw[3] = {0, 0, 0}, ph[3] = {0, 0, 0}, cols[3] = {0, 0, 0}, rows[3] = {0, 0, 0};
    FdctEdge edge[3];
    memset(edge, 0, sizeof edge);
    size_t pitch[3] = {0, 0, 0}, off[3] = {0, 0, 0}, total = 0;
    bool width_partial = false;
    for (int i = 0; i < nc; i++) {
      const unsigned aligned = (unsigned)sc.blocks_w[i] * 8;
      if (image_edges) {
        const int hs = nc == 1 ? 1 : sc.h_samp[i], vs = nc == 1 ? 1 : sc.v_samp[i];
        pw[i] = (sc.w * hs + hmax - 1) / hmax;
        ph[i] = (sc.h * vs + vmax - 1) / vmax;
        if ((unsigned)sc.blocks_w[i] != (pw[i] + 7) / 8 || (unsigned)sc.blocks_h[i] != (ph[i] + 7) / 8)
          return err_status(UHDR_CODEC_INVALID_PARAM, "component %d: %dx%d blocks are not the real blocks of a %ux%u plane", i, sc.blocks_w[i], sc.blocks_h[i], pw[i], ph[i]);
        if (!planes[i] || strides[i] < pw[i]) return err_status(UHDR_CODEC_INVALID_PARAM, "plane %d: nullptr or stride below the plane width", i);
        edge[i].on = (pw[i] % 8 || ph[i] % 8) ? 1 : 0;
        edge[i].w = (int)pw[i];
        edge[i].h = (int)ph[i];
        edge[i].col_mode = strides[i] >= aligned ? 0 : 1;  // jpegencoderhelper.cpp:257: strides[i] < alignedPlaneWidth[i] copies rows into a scratch MCU row
        edge[i].fill = i == 0 ? 0 : 128;
        edge[i].mcu_rows = vs * 8;
        cols[i] = edge[i].col_mode == 0 ? aligned : pw[i];
        rows[i] = ph[i];
        if (pw[i] % 8) width_partial = true;
      } else {
        if (!planes[i] || strides[i] < aligned) return err_status(UHDR_CODEC_INVALID_PARAM, "plane %d: nullptr or stride below blocks_w * 8", i);
        pw[i] = cols[i] = aligned;
        ph[i] = rows[i] = (unsigned)sc.blocks_h[i] * 8;
      }
      pitch[i] = ((size_t)aligned + 63) & ~(size_t)63;
      off[i] = total;
      total += pitch[i] * (size_t)sc.blocks_h[i] * 8;
    }
    // planes the library itself produced a moment ago (resident_keep) are read where they are -- unless a plane's width
    // is not whole blocks: the bytes BEHIND the width are then part of the input (the caller's stride bytes) and only the
    // host buffer has them
    const uhdr_hip_ctx::Resident* held = nullptr;
    if (c->resident_on && !width_partial)
      for (const auto& r : c->resident) {
        bool ok = r.valid && r.fmt != UHDR_IMG_FMT_24bppRGB888 && r.fmt != UHDR_IMG_FMT_32bppRGBA8888;
        for (int i = 0; ok && i < nc; i++)
          ok = r.host[i] == planes[i] && r.host_stride[i] == strides[i] && r.pcols[i] >= cols[i] && r.prows[i] >= rows[i];
        if (ok) { held = &r; break; }
      }
    auto fdct = [&](int i, const uint8_t* d, size_t dpitch) -> uhdr_error_info_t {
      for (int k = 0; k < 64; k++)
        if (qtable[i][k] == 0 || qtable[i][k] > 255) return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d out of baseline range", k);
      ProfScope ps(c, "fdct_quant");
      HIP_TRY(launch_fdct_quant(d, dpitch, sc.blocks_w[i], sc.blocks_h[i], qtable[i], (int16_t*)c->jpg[1 + i].p, c->stream, image_edges ? &edge[i] : nullptr));
      return ok_status();
    };
    if (held) {
      c->stats.resident_hits++;
      for (int i = 0; i < nc; i++) UHDR_TRY(fdct(i, (const uint8_t*)held->buf.p + held->off[i], held->dev_stride[i]));
    } else {
      if (c->resident_on) UHDR_TRY(resident_write_back_all(c));  // the host planes are read below
      UHDR_TRY(ensure(c->jpg[4], total));
      for (int i = 0; i < nc; i++) {
        uint8_t* d = (uint8_t*)c->jpg[4].p + off[i];
        HIP_TRY(hipMemcpy2DAsync(d, pitch[i], planes[i], strides[i], cols[i], rows[i], hipMemcpyHostToDevice, c->stream));
        UHDR_TRY(fdct(i, d, pitch[i]));
      }
    }
  } else {
    if (nc != 3 || bpm != 3) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "packed RGB input is a 3-component 4:4:4 scan");
    if (!image_edges && (sc.w % 8 || sc.h % 8))
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "packed RGB input needs dimensions that are multiples of 8 here (uhdr_hip_jpeg_encode_image replicates the edges as libjpeg does)");
    if ((unsigned)sc.blocks_w[0] != (sc.w + 7) / 8 || (unsigned)sc.blocks_h[0] != (sc.h + 7) / 8)
      return err_status(UHDR_CODEC_INVALID_PARAM, "a %dx%d block grid does not match a %ux%u RGB image", sc.blocks_w[0], sc.blocks_h[0], sc.w, sc.h);
    if (memcmp(qtable[1], qtable[2], 64 * sizeof(uint16_t)))
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "packed RGB input with different Cb and Cr quantization tables is outside the HIP path");
    if (!planes[0] || strides[0] < sc.w) return err_status(UHDR_CODEC_INVALID_PARAM, "RGB image: nullptr or stride below the width");
    for (int i = 0; i < 64; i++)
      if (qtable[0][i] == 0 || qtable[0][i] > 255 || qtable[1][i] == 0 || qtable[1][i] > 255)
        return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d out of baseline range", i);
    const uint8_t* dsrc = nullptr;
    size_t dpitch = 0;
    const uhdr_img_fmt_t fmt = rgb_channels == 3 ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_32bppRGBA8888;
    const uhdr_hip_ctx::Resident* held = nullptr;
    if (c->resident_on)
      for (const auto& r : c->resident)
        if (r.valid && r.fmt == fmt && r.host[0] == planes[0] && r.host_stride[0] == strides[0] && r.pcols[0] >= sc.w && r.prows[0] >= sc.h &&
            ((size_t)r.dev_stride[0] * rgb_channels) % (rgb_channels == 4 ? 16 : 8) == 0) { held = &r; break; }  // (the fused kernel's row alignment)
    if (held) {  // the gain map generateGainMap has just written (resident_keep): read where it is
      c->stats.resident_hits++;
      dsrc = (const uint8_t*)held->buf.p + held->off[0];
      dpitch = (size_t)held->dev_stride[0] * rgb_channels;
    } else {
      if (c->resident_on) UHDR_TRY(resident_write_back_all(c));  // the host planes are read below
      const size_t pitch_px = ((size_t)sc.w + 15) & ~(size_t)15;
      UHDR_TRY(ensure(c->jpg[4], pitch_px * (size_t)rgb_channels * sc.h));
      HIP_TRY(hipMemcpy2DAsync(c->jpg[4].p, pitch_px * rgb_channels, planes[0], (size_t)strides[0] * rgb_channels, (size_t)sc.w * rgb_channels, sc.h,
                               hipMemcpyHostToDevice, c->stream));
      dsrc = (const uint8_t*)c->jpg[4].p;
      dpitch = pitch_px * rgb_channels;
    }
    ProfScope ps(c, "fdct_quant");
    HIP_TRY(launch_fdct_quant_rgb(dsrc, dpitch, rgb_channels, sc.blocks_w[0], sc.blocks_h[0], qtable[0], qtable[1], (int16_t*)c->jpg[1].p,
                                  (int16_t*)c->jpg[2].p, (int16_t*)c->jpg[3].p, c->stream, (int)sc.w, (int)sc.h));
  }
  size_t cap = coef_bytes / 4 + (1u << 20), n = 0;
  UHDR_TRY(ensure(c->jpg[0], cap));
  uhdr_error_info_t hs = uhdr_hip_huffman_encode_dev(c, &sc, (uint8_t*)c->jpg[0].p, c->jpg[0].cap, &n);
  if (hs.error_code == UHDR_CODEC_MEM_ERROR && n > c->jpg[0].cap) {  // busier data than the guess: the call reported the size it needs
    UHDR_TRY(ensure(c->jpg[0], n));
    hs = uhdr_hip_huffman_encode_dev(c, &sc, (uint8_t*)c->jpg[0].p, c->jpg[0].cap, &n);
  }
  if (hs.error_code != UHDR_CODEC_OK) return hs;
  *out_bytes = n;
  if (n > out_capacity) return err_status(UHDR_CODEC_MEM_ERROR, "output buffer of %zu bytes is too small for %zu bytes of entropy-coded data", out_capacity, n);
  HIP_TRY(hipMemcpyAsync(out, c->jpg[0].p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

uhdr_error_info_t uhdr_hip_jpeg_encode_scan(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_scan_t* scan, const uint16_t qtable[3][64],
                                            const uint8_t* const planes[3], const unsigned int strides[3], int rgb_channels, uint8_t* out,
                                            size_t out_capacity, size_t* out_bytes) {
  return jpeg_encode_impl(c, scan, qtable, planes, strides, rgb_channels, out, out_capacity, out_bytes, false);
}
uhdr_error_info_t uhdr_hip_jpeg_encode_image(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_scan_t* scan, const uint16_t qtable[3][64],
                                             const uint8_t* const planes[3], const unsigned int strides[3], int rgb_channels, uint8_t* out,
                                             size_t out_capacity, size_t* out_bytes) {
  return jpeg_encode_impl(c, scan, qtable, planes, strides, rgb_channels, out, out_capacity, out_bytes, true);
}

// JpegDecoderHelper::decompressImage (jpegdecoderhelper.cpp:169-535) for a baseline file whose headers are parsed: entropy
// decode, dequantization, JDCT_ISLOW IDCT and (for RGB / RGBA output of a 4:4:4 file) ycc_rgb_convert on the device; only
// the compressed bytes go up and only the decoded samples come down.
static uhdr_error_info_t jpeg_decode_scan_impl(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_header_t* hdr, const uint8_t* scan_data, size_t scan_bytes,
                                               int out_channels, int variant, uint8_t* const planes[3], const unsigned int hstride[3],
                                               const unsigned int vstride[3]);
uhdr_error_info_t uhdr_hip_jpeg_decode_scan(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_header_t* hdr, const uint8_t* scan_data, size_t scan_bytes,
                                            int out_channels, int variant, uint8_t* const planes[3], const unsigned int hstride[3],
                                            const unsigned int vstride[3]) {
  const auto t0 = std::chrono::steady_clock::now();
  const uhdr_error_info_t r = jpeg_decode_scan_impl(c, hdr, scan_data, scan_bytes, out_channels, variant, planes, hstride, vstride);
  if (c) c->stats.last_jpeg_decode_scan_ns = (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
  return r;
}
static uhdr_error_info_t jpeg_decode_scan_impl(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_header_t* hdr, const uint8_t* scan_data, size_t scan_bytes,
                                               int out_channels, int variant, uint8_t* const planes[3], const unsigned int hstride[3],
                                               const unsigned int vstride[3]) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!hdr || !scan_data || !planes || !hstride || !vstride) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument for jpeg_decode_scan");
  const DbgClock dbg;
  if (out_channels != 0 && out_channels != 3 && out_channels != 4) return err_status(UHDR_CODEC_INVALID_PARAM, "out_channels is 0 (planes), 3 (RGB888) or 4 (RGBA8888), received %d", out_channels);
  uhdr_hip_jpeg_scan_t sc = hdr->scan;
  const int nc = sc.num_components;
  int mpr = 0, mrows = 0, bpm = 0;
  UHDR_TRY(check_scan(&sc, false, &mpr, &mrows, &bpm));
  if (out_channels != 0) {
    if (nc != 3 || sc.h_samp[0] != 1 || sc.v_samp[0] != 1 || sc.h_samp[1] != 1 || sc.v_samp[1] != 1 || sc.h_samp[2] != 1 || sc.v_samp[2] != 1)
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "RGB output needs a 3-component 4:4:4 file (libjpeg's upsampling is outside the HIP path)");
    if (memcmp(hdr->qtable[1], hdr->qtable[2], sizeof hdr->qtable[1]))
      return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "RGB output with different Cb and Cr quantization tables is outside the HIP path");
    if (!planes[0] || hstride[0] < sc.w || vstride[0] < sc.h) return err_status(UHDR_CODEC_INVALID_PARAM, "destination smaller than the %ux%u image", sc.w, sc.h);
  }
  // the entropy-coded data ends at the first marker that is neither a stuffed zero, a fill byte nor RSTn (T.81 B.1.1.2 / B.2.1)
  auto walk = [&]() -> size_t {
    size_t e = 0;
    while (e < scan_bytes) {
      const uint8_t* f = (const uint8_t*)memchr(scan_data + e, 0xff, scan_bytes - e);
      if (!f) { e = scan_bytes; break; }
      e = (size_t)(f - scan_data);
      if (e + 1 >= scan_bytes) { e = scan_bytes; break; }
      const uint8_t m = scan_data[e + 1];
      if (m == 0x00 || (m & 0xf8) == 0xd0) { e += 2; continue; }
      if (m == 0xff) { e += 1; continue; }
      break;
    }
    return e;
  };
  // Round 5: that walk reads every byte on the host (0.1 ms for a 4K frame) to find what is nearly always the EOI marker in the
  // buffer's last two bytes.  So: take that for the end, and let the device -- which reads every byte anyway -- report any other
  // marker inside (stray_marker check, a pinned status word); only then is the walk made and the decode repeated on its prefix.
  static const bool always_walk = getenv("UHDR_HIP_JPEG_WALK") != nullptr;
  bool guessed = !always_walk && scan_bytes >= 3 && scan_bytes < 0xFFFFFFF0ull && scan_data[scan_bytes - 2] == 0xff && scan_data[scan_bytes - 1] == 0xd9;
  size_t nbytes = guessed ? scan_bytes - 2 : walk();
  if (nbytes == 0) return err_status(UHDR_CODEC_INVALID_PARAM, "no entropy-coded data");
  dbg.mark("jpeg_decode_scan: end of the entropy-coded data found");
  HIP_TRY(hipSetDevice(c->device));
  UHDR_TRY(ensure(c->jpg[0], nbytes + 64));
  HIP_TRY(hipMemcpyAsync(c->jpg[0].p, scan_data, nbytes, hipMemcpyHostToDevice, c->stream));
  dbg.mark("jpeg_decode_scan: compressed bytes on their way up");
  if (!c->h_flags) HIP_TRY(hipHostMalloc((void**)&c->h_flags, 64 * sizeof(uint32_t), hipHostMallocDefault));
  uint32_t* stray = c->h_flags + 40;
  if (guessed) {
    *stray = 0;
    HIP_TRY(launch_stray_marker_check((const uint8_t*)c->jpg[0].p, (uint32_t)nbytes, stray, c->stream));
  }
  for (int i = 0; i < nc; i++) {
    UHDR_TRY(ensure(c->jpg[1 + i], (size_t)sc.blocks_w[i] * sc.blocks_h[i] * 64 * sizeof(int16_t)));
    sc.coef[i] = (const int16_t*)c->jpg[1 + i].p;
  }
  c->huff_serial_ok = false;
  uhdr_error_info_t hs = uhdr_hip_huffman_decode_dev(c, &sc, &hdr->tables, (const uint8_t*)c->jpg[0].p, nbytes);
  if (guessed) {
    if (hs.error_code != UHDR_CODEC_OK) HIP_TRY(hipStreamSynchronize(c->stream));  // (an early return may have skipped the decoder's own)
    if (*stray != 0) {  // a marker inside what was taken for entropy-coded data: the data ends there (libjpeg stops at it too)
      guessed = false;
      nbytes = walk();
      if (nbytes == 0) { c->huff_serial_ok = true; return err_status(UHDR_CODEC_INVALID_PARAM, "no entropy-coded data"); }
      hs = uhdr_hip_huffman_decode_dev(c, &sc, &hdr->tables, (const uint8_t*)c->jpg[0].p, nbytes);
    }
  }
  c->huff_serial_ok = true;
  if (hs.error_code != UHDR_CODEC_OK) return hs;
  dbg.mark("jpeg_decode_scan: entropy decode returned");
  uhdr_hip_ctx::Resident* res = c->resident_on ? &c->resident[c->resident_next++ % 2] : nullptr;
  DeviceBuf* out_buf = res ? &res->buf : &c->jpg[4];
  if (res) {
    resident_retire(c, *res, false);
    const DeviceBuf keep = res->buf;
    *res = uhdr_hip_ctx::Resident();
    res->buf = keep;
    resident_drop(c, planes[0]);  // an older copy of what this call overwrites on the host
  }
  // lazy downloads (uhdr_hip_resident_lazy): an image the handoff keeps is not written to the caller's planes
  bool lazy = res && c->resident_lazy;
  if (lazy && out_channels == 0) {
    const int hs0 = nc == 3 ? sc.h_samp[0] : 1, vs0 = nc == 3 ? sc.v_samp[0] : 1;
    lazy = (nc == 1 || (sc.h_samp[1] == 1 && sc.v_samp[1] == 1 && sc.h_samp[2] == 1 && sc.v_samp[2] == 1)) && hs0 <= 2 && vs0 <= 2 && !(hs0 == 1 && vs0 == 2);
  }
  if (out_channels == 0) {
    size_t pitch[3] = {0, 0, 0}, off[3] = {0, 0, 0}, total = 0;
    for (int i = 0; i < nc; i++) {
      pitch[i] = ((size_t)sc.blocks_w[i] * 8 + 63) & ~(size_t)63;
      off[i] = total;
      total += pitch[i] * (size_t)sc.blocks_h[i] * 8;
    }
    UHDR_TRY(ensure(*out_buf, total));
    for (int i = 0; i < nc; i++) {
      if (!planes[i]) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for destination plane %d", i);
      uint8_t* d = (uint8_t*)out_buf->p + off[i];
      UHDR_TRY(uhdr_hip_idct_dequant_dev(c, sc.coef[i], sc.blocks_w[i], sc.blocks_h[i], hdr->qtable[i], d, pitch[i]));
      const size_t cols = hstride[i] < (unsigned)sc.blocks_w[i] * 8 ? hstride[i] : (size_t)sc.blocks_w[i] * 8;
      const size_t rows = vstride[i] < (unsigned)sc.blocks_h[i] * 8 ? vstride[i] : (size_t)sc.blocks_h[i] * 8;
      if (!lazy) HIP_TRY(hipMemcpy2DAsync(planes[i], hstride[i], d, pitch[i], cols, rows, hipMemcpyDeviceToHost, c->stream));
      if (res) {
        res->host[i] = planes[i]; res->host_stride[i] = hstride[i]; res->off[i] = off[i]; res->dev_stride[i] = (unsigned int)pitch[i];
        res->prows[i] = (unsigned int)rows; res->pcols[i] = (unsigned int)cols;
      }
    }
    if (res) {
      const int hs0 = nc == 3 ? sc.h_samp[0] : 1, vs0 = nc == 3 ? sc.v_samp[0] : 1;
      res->fmt = nc == 1 ? UHDR_IMG_FMT_8bppYCbCr400
                 : hs0 == 2 && vs0 == 2 ? UHDR_IMG_FMT_12bppYCbCr420
                 : hs0 == 2 && vs0 == 1 ? UHDR_IMG_FMT_16bppYCbCr422
                 : hs0 == 1 && vs0 == 1 ? UHDR_IMG_FMT_24bppYCbCr444 : UHDR_IMG_FMT_UNSPECIFIED;
      res->w = hstride[0] < (unsigned)sc.blocks_w[0] * 8 ? hstride[0] : (unsigned)sc.blocks_w[0] * 8;
      res->h = vstride[0] < (unsigned)sc.blocks_h[0] * 8 ? vstride[0] : (unsigned)sc.blocks_h[0] * 8;
      bool plain = nc == 1 || (sc.h_samp[1] == 1 && sc.v_samp[1] == 1 && sc.h_samp[2] == 1 && sc.v_samp[2] == 1);
      res->valid = plain && res->fmt != UHDR_IMG_FMT_UNSPECIFIED;
      if (lazy && !res->valid) return err_status(UHDR_CODEC_ERROR, "internal: lazy download of an image the handoff does not keep");
      res->host_unwritten = lazy;
    }
  } else {
    uhdr_raw_image_t rgb;
    memset(&rgb, 0, sizeof rgb);
    rgb.fmt = out_channels == 3 ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_32bppRGBA8888;
    rgb.w = sc.w;
    rgb.h = sc.h;
    const size_t pitch_px = ((size_t)sc.w + 63) & ~(size_t)63;
    UHDR_TRY(ensure(*out_buf, pitch_px * (size_t)out_channels * sc.h));
    rgb.planes[0] = out_buf->p;
    rgb.stride[0] = (unsigned int)pitch_px;
    UHDR_TRY(uhdr_hip_idct_dequant_rgb_dev(c, sc.coef[0], sc.coef[1], sc.coef[2], sc.blocks_w[0], sc.blocks_h[0], hdr->qtable[0], hdr->qtable[1], variant, &rgb));
    if (!lazy)
      HIP_TRY(hipMemcpy2DAsync(planes[0], (size_t)hstride[0] * out_channels, out_buf->p, pitch_px * out_channels, (size_t)sc.w * out_channels, sc.h,
                               hipMemcpyDeviceToHost, c->stream));
    if (res) {
      res->fmt = rgb.fmt; res->w = sc.w; res->h = sc.h;
      res->host[0] = planes[0]; res->host_stride[0] = hstride[0]; res->off[0] = 0; res->dev_stride[0] = (unsigned int)pitch_px;
      res->prows[0] = sc.h; res->pcols[0] = sc.w;
      res->valid = true;
      res->host_unwritten = lazy;
    }
  }
  if (lazy) c->stats.lazy_downloads_skipped++;
  dbg.mark("jpeg_decode_scan: IDCT (and download) enqueued");
  // lazy: nothing was copied to the caller's planes and whoever reads the device copy does so on this stream, in order -- no
  // host synchronisation (the entropy decoder above has made its own: malformed data has surfaced by now)
  if (!lazy) HIP_TRY(hipStreamSynchronize(c->stream));
  dbg.mark("jpeg_decode_scan: done");
  return ok_status();
}

// Device-resident handoff between the decode and the apply stage of one uhdr_decode (JpegR::decodeJPEGR, jpegr.cpp:1467-
// 1530): the planes uhdr_hip_jpeg_decode_scan wrote into the caller's buffers stay on the device until _end, and the host
// variant of uhdr_hip_apply_gainmap, handed exactly those buffers (same plane pointers, strides and format), reads the
// device copy instead of uploading 45 MB (4K base + scale-1 RGBA map) it has just downloaded.  The caller promises not
// to write to those host buffers in between -- decodeJPEGR's are private to its JpegDecoderHelper locals.
void uhdr_hip_resident_begin(uhdr_hip_ctx_t* c) {
  if (!c) return;
  c->resident_on = true;
  c->resident_lazy = false;
  c->pending.on = false;
  // reopened inside a session (the facade, when a stage falls back to the reference's CPU code): that code reads the host planes
  for (auto& r : c->resident) resident_retire(c, r, false);
}
void uhdr_hip_resident_end(uhdr_hip_ctx_t* c) {
  if (!c) return;
  c->resident_on = false;
  c->resident_lazy = false;
  for (auto& r : c->resident) {
    if (r.valid && r.adopted) {  // the copy the caller left to the library outlives the session: the buffer changes hands
      uhdr_hip_ctx::PendingCopy& p = c->pending;
      std::swap(p.buf, r.buf);
      const size_t bps = bytes_per_sample(r.fmt);
      p.off = r.off[0]; p.pitch = (size_t)r.dev_stride[0] * bps; p.w = r.adopt_w; p.h = r.adopt_h; p.bps = (unsigned int)bps;
      p.expand = r.adopt_expand;
      p.dst = r.adopt_dst; p.dst_pitch = (size_t)r.adopt_stride * (r.adopt_expand ? 4 : bps);
      p.on = true;
    }
    r.valid = false;
    r.host_unwritten = false;
    r.adopted = false;
  }
}
void uhdr_hip_resident_lazy(uhdr_hip_ctx_t* c, int on) {
  if (c) c->resident_lazy = c->resident_on && on != 0;
}
uhdr_error_info_t uhdr_hip_resident_flush(uhdr_hip_ctx_t* c) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipSetDevice(c->device));
  if (c->sticky.error_code != UHDR_CODEC_OK) {
    const uhdr_error_info_t e = c->sticky;
    c->sticky = ok_status();
    return e;
  }
  return resident_write_back_all(c);
}
int uhdr_hip_resident_adopt(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* src, const uhdr_raw_image_t* dst) {
  if (!c || !src || !dst || !c->resident_on || !dst->planes[0]) return 0;
  const bool expand = src->fmt == UHDR_IMG_FMT_24bppRGB888 && dst->fmt == UHDR_IMG_FMT_32bppRGBA8888;
  if ((src->fmt != dst->fmt && !expand) || src->w != dst->w || src->h != dst->h || dst->stride[0] < dst->w) return 0;
  if (src->fmt != UHDR_IMG_FMT_8bppYCbCr400 && src->fmt != UHDR_IMG_FMT_24bppRGB888 && src->fmt != UHDR_IMG_FMT_32bppRGBA8888) return 0;
  for (auto& r : c->resident) if (r.adopted) return 0;  // one at a time
  for (auto& r : c->resident) {
    if (!r.valid || !r.host_unwritten || r.fmt != src->fmt || r.host[0] != src->planes[0] || r.host_stride[0] != src->stride[0]) continue;
    if (src->w > r.pcols[0] || src->h > r.prows[0]) continue;
    r.adopted = true;
    r.adopt_dst = dst->planes[0];
    r.adopt_stride = dst->stride[0];
    r.adopt_w = src->w;
    r.adopt_h = src->h;
    r.adopt_expand = expand;
    return 1;
  }
  return 0;
}
uhdr_error_info_t uhdr_hip_resident_materialize(uhdr_hip_ctx_t* c) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  HIP_TRY(hipSetDevice(c->device));
  UHDR_TRY(resident_write_back_all(c));  // inside the session: the same as _flush
  uhdr_hip_ctx::PendingCopy& p = c->pending;
  if (!p.on) return ok_status();
  UHDR_TRY(adopted_copy_out(c, (const char*)p.buf.p + p.off, p.pitch, p.bps, p.expand, p.w, p.h, p.dst, p.dst_pitch));
  HIP_TRY(hipStreamSynchronize(c->stream));
  p.on = false;
  c->stats.lazy_downloads_done++;
  return ok_status();
}
void uhdr_hip_resident_forget(uhdr_hip_ctx_t* c) {
  if (!c) return;
  c->pending.on = false;
  for (auto& r : c->resident) r.adopted = false;
}

// Host helper: a complete baseline JFIF file around entropy-coded data (marker order of jcmarker.c: SOI, APP0, DQT,
// SOF0, DHT, DRI, SOS ... EOI).  Returns the file size, or 0 when `cap` is too small / the description is invalid.
size_t uhdr_hip_jpeg_assemble(const uhdr_hip_jpeg_scan_t* sc, const uint16_t qt_luma[64], const uint16_t qt_chroma[64], const uint8_t* scan_data,
                              size_t scan_bytes, uint8_t* out, size_t cap) {
  int mpr = 0, mrows = 0, bpm = 0;
  if (check_scan(sc, false, &mpr, &mrows, &bpm).error_code != UHDR_CODEC_OK || !qt_luma || !scan_data || !out) return 0;
  if (sc->num_components > 1 && !qt_chroma) return 0;
  const int nc = sc->num_components, ntab = nc > 1 ? 2 : 1;
  const uint8_t* zz = host::jpeg_zigzag_to_natural();
  std::vector<uint8_t> v;
  v.reserve(scan_bytes + 1024);
  auto put = [&](unsigned b) { v.push_back((uint8_t)b); };
  auto put16 = [&](unsigned x) { put(x >> 8); put(x & 0xff); };
  put(0xff); put(0xd8);
  put(0xff); put(0xe0); put16(16); for (char ch : {'J', 'F', 'I', 'F'}) put((unsigned char)ch); put(0); put(1); put(1); put(0); put16(1); put16(1); put(0); put(0);
  for (int t = 0; t < ntab; t++) {
    const uint16_t* q = t ? qt_chroma : qt_luma;
    put(0xff); put(0xdb); put16(67); put(t);
    for (int i = 0; i < 64; i++) {
      if (q[zz[i]] == 0 || q[zz[i]] > 255) return 0;  // baseline: 8-bit tables
      put(q[zz[i]]);
    }
  }
  put(0xff); put(0xc0); put16(8 + 3 * nc); put(8); put16(sc->h); put16(sc->w); put(nc);
  for (int i = 0; i < nc; i++) { put(i + 1); put(((nc == 1 ? 1 : sc->h_samp[i]) << 4) | (nc == 1 ? 1 : sc->v_samp[i])); put(i ? 1 : 0); }
  for (int t = 0; t < ntab; t++) {
    for (int ac = 0; ac < 2; ac++) {
      uint8_t bits[17], vals[256];
      const int nv = host::jpeg_std_huff_table(ac, t, bits, vals);
      put(0xff); put(0xc4); put16(2 + 1 + 16 + nv); put((ac << 4) | t);
      for (int i = 1; i <= 16; i++) put(bits[i]);
      for (int i = 0; i < nv; i++) put(vals[i]);
    }
  }
  if (sc->restart_interval > 0) { put(0xff); put(0xdd); put16(4); put16((unsigned)sc->restart_interval); }
  put(0xff); put(0xda); put16(6 + 2 * nc); put(nc);
  for (int i = 0; i < nc; i++) { put(i + 1); put(i ? 0x11 : 0x00); }
  put(0); put(63); put(0);
  v.insert(v.end(), scan_data, scan_data + scan_bytes);
  put(0xff); put(0xd9);
  if (v.size() > cap) return 0;
  memcpy(out, v.data(), v.size());
  return v.size();
}

}  // extern "C"
#pragma GCC visibility pop
