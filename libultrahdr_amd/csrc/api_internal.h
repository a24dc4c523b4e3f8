// api_internal.h -- shared by the five host translation units behind the C ABI of libuhdr_hip.so (include/uhdr_hip.h):
//   api_context.cpp   context, device buffers, host <-> device staging, the device-resident handoff between calls, profiling hooks
//   api_gainmap.cpp   applyGainMap, generateGainMap, toneMap, convertYuv / convert_raw_input_to_ycbcr, effects, copy_raw_image
//   api_stripes.cpp   row-striped two-pass generation across GPUs: communicator, the one all-reduce, the gathers
//   api_jpeg.cpp      FDCT / IDCT stages, the fused encode chains, uhdr_hip_jpeg_encode_* / _decode_scan, file assembly
//   api_entropy.cpp   baseline Huffman coding: encoder launch logic, the decode forms of a DHT set, the attempt ladder of the decoder
// (round 5: these were one 3 700-line file).  Argument validation mirrors the reference operators; there is deliberately NO CPU
// implementation behind the entry points: if HIP is unusable the calls fail with UHDR_CODEC_ERROR.
#ifndef UHDR_HIP_API_INTERNAL_H
#define UHDR_HIP_API_INTERNAL_H

// every entry point include/uhdr_hip.h declares is exported (the library is built with -fvisibility=hidden and a version script
// that lets uhdr_hip_* through); everything else in these files stays internal
#pragma GCC visibility push(default)
#include "uhdr_hip.h"
#pragma GCC visibility pop

#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <functional>
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

inline uhdr_error_info_t ok_status() {
  uhdr_error_info_t s;
  s.error_code = UHDR_CODEC_OK;
  s.has_detail = 0;
  s.detail[0] = 0;
  return s;
}
inline uhdr_error_info_t err_status(uhdr_codec_err_t code, const char* fmt, ...) {
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
  // round 5: a second context on the same device (own stream, scratch, table cache), created on first use by the entry points that
  // code the TWO scans of an UltraHDR file concurrently (uhdr_hip_huffman_encode2_dev / _decode2_dev): the entropy stages are
  // latency- and occupancy-bound for long stretches (one-workgroup scans, the write pass, the stragglers), which two streams overlap
  uhdr_hip_ctx* aux = nullptr;
  // round 6: the thread that drives the auxiliary context, kept for the life of the context (a std::thread per call cost 40-60 us of
  // an 800 us round trip: creation, first scheduling, join); see AuxWorker in api_entropy.cpp
  struct AuxWorker* aux_worker = nullptr;
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
  CoefSrc* coef_src_last_slot = nullptr;  // the slot of the latest upload, and what went up: the same descriptor again takes no copy
  std::vector<uint8_t> coef_src_last;
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
  // round 6: uhdr_hip_encode_api1_scans_dev lets the entropy stage's launches follow the fused chain without a host synchronisation in
  // between: the chain leaves the metadata's inputs here (the copy to h_mm is enqueued) and the entry point finishes them after the
  // entropy stage's own synchronisation
  bool defer_md = false;
  // (round 6) with defer_md the range's read-back leaves the main stream: an event behind the range kernel, the copy on md_stream -- it used to sit, with its
  // launch gaps, between the map's blocks and the map scan's first entropy kernel
  hipStream_t md_stream = nullptr;
  hipEvent_t md_ev = nullptr;
  // uhdr_hip_encode_api1_scans_dev (round 6): called by the fused chain as soon as the base image's blocks are enqueued on the auxiliary stream -- the
  // caller posts the base scan's entropy coding to the worker thread there, two kernels into the chain instead of behind it.  side_job_posted: the
  // auxiliary context belongs to that job until the caller has waited for it.
  std::function<void()> on_side_launched;
  bool side_job_posted = false;
  struct DeferredMd { bool valid = false, run = false; uhdr_hip_encode_cfg_t cfg; uhdr_color_transfer_t hdr_ct; int use_base_cg = 1; } deferred_md;
  hipEvent_t aux_ev = nullptr;   // orders the auxiliary context's stream behind this one (two-scan entropy entry points)
  hipEvent_t aux_ev2 = nullptr;  // ... and this one behind the auxiliary stream (the fused API-1 chain's base-image launch)
  // profiling
  bool prof = false;
  std::vector<ProfEntry> prof_entries;
};

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

namespace uhdr_api {
// ---- api_context.cpp: device buffers, image geometry, host <-> device staging, the device-resident handoff, per-call tables ----
uhdr_error_info_t ensure(DeviceBuf& b, size_t bytes);
size_t bytes_per_sample(int fmt);
bool plane_geom(const uhdr_raw_image_t* im, int pl, size_t* rows, size_t* width);
size_t plane_bytes(const uhdr_raw_image_t* im, int pl);
uhdr_error_info_t validate_image(const uhdr_raw_image_t* im, const char* what);
ImageView view_of(const uhdr_raw_image_t* im);
ImageViewMut view_mut_of(const uhdr_raw_image_t* im);
uhdr_error_info_t adopted_copy_out(uhdr_hip_ctx* c, const void* src, size_t src_pitch, unsigned int bps, bool expand, unsigned int w, unsigned int h,
                                   void* dst, size_t dst_pitch);
uhdr_error_info_t resident_write_back(uhdr_hip_ctx* c, uhdr_hip_ctx::Resident& r);
uhdr_error_info_t resident_write_back_all(uhdr_hip_ctx* c);
void resident_retire(uhdr_hip_ctx* c, uhdr_hip_ctx::Resident& r, bool host_is_rewritten);
void resident_drop(uhdr_hip_ctx* c, const void* host_plane);
uhdr_error_info_t fast_h2d(uhdr_hip_ctx* c, void* dst, const void* src, size_t bytes);
uhdr_error_info_t stage_in(uhdr_hip_ctx* c, int slot, const uhdr_raw_image_t* host, uhdr_raw_image_t* dev, bool upload);
bool resident_keeps(int fmt);
uhdr_error_info_t resident_keep(uhdr_hip_ctx* c, const uhdr_raw_image_t* dev, const uhdr_raw_image_t* host, bool host_unwritten = false);
uhdr_error_info_t stage_out(uhdr_hip_ctx* c, const uhdr_raw_image_t* dev, uhdr_raw_image_t* host);
uhdr_error_info_t upload_lut(float** dst, const std::vector<float>& src, hipStream_t s);
uhdr_error_info_t upload_math(uhdr_hip_ctx* c);
uhdr_error_info_t upload_step_table(const host::OetfBuckets& b, float** slot, StepTab* meta, hipStream_t s);
uhdr_error_info_t gain_step_table(uhdr_hip_ctx* c, const GenParams& p, StepTab* out);
uhdr_error_info_t select_hdr_lut(uhdr_hip_ctx* c, uhdr_color_transfer_t ct, const float** lut, int* n);
uhdr_error_info_t validate_metadata(const uhdr_gainmap_metadata_t* m);
bool is_rgb_fmt_host(int fmt);
uhdr_error_info_t validate_apply(const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* gm, const uhdr_gainmap_metadata_t* md, uhdr_color_transfer_t out_ct,
                                 const uhdr_raw_image_t* dest);
uhdr_error_info_t get_apply_tables(uhdr_hip_ctx* c, const uhdr_gainmap_metadata_t& md, float weight, int idw_scale, const float** d_out);
// ---- api_gainmap.cpp: parameter blocks of the gain-map operators (shared with the striped and the fused entry points) ----
uhdr_error_info_t fill_gen_params(uhdr_hip_ctx* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg, GenParams* p,
                                  int* use_base_cg, float* hdr_white_nits_out, bool sdr_in_registers = false);
void fill_finalize(MinmaxTableParams* t, const uhdr_hip_encode_cfg_t* cfg);
void fill_gainmap_desc(const uhdr_raw_image_t* hdr, const GenParams& p, uhdr_raw_image_t* gm);
uhdr_error_info_t generate_gainmap_finalize_md(const uhdr_hip_encode_cfg_t* cfg, uhdr_color_transfer_t hdr_ct, int use_base_cg, const float mm[6], uhdr_gainmap_metadata_t* md);
void note_table_stats(uhdr_hip_ctx* c, const uhdr_hip_encode_cfg_t* cfg);
uhdr_error_info_t two_pass_tail(uhdr_hip_ctx* c, const GenParams& p, int n_partials, const uhdr_hip_encode_cfg_t* cfg, uhdr_raw_image_t* gm);
uhdr_error_info_t fill_tone_map_params(uhdr_hip_ctx* c, const uhdr_raw_image_t* hdr, ToneMapParams* pp);
// ---- api_stripes.cpp: THE collective of the hot path ----
uhdr_error_info_t comm_all_reduce_min(uhdr_hip_ctx* c, float* buf, size_t n);
// ---- api_entropy.cpp ----
uhdr_error_info_t check_scan(const uhdr_hip_jpeg_scan_t* sc, bool need_coef, int* mcus_per_row, int* mcu_rows, int* blocks_per_mcu);
// the context's auxiliary context (own stream, scratch, table cache; created on first use) and the hand-back of what it counted / timed
uhdr_error_info_t aux_context(uhdr_hip_ctx* c, uhdr_hip_ctx** out);
void aux_merge(uhdr_hip_ctx* c);
// a job for the context's worker thread (created on first use; false: no thread to be had, nothing posted) / wait for it
bool aux_post(uhdr_hip_ctx* c, std::function<void()> job);
void aux_wait(uhdr_hip_ctx* c);
}  // namespace uhdr_api
using namespace uhdr_api;

void aux_worker_destroy(uhdr_hip_ctx* c);  // api_entropy.cpp: stops and joins the context's worker thread

#endif  // UHDR_HIP_API_INTERNAL_H
