// api_context.cpp -- the context and everything a call needs around its kernels (see api_internal.h).
#include "api_internal.h"

#include <sys/mman.h>
#include <mutex>
#include <algorithm>

namespace uhdr_api {
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
  // Round 6: the runtime's pageable path only for small copies.  A caller's buffer of a few MB (the compressed scans of a 4K file) that the
  // runtime had never seen stalled its staged upload for 10-16 ms one call in four (profiles/r05_decode_stalls.txt); round 5 dodged that
  // with a thread-lifetime copy in the facade.  From 64 KiB on the bytes go through this context's pinned ring instead -- below 4 MiB copied in
  // by the calling thread alone -- so that the runtime only ever sees pinned memory.
  if (nthreads == 0 || bytes < ((size_t)64 << 10)) {
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    return ok_status();
  }
  const bool threaded = bytes >= 4 * kPiece;
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
  if (!threaded) {
    memcpy(stage, src, bytes);
    HIP_TRY(hipMemcpyAsync(dst, stage, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipEventRecord(pa.ev, c->stream));
    pa.ev_pending = true;
    pa.off += need;
    return ok_status();
  }
  const size_t npieces = (bytes + kPiece - 1) / kPiece;
  std::unique_ptr<std::atomic<unsigned char>[]> done(new (std::nothrow) std::atomic<unsigned char>[npieces]);
  if (!done) return err_status(UHDR_CODEC_MEM_ERROR, "could not allocate %zu bytes of upload bookkeeping", npieces);
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
  try {
    pool.reserve((size_t)nt);
  } catch (...) {  // (bad_alloc must not cross the C ABI: this thread alone copies, see below)
  }
  for (int t = 0; t < nt && pool.capacity() > pool.size(); t++) {
    try {
      pool.emplace_back(work);
    } catch (...) {  // no thread to be had (a process at its limit): whoever started, or this thread alone, does the copying
      break;
    }
  }
  if (pool.empty()) work();
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
uhdr_error_info_t resident_keep(uhdr_hip_ctx* c, const uhdr_raw_image_t* dev, const uhdr_raw_image_t* host, bool host_unwritten) {
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
// A large download lands in caller-owned pageable memory that may never have been touched (a freshly allocated output image:
// the facade's blocks come from calloc, facade/make_patch.py).  The runtime pins such a destination in place, which takes the
// page faults one after another on the calling thread: 66 MB of RGBA_F16 took 5.4 ms instead of the 1.8 ms of a populated
// buffer (profiles/r05_api_trace.txt).  MADV_POPULATE_WRITE on a few threads (the upload's thread count), after asking for huge
// pages, takes the faults in parallel and leaves contents alone; on populated memory it is a page-table walk.  Failure of the hint is not an error.
static void populate_pages(void* p, size_t bytes) {
  static const int nthreads = [] {
    const char* e = getenv("UHDR_HIP_UPLOAD_THREADS");
    int v = e ? atoi(e) : 4;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && (unsigned)v > hw) v = (int)hw;
    return v < 0 ? 0 : (v > 16 ? 16 : v);
  }();
  if (nthreads == 0 || bytes < ((size_t)8 << 20)) return;
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
  const uintptr_t pg = 4096, a = ((uintptr_t)p + pg - 1) & ~(pg - 1), e = ((uintptr_t)p + bytes) & ~(pg - 1);
  if (e <= a) return;
  (void)madvise((void*)a, e - a, MADV_HUGEPAGE);  // where the system allows it on request: 2 MiB faults, 3 x fewer ms (66 MB: 1.0 against 3.0)
  const size_t per = (((e - a) / (size_t)nthreads) + pg - 1) & ~(size_t)(pg - 1);
  std::vector<std::thread> pool;
  for (uintptr_t o = a; o < e; o += per) {
    const size_t n = o + per <= e ? per : e - o;
    try {
      pool.emplace_back([o, n] { (void)madvise((void*)o, n, MADV_POPULATE_WRITE); });
    } catch (...) {  // no thread to be had: the rest of the range is faulted in by the copy itself
      break;
    }
  }
  for (auto& t : pool) t.join();
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
      populate_pages(host->planes[pl], ((rows - 1) * (size_t)host->stride[pl] + width) * bps);
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

// Linearisation table of an HDR input for the encode kernels: inverse OETF with, for HLG, the
// per-channel OOTF (hlgOotfApprox, gainmapmath.cpp:293-295) folded in node by node.

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
uhdr_error_info_t upload_step_table(const host::OetfBuckets& b, float** slot, StepTab* meta, hipStream_t s) {
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
uhdr_error_info_t gain_step_table(uhdr_hip_ctx* c, const GenParams& p, StepTab* out) {
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
}  // namespace uhdr_api

// -------------------------------------------------------------------------------------------------
// context
// -------------------------------------------------------------------------------------------------

const char* uhdr_hip_version(void) { return "libuhdr_hip 0.3 (gfx950; reference libultrahdr 2.0.2 hot path)"; }

int uhdr_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int uhdr_hip_current_device(void) {
  int d = -1;
  if (hipGetDevice(&d) != hipSuccess) return -1;
  return d;
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
  aux_worker_destroy(c);
  if (c->aux) uhdr_hip_destroy(c->aux);
  if (c->exchange.p) (void)hipFree(c->exchange.p);
  if (c->affine.p) (void)hipFree(c->affine.p);
  if (c->d_srgb_of_byte) (void)hipFree(c->d_srgb_of_byte);
  if (c->h_mm) (void)hipHostFree(c->h_mm);
  if (c->aux_ev) (void)hipEventDestroy(c->aux_ev);
  if (c->aux_ev2) (void)hipEventDestroy(c->aux_ev2);
  if (c->md_ev) (void)hipEventDestroy(c->md_ev);
  if (c->md_stream) (void)hipStreamDestroy(c->md_stream);
  if (c->h_flags) (void)hipHostFree(c->h_flags);
  if (c->pin.p) (void)hipHostFree(c->pin.p);
  if (c->pin.ev) (void)hipEventDestroy(c->pin.ev);
  if (c->huff_tabs.dev.p) (void)hipFree(c->huff_tabs.dev.p);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int uhdr_hip_recycle(uhdr_hip_ctx_t* c, size_t keep_bytes) {
  if (!c) return -1;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  uhdr_hip_resident_forget(c);
  uhdr_hip_resident_begin(c);  // drops the resident copies (nothing of the previous user's is found by the next one)
  uhdr_hip_resident_end(c);
  c->sticky = ok_status();
  for (auto& h : c->huff_hint) h = uhdr_hip_ctx::HuffHint();
  c->stats = uhdr_hip_stats_t();
  c->deferred_md.valid = false;
  // the retained buffers, largest first, until the rest fits keep_bytes
  std::vector<DeviceBuf*> bufs;
  for (auto& b : c->scratch) bufs.push_back(&b);
  for (auto& b : c->jpg) bufs.push_back(&b);
  for (auto& b : c->enc) bufs.push_back(&b);
  for (auto& r : c->resident) bufs.push_back(&r.buf);
  bufs.push_back(&c->pending.buf);
  bufs.push_back(&c->pending.tmp);
  size_t total = c->pin.cap;
  for (DeviceBuf* b : bufs) total += b->p ? b->cap : 0;
  std::sort(bufs.begin(), bufs.end(), [](const DeviceBuf* a, const DeviceBuf* b) { return (a->p ? a->cap : 0) > (b->p ? b->cap : 0); });
  for (DeviceBuf* b : bufs) {
    if (total <= keep_bytes) break;
    if (!b->p) continue;
    (void)hipFree(b->p);
    total -= b->cap;
    b->p = nullptr;
    b->cap = 0;
  }
  if (total > keep_bytes && c->pin.p) {
    if (c->pin.ev_pending && c->pin.ev) (void)hipEventSynchronize(c->pin.ev);
    (void)hipHostFree(c->pin.p);
    c->pin.p = nullptr;
    c->pin.cap = c->pin.off = 0;
    c->pin.ev_pending = false;
  }
  if (c->aux) (void)uhdr_hip_recycle(c->aux, keep_bytes);
  return c->device;
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

// ---- the seam's stage tallies (include/uhdr_hip.h) ------------------------------------------------------------------
namespace {
std::mutex g_seam_mu;
std::vector<uhdr_hip_seam_stage_t> g_seam_rows;
unsigned long long g_seam_seq = 0;
// UHDR_HIP_SEAM_STATS_FILE=<path>: the table is written there as JSON when the process exits -- how a test reads it out of a
// process it cannot call into (the reference's own ultrahdr_app linked against the facade)
void seam_dump_at_exit() {
  const char* path = getenv("UHDR_HIP_SEAM_STATS_FILE");
  FILE* f = path ? fopen(path, "w") : nullptr;
  if (!f) return;
  std::lock_guard<std::mutex> lk(g_seam_mu);
  fprintf(f, "{");
  for (size_t i = 0; i < g_seam_rows.size(); i++) {
    const auto& r = g_seam_rows[i];
    fprintf(f, "%s\"%s\": {\"device\": %llu, \"reference\": %llu, \"device_ms\": %.3f, \"first_seq\": %llu}", i ? ", " : "", r.name, r.device_calls,
            r.reference_calls, r.device_ms, r.first_seq);
  }
  fprintf(f, "}\n");
  fclose(f);
}
}  // namespace

void uhdr_hip_seam_note(const char* stage, int on_device, double ms) {
  if (!stage) return;
  static const bool dump = [] { return getenv("UHDR_HIP_SEAM_STATS_FILE") ? (atexit(seam_dump_at_exit), true) : false; }();
  (void)dump;
  std::lock_guard<std::mutex> lk(g_seam_mu);
  uhdr_hip_seam_stage_t* row = nullptr;
  for (auto& r : g_seam_rows)
    if (!strncmp(r.name, stage, sizeof r.name - 1)) { row = &r; break; }
  if (!row) {
    if (g_seam_rows.size() >= 64) return;  // the facade has 16 stage names
    uhdr_hip_seam_stage_t r;
    memset(&r, 0, sizeof r);
    strncpy(r.name, stage, sizeof r.name - 1);
    r.first_seq = g_seam_seq;
    g_seam_rows.push_back(r);
    row = &g_seam_rows.back();
  }
  g_seam_seq++;
  if (on_device) { row->device_calls++; row->device_ms += ms; }
  else row->reference_calls++;
  row->last_ms = ms;
}

int uhdr_hip_seam_stats(uhdr_hip_seam_stage_t* out, int capacity) {
  std::lock_guard<std::mutex> lk(g_seam_mu);
  const int n = (int)g_seam_rows.size();
  for (int i = 0; out && i < n && i < capacity; i++) out[i] = g_seam_rows[i];
  return n;
}

void uhdr_hip_seam_stats_reset(void) {
  std::lock_guard<std::mutex> lk(g_seam_mu);
  g_seam_rows.clear();
  g_seam_seq = 0;
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
