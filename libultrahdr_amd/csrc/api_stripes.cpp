// api_stripes.cpp -- one process per GPU, images sharded by row stripe: the one exchange step of the hot path (see api_internal.h).
#include "api_internal.h"

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

namespace uhdr_api {
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
}  // namespace uhdr_api

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
  return generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
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
