// api_gainmap.cpp -- the gain-map operators of UltraHdr (jpegr.cpp:436-2203, gainmapmath.cpp:1291-1613) behind the C ABI (see api_internal.h).
#include "api_internal.h"

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
  CoefSrc* slot = c->coef_src_last_slot;
  if (!slot || c->coef_src_last.size() != sizeof cs || memcmp(c->coef_src_last.data(), &cs, sizeof cs) != 0) {
    // (a service decodes frame after frame of one geometry into the same buffers: the descriptor of the latest upload is the one wanted again,
    // and a 7 us staged copy in front of every launch goes away)
    slot = c->d_coef_src + (c->coef_src_next++ % kSlots);
    HIP_TRY(hipMemcpyAsync(slot, &cs, sizeof cs, hipMemcpyHostToDevice, c->stream));  // pageable source: staged before return
    c->coef_src_last.assign((const uint8_t*)&cs, (const uint8_t*)&cs + sizeof cs);
    c->coef_src_last_slot = slot;
  }
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
uhdr_error_info_t uhdr_api::fill_gen_params(uhdr_hip_ctx* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                         const uhdr_hip_encode_cfg_t* cfg, GenParams* p, int* use_base_cg,
                                         float* hdr_white_nits_out, bool sdr_in_registers) {
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
void uhdr_api::fill_finalize(MinmaxTableParams* t, const uhdr_hip_encode_cfg_t* cfg) {
  t->nch = (cfg && cfg->use_multi_channel_gainmap) ? 3 : 1;
  t->has_max_hint = cfg && cfg->max_content_boost != FLT_MAX;
  t->has_min_hint = cfg && cfg->min_content_boost != FLT_MIN;
  t->log2_max_hint = t->has_max_hint ? log2f(cfg->max_content_boost) : 0.0f;
  t->log2_min_hint = t->has_min_hint ? log2f(cfg->min_content_boost) : 0.0f;
  t->gamma = cfg ? cfg->gamma : 1.0f;
}

void uhdr_api::fill_gainmap_desc(const uhdr_raw_image_t* hdr, const GenParams& p, uhdr_raw_image_t* gm) {
  gm->fmt = p.multichannel ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_8bppYCbCr400;
  gm->cg = hdr->cg; gm->ct = hdr->ct; gm->range = hdr->range;
  gm->w = p.map_w; gm->h = p.map_h;
}

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
  return generate_gainmap_finalize_md(cfg, hdr_ct, use_base_cg, mm, md);
}

// the metadata fill of jpegr.cpp:1031-1048 from a FINAL per-channel range
uhdr_error_info_t uhdr_api::generate_gainmap_finalize_md(const uhdr_hip_encode_cfg_t* cfg, uhdr_color_transfer_t hdr_ct,
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

void uhdr_api::note_table_stats(uhdr_hip_ctx* c, const uhdr_hip_encode_cfg_t* cfg) {  // after the synchronisation that landed c->h_mm
  const int nch = cfg->use_multi_channel_gainmap ? 3 : 1;
  for (int i = 0; i < nch; i++) {
    if (c->h_mm[6 + i] != 0.0f) c->stats.generate_channels_tabled++;
    else c->stats.generate_channels_per_sample++;
  }
}

// pass 1's partials -> extrema -> final range -> step tables -> pass 2, all stream ordered; the final range is copied to
// the pinned c->h_mm for the caller's metadata fill (after ITS synchronisation)
uhdr_error_info_t uhdr_api::two_pass_tail(uhdr_hip_ctx* c, const GenParams& p, int n_partials, const uhdr_hip_encode_cfg_t* cfg, uhdr_raw_image_t* gm) {
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
  return generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
}

// -------------------------------------------------------------------------------------------------
// toneMap
// -------------------------------------------------------------------------------------------------
// everything of ToneMapParams that depends on the HDR image only (the caller sets p->sdr)
uhdr_error_info_t uhdr_api::fill_tone_map_params(uhdr_hip_ctx* c, const uhdr_raw_image_t* hdr, ToneMapParams* pp) {
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
