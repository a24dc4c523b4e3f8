// api_jpeg.cpp -- the JPEG block stages and the fused encode / decode chains around them (see api_internal.h).
#include "api_internal.h"

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
  return generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
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
  bool md_on_side = false;
  auto note_hip = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && local.error_code == UHDR_CODEC_OK) local = err_status(UHDR_CODEC_ERROR, "%s: %s", what, hipGetErrorString(e));
  };
  // profiling: the per-stage families below, and ONE event pair around the whole chain ("encode_api1_chain": first launch's start to
  // last launch's end, the gaps between the four launches included) -- destroyed, i.e. recorded, before the metadata copy
  std::unique_ptr<ProfScope> chain(new ProfScope(c, "encode_api1_chain"));
  // Round 6: the base image's launch (convertYuv + three FDCTs) depends on nothing the gain-map passes compute -- it runs on the auxiliary
  // stream UNDER pass 1 / the range kernel / the map's blocks instead of behind them (the chain's four launches were strictly serial: 21 of
  // 103 us at 4K).  Whole images only (a stripe's chain keeps its order around the exchange); UHDR_HIP_NO_CHAIN_OVERLAP=1: the serial form.
  uhdr_hip_ctx* side = nullptr;
  if (run && !striped && !getenv("UHDR_HIP_NO_CHAIN_OVERLAP")) {
    uhdr_hip_ctx* x = nullptr;
    if (aux_context(c, &x).error_code == UHDR_CODEC_OK && x) {
      hipError_t e = hipSuccess;
      if (!c->aux_ev) e = hipEventCreateWithFlags(&c->aux_ev, hipEventDisableTiming);
      if (e == hipSuccess && !c->aux_ev2) e = hipEventCreateWithFlags(&c->aux_ev2, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventRecord(c->aux_ev, c->stream);  // the intents (and whatever produced them on this stream) are ready
      if (e == hipSuccess) e = hipStreamWaitEvent(x->stream, c->aux_ev, 0);
      if (e == hipSuccess) {
        {
          ProfScope ps(x, "fdct_quant");
          note_hip(launch_base_blocks(view_of(sdr), convert ? &conv : nullptr, qt_base[0], qt_base[1], blocks->base_coef, x->stream), "base blocks");
        }
        note_hip(hipEventRecord(c->aux_ev2, x->stream), "base blocks event");
        side = x;
        if (c->on_side_launched) c->on_side_launched();
      }
    }
  }
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
      if (c->defer_md) {  // the range is final here: its read-back goes to a side stream, behind this event
        if (!c->md_stream) note_hip(hipStreamCreateWithFlags(&c->md_stream, hipStreamNonBlocking), "metadata stream");
        if (!c->md_ev) note_hip(hipEventCreateWithFlags(&c->md_ev, hipEventDisableTiming), "metadata event");
        if (c->md_stream && c->md_ev) {
          note_hip(hipEventRecord(c->md_ev, c->stream), "metadata event record");
          note_hip(hipStreamWaitEvent(c->md_stream, c->md_ev, 0), "metadata stream wait");
          note_hip(hipMemcpyAsync(c->h_mm, final_mm, 9 * sizeof(float), hipMemcpyDeviceToHost, c->md_stream), "metadata copy");
          md_on_side = true;
        }
      }
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
    if (!side) note_hip(launch_base_blocks(view_of(sdr), convert ? &conv : nullptr, qt_base[0], qt_base[1], blocks->base_coef, c->stream), "base blocks");
  }
  if (side) {  // the base image's coefficients belong to this stream's order from here on (whatever happened above)
    note_hip(hipStreamWaitEvent(c->stream, c->aux_ev2, 0), "base blocks wait");
    if (!c->side_job_posted) aux_merge(c);  // (else the auxiliary context is the worker thread's until the caller has waited for its job)
  }
  chain.reset();
  if (!md_on_side) note_hip(hipMemcpyAsync(c->h_mm, final_mm, 9 * sizeof(float), hipMemcpyDeviceToHost, c->stream), "metadata copy");
  if (c->defer_md && !striped) {  // the caller synchronises later and finishes the metadata then (finish_deferred_md)
    if (local.error_code != UHDR_CODEC_OK) return local;
    c->deferred_md.valid = true;
    c->deferred_md.run = run;
    c->deferred_md.cfg = *cfg;
    c->deferred_md.hdr_ct = hdr->ct;
    c->deferred_md.use_base_cg = use_base_cg;
    return ok_status();
  }
  note_hip(hipStreamSynchronize(c->stream), "synchronize");  // the only host synchronisation: the metadata needs the (merged) range
  if (xchg.error_code != UHDR_CODEC_OK) return xchg;
  if (local.error_code != UHDR_CODEC_OK) return local;
  float mm[6];
  memcpy(mm, c->h_mm, sizeof mm);
  if (run) note_table_stats(c, cfg);
  return generate_gainmap_finalize_md(cfg, hdr->ct, use_base_cg, mm, md);
}

// JpegR::encodeJPEGR API-1 (jpegr.cpp:253-316) from its two raw intents to its two entropy-coded scans in ONE entry point: the
// intents go up once (fast_h2d), the fused chain leaves coefficient blocks in HBM, the marker-less Huffman coder turns them into
// the reference's bytes, and only those come down.  What the facade's seam at encodeJPEGR calls.
static uhdr_error_info_t encode_api1_scans_impl(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                                uhdr_color_gamut_t base_encoding, const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                                uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc, uint8_t* base_scan, size_t base_capacity,
                                                size_t* base_bytes, uint8_t* map_scan, size_t map_capacity, size_t* map_bytes, bool dev = false);
// ... and on DEVICE-resident intents, into DEVICE buffers (round 6): one call per direction of the round trip, so that no binding
// layer sits between the stages (a Python caller's 30-100 us between two entry points were a third of a 4K round trip)
uhdr_error_info_t uhdr_hip_encode_api1_scans_dev(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                                 uhdr_color_gamut_t base_encoding, const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                                 uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc, uint8_t* base_scan, size_t base_capacity,
                                                 size_t* base_bytes, uint8_t* map_scan, size_t map_capacity, size_t* map_bytes) {
  return encode_api1_scans_impl(c, sdr, hdr, cfg, base_encoding, qt_base, qt_map, md, gainmap_desc, base_scan, base_capacity, base_bytes, map_scan, map_capacity,
                                map_bytes, true);
}
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
                                                size_t* base_bytes, uint8_t* map_scan, size_t map_capacity, size_t* map_bytes, bool dev) {
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
  if (dev) {
    ds = *sdr;
    dh = *hdr;
  } else {
    if (c->resident_on) UHDR_TRY(resident_write_back_all(c));
    UHDR_TRY(stage_in(c, 0, sdr, &ds, true));
    UHDR_TRY(stage_in(c, 1, hdr, &dh, true));
  }
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
  // the two scans
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
  size_t nbs = 0, nms = 0;
  // Round 6, device-resident callers: the base image's scan is coded on the auxiliary stream, straight behind the base image's blocks there -- the
  // worker thread gets the job the moment that launch is out, while this thread is still enqueuing the gain-map passes -- and the map's scan on
  // this stream behind the map's blocks.  (Both scans used to wait, by event, for the WHOLE chain: the base image's 150 us of entropy coding
  // started 80 us later than its input was ready, and ended the encode.)
  uhdr_error_info_t rb = ok_status();
  static const bool late_base = getenv("UHDR_HIP_NO_EARLY_BASE_SCAN") != nullptr;
  if (dev && !late_base) {
    const int devno = c->device;
    c->on_side_launched = [&, devno] {
      uhdr_hip_ctx* x = c->aux;
      c->side_job_posted = aux_post(c, [&, x, devno] {
        (void)hipSetDevice(devno);
        rb = uhdr_hip_huffman_encode_dev(x, &sb, base_scan, base_capacity, &nbs);
      });
    };
  }
  c->side_job_posted = false;
  c->defer_md = dev;  // device-resident callers: no host synchronisation between the chain and the entropy stage
  c->deferred_md.valid = false;
  const uhdr_error_info_t fs = uhdr_hip_encode_api1_fused_dev(c, &ds, &dh, cfg, base_encoding, qt_base, qt_map, &blocks, md, nullptr);
  c->defer_md = false;
  c->on_side_launched = nullptr;
  const bool base_posted = c->side_job_posted;
  c->side_job_posted = false;
  if (fs.error_code != UHDR_CODEC_OK) {
    if (base_posted) { aux_wait(c); aux_merge(c); }
    return fs;
  }
  auto finish_deferred_md = [&]() -> uhdr_error_info_t {  // (after a synchronisation of c->stream)
    if (!c->deferred_md.valid) return ok_status();
    if (c->md_stream) HIP_TRY(hipStreamSynchronize(c->md_stream));  // (the range's copy left long ago: behind the range kernel)
    c->deferred_md.valid = false;
    float mm[6];
    memcpy(mm, c->h_mm, sizeof mm);
    if (c->deferred_md.run) note_table_stats(c, &c->deferred_md.cfg);
    return generate_gainmap_finalize_md(&c->deferred_md.cfg, c->deferred_md.hdr_ct, c->deferred_md.use_base_cg, mm, md);
  };
  if (gainmap_desc) {  // what generateGainMap's freshly allocated image would say (jpegr.cpp:714-716); planes untouched
    gainmap_desc->fmt = nch == 3 ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_8bppYCbCr400;
    gainmap_desc->cg = hdr->cg; gainmap_desc->ct = hdr->ct; gainmap_desc->range = hdr->range;
    gainmap_desc->w = mw; gainmap_desc->h = mh;
  }
  if (dev) {  // straight into the caller's device buffers
    uhdr_error_info_t e2;
    if (base_posted) {
      const uhdr_error_info_t ra = uhdr_hip_huffman_encode_dev(c, &sm, map_scan, map_capacity, &nms);
      aux_wait(c);
      aux_merge(c);
      e2 = rb.error_code != UHDR_CODEC_OK ? rb : ra;
    } else {
      e2 = uhdr_hip_huffman_encode2_dev(c, &sb, base_scan, base_capacity, &nbs, &sm, map_scan, map_capacity, &nms);
    }
    *base_bytes = nbs;
    *map_bytes = nms;
    if (e2.error_code != UHDR_CODEC_OK) {
      (void)hipStreamSynchronize(c->stream);
      c->deferred_md.valid = false;
      return e2;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));  // (the entropy stage has synchronised already: the metadata's copy is long done)
    return finish_deferred_md();
  }
  UHDR_TRY(ensure(c->enc[1], base_capacity + 64));
  UHDR_TRY(ensure(c->enc[2], map_capacity + 64));
  // both scans at once (round 5): the base image on this context's stream, the map on the auxiliary one
  const uhdr_error_info_t e2 = uhdr_hip_huffman_encode2_dev(c, &sb, (uint8_t*)c->enc[1].p, base_capacity, &nbs, &sm, (uint8_t*)c->enc[2].p, map_capacity, &nms);
  *base_bytes = nbs;
  *map_bytes = nms;
  if (e2.error_code != UHDR_CODEC_OK) return e2;
  HIP_TRY(hipMemcpyAsync(base_scan, c->enc[1].p, nbs, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(map_scan, c->enc[2].p, nms, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
}

// JpegR::encodeJPEGR API-0 (jpegr.cpp:179-244) for the HDR intents its tone map renders to RGBA8888 -- RGBA1010102 (BASELINE config 3) and
// RGBA half float -- from the raw intent to the two entropy-coded scans in ONE entry point (round 6; what the facade's seam at encodeJPEGR
// API-0 calls): the intent goes up once, uhdr_hip_encode_api0_fused_dev leaves the YCbCr 4:4:4 base image and the one-pass gain map in HBM,
// FDCT + quantize (rgb -> ycc inside for a three-channel map), both scans Huffman-coded concurrently, and only the bytes come down.
// The reference overrides the preset to UHDR_USAGE_REALTIME on this path (jpegr.cpp:207) and passes use_luminance = false: so does this.
uhdr_error_info_t uhdr_hip_encode_api0_scans(uhdr_hip_ctx_t* c, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg_in, const uint16_t qt_base[2][64],
                                             const uint16_t qt_map[2][64], uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc,
                                             uhdr_color_gamut_t* sdr_cg, uint8_t* base_scan, size_t base_capacity, size_t* base_bytes, uint8_t* map_scan,
                                             size_t map_capacity, size_t* map_bytes) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!hdr || !cfg_in || !qt_base || !qt_map || !md || !base_scan || !map_scan || !base_bytes || !map_bytes)
    return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument");
  if (hdr->fmt != UHDR_IMG_FMT_32bppRGBA1010102 && hdr->fmt != UHDR_IMG_FMT_64bppRGBAHalfFloat)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-0 chain takes UHDR_IMG_FMT_32bppRGBA1010102 or UHDR_IMG_FMT_64bppRGBAHalfFloat. Received %d", hdr->fmt);
  if (cfg_in->map_dimension_scale_factor != 1)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-0 chain needs a full-resolution gain map (scale factor 1), received %d", cfg_in->map_dimension_scale_factor);
  if (hdr->w == 0 || hdr->h == 0 || hdr->w % 8 || hdr->h % 8 || hdr->w > 65535 || hdr->h > 65535)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "the fused API-0 chain needs dimensions that are multiples of 8 (received %ux%u); use the operators", hdr->w, hdr->h);
  if (c->comm != nullptr || c->comm_custom) return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "a context with a communicator encodes stripes");
  UHDR_TRY(validate_image(hdr, "hdr intent"));
  for (int t = 0; t < 2; t++)
    for (int i = 0; i < 64; i++)
      if (qt_base[t][i] == 0 || qt_base[t][i] > 255 || qt_map[t][i] == 0 || qt_map[t][i] > 255)
        return err_status(UHDR_CODEC_INVALID_PARAM, "quantization table entry %d out of baseline range", i);
  HIP_TRY(hipSetDevice(c->device));
  uhdr_hip_encode_cfg_t cfg = *cfg_in;
  cfg.preset = UHDR_USAGE_REALTIME;  // jpegr.cpp:207
  cfg.sdr_is_601 = 0;                // jpegr.cpp:213-214
  cfg.use_luminance = 0;
  const unsigned w = hdr->w, h = hdr->h;
  const int nch = cfg.use_multi_channel_gainmap ? 3 : 1;
  uhdr_raw_image_t dh;
  if (c->resident_on) UHDR_TRY(resident_write_back_all(c));
  UHDR_TRY(stage_in(c, 1, hdr, &dh, true));
  // device images: the base image's three 4:4:4 planes and the map (rows padded to 64 samples), then the coefficient arrays
  const size_t pitch = ((size_t)w + 63) & ~(size_t)63, plane = pitch * h, map_pitch_px = pitch;
  UHDR_TRY(ensure(c->enc[1], 3 * plane + 256));
  UHDR_TRY(ensure(c->enc[2], map_pitch_px * (size_t)nch * h + 256));
  const size_t nblk = (size_t)(w / 8) * (h / 8), cbytes = (nblk * 128 + 255) & ~(size_t)255;
  UHDR_TRY(ensure(c->enc[0], cbytes * (size_t)(3 + nch)));
  uhdr_raw_image_t ycc, gm;
  memset(&ycc, 0, sizeof ycc);
  memset(&gm, 0, sizeof gm);
  for (int i = 0; i < 3; i++) {
    ycc.planes[i] = (uint8_t*)c->enc[1].p + (size_t)i * plane;
    ycc.stride[i] = (unsigned int)pitch;
  }
  gm.planes[0] = c->enc[2].p;
  gm.stride[0] = (unsigned int)map_pitch_px;
  UHDR_TRY(uhdr_hip_encode_api0_fused_dev(c, &dh, &cfg, nullptr, &ycc, md, &gm));
  if (sdr_cg) *sdr_cg = ycc.cg;  // what toneMap gives its SDR rendition (jpegr.cpp:2024-2030) and convert_raw_input_to_ycbcr keeps
  if (gainmap_desc) {
    *gainmap_desc = gm;
    gainmap_desc->planes[0] = gainmap_desc->planes[1] = gainmap_desc->planes[2] = nullptr;
  }
  int16_t* coef[6];
  for (int i = 0; i < 3 + nch; i++) coef[i] = (int16_t*)((uint8_t*)c->enc[0].p + (size_t)i * cbytes);
  for (int i = 0; i < 3; i++)
    UHDR_TRY(uhdr_hip_fdct_quant_dev(c, (const uint8_t*)ycc.planes[i], pitch, (int)(w / 8), (int)(h / 8), qt_base[i ? 1 : 0], coef[i]));
  if (nch == 3) UHDR_TRY(uhdr_hip_fdct_quant_rgb_dev(c, &gm, qt_map[0], qt_map[1], coef[3], coef[4], coef[5]));
  else UHDR_TRY(uhdr_hip_fdct_quant_dev(c, (const uint8_t*)gm.planes[0], (size_t)gm.stride[0], (int)(w / 8), (int)(h / 8), qt_map[0], coef[3]));
  uhdr_hip_jpeg_scan_t sb, sm;
  memset(&sb, 0, sizeof sb);
  memset(&sm, 0, sizeof sm);
  sb.num_components = 3;
  sm.num_components = nch;
  sb.w = sm.w = w;
  sb.h = sm.h = h;
  for (int i = 0; i < 3; i++) {
    sb.coef[i] = coef[i];
    sb.blocks_w[i] = (int)(w / 8); sb.blocks_h[i] = (int)(h / 8);
    sb.h_samp[i] = sb.v_samp[i] = 1;
  }
  for (int i = 0; i < nch; i++) {
    sm.coef[i] = coef[3 + i];
    sm.blocks_w[i] = (int)(w / 8); sm.blocks_h[i] = (int)(h / 8);
    sm.h_samp[i] = sm.v_samp[i] = 1;
  }
  if (base_capacity > 0xFFFFFFF0u) base_capacity = 0xFFFFFFF0u;
  if (map_capacity > 0xFFFFFFF0u) map_capacity = 0xFFFFFFF0u;
  // the scans' device buffers: the image planes are free again once the FDCTs have run -- but those are only enqueued, so separate ones
  UHDR_TRY(ensure(c->jpg[0], base_capacity + 64));
  UHDR_TRY(ensure(c->jpg[4], map_capacity + 64));
  size_t nbs = 0, nms = 0;
  const uhdr_error_info_t e2 = uhdr_hip_huffman_encode2_dev(c, &sb, (uint8_t*)c->jpg[0].p, base_capacity, &nbs, &sm, (uint8_t*)c->jpg[4].p, map_capacity, &nms);
  *base_bytes = nbs;
  *map_bytes = nms;
  if (e2.error_code != UHDR_CODEC_OK) return e2;
  HIP_TRY(hipMemcpyAsync(base_scan, c->jpg[0].p, nbs, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(map_scan, c->jpg[4].p, nms, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return ok_status();
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
  UHDR_TRY(fast_h2d(c, c->jpg[0].p, scan_data, nbytes));  // through the pinned ring: the runtime never stages the caller's (fresh, pageable) copy of the file
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

// JpegR::decodeJPEGR behind its container parsing (jpegr.cpp:1469-1531) on DEVICE-resident data in ONE entry point (round 6): the two
// JPEG streams' parsed headers + their entropy-coded bytes in HBM -> [both scans entropy-decoded concurrently] -> the gain map's
// dequant + IDCT (+ ycc -> rgb for a three-channel map) -> applyGainMap with the base image's dequant + IDCT inside the kernel -> dest.
// Only enqueues the two sample-domain launches after the (synchronous) entropy stage: dest is ready in stream order.
uhdr_error_info_t uhdr_hip_decode_api1_scans_dev(uhdr_hip_ctx_t* c, const uhdr_hip_jpeg_header_t* base, const uint8_t* base_data, size_t base_bytes,
                                                 uhdr_color_gamut_t base_cg, const uhdr_hip_jpeg_header_t* map, const uint8_t* map_data, size_t map_bytes,
                                                 uhdr_color_gamut_t map_cg, int libjpeg_variant, const uhdr_gainmap_metadata_t* md,
                                                 uhdr_color_transfer_t output_ct, uhdr_img_fmt_t output_format, float max_display_boost,
                                                 uhdr_raw_image_t* dest) {
  if (!c) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr for uhdr_hip context");
  if (!base || !base_data || !map || !map_data || !md || !dest) return err_status(UHDR_CODEC_INVALID_PARAM, "received nullptr argument for decode_api1_scans");
  if (libjpeg_variant != 0 && libjpeg_variant != 1) return err_status(UHDR_CODEC_INVALID_PARAM, "unknown libjpeg variant %d", libjpeg_variant);
  uhdr_hip_jpeg_scan_t sb = base->scan, sm = map->scan;
  int mpr = 0, mrows = 0, bpm = 0;
  UHDR_TRY(check_scan(&sb, false, &mpr, &mrows, &bpm));
  if (sb.num_components != 3 || sb.h_samp[0] != 2 || sb.v_samp[0] != 2 || sb.h_samp[1] != 1 || sb.v_samp[1] != 1 || sb.h_samp[2] != 1 || sb.v_samp[2] != 1)
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "decode_api1_scans takes a 4:2:0 base image (the form JpegR writes); decode the scans with uhdr_hip_jpeg_decode_scan");
  UHDR_TRY(check_scan(&sm, false, &mpr, &mrows, &bpm));
  const int nm = sm.num_components;
  if (nm == 3 && (bpm != 3 || memcmp(map->qtable[1], map->qtable[2], sizeof map->qtable[1])))
    return err_status(UHDR_CODEC_UNSUPPORTED_FEATURE, "a three-channel gain map is a 4:4:4 scan with one chroma quantization table here");
  if (base_bytes == 0 || map_bytes == 0 || base_bytes > 0xFFFFFFF0ull || map_bytes > 0xFFFFFFF0ull) return err_status(UHDR_CODEC_INVALID_PARAM, "received no (or too much) entropy-coded data");
  HIP_TRY(hipSetDevice(c->device));
  // coefficient arrays: the base image's in jpg[1..3], the map's behind one another in enc[0]
  for (int i = 0; i < 3; i++) {
    UHDR_TRY(ensure(c->jpg[1 + i], (size_t)sb.blocks_w[i] * sb.blocks_h[i] * 128));
    sb.coef[i] = (const int16_t*)c->jpg[1 + i].p;
  }
  size_t off = 0, o_map[3] = {0, 0, 0};
  for (int i = 0; i < nm; i++) { o_map[i] = off; off += ((size_t)sm.blocks_w[i] * sm.blocks_h[i] * 128 + 255) & ~(size_t)255; }
  UHDR_TRY(ensure(c->enc[0], off));
  for (int i = 0; i < nm; i++) sm.coef[i] = (const int16_t*)((const uint8_t*)c->enc[0].p + o_map[i]);
  c->huff_serial_ok = false;  // a scan the parallel decoder declines goes back to the caller (UNSUPPORTED_FEATURE), not to one lane
  auto tables_of = [](const uhdr_hip_jpeg_header_t* h) -> const uhdr_hip_huff_tables_t* {  // an all-zero DHT block: the Annex K tables
    for (int t = 0; t < 4; t++)
      for (int l = 0; l < 17; l++)
        if (h->tables.bits[t][l]) return &h->tables;
    return nullptr;
  };
  const uhdr_error_info_t hs = uhdr_hip_huffman_decode2_dev(c, &sb, tables_of(base), base_data, base_bytes, &sm, tables_of(map), map_data, map_bytes);
  c->huff_serial_ok = true;
  if (hs.error_code != UHDR_CODEC_OK) return hs;
  // the decoded gain-map image, device resident
  uhdr_raw_image_t gm;
  memset(&gm, 0, sizeof gm);
  gm.cg = map_cg;
  gm.ct = UHDR_CT_UNSPECIFIED;
  gm.range = UHDR_CR_FULL_RANGE;
  gm.w = sm.w;
  gm.h = sm.h;
  const size_t pitch_px = ((size_t)sm.w + 63) & ~(size_t)63;
  if (nm == 3) {
    gm.fmt = UHDR_IMG_FMT_32bppRGBA8888;
    UHDR_TRY(ensure(c->jpg[4], pitch_px * 4 * sm.h));
    gm.planes[0] = c->jpg[4].p;
    gm.stride[0] = (unsigned int)pitch_px;
    UHDR_TRY(uhdr_hip_idct_dequant_rgb_dev(c, sm.coef[0], sm.coef[1], sm.coef[2], sm.blocks_w[0], sm.blocks_h[0], map->qtable[0], map->qtable[1], libjpeg_variant, &gm));
  } else {
    gm.fmt = UHDR_IMG_FMT_8bppYCbCr400;
    const size_t pitch = ((size_t)sm.blocks_w[0] * 8 + 63) & ~(size_t)63;
    UHDR_TRY(ensure(c->jpg[4], pitch * (size_t)sm.blocks_h[0] * 8));
    gm.planes[0] = c->jpg[4].p;
    gm.stride[0] = (unsigned int)pitch;
    UHDR_TRY(uhdr_hip_idct_dequant_dev(c, sm.coef[0], sm.blocks_w[0], sm.blocks_h[0], map->qtable[0], (uint8_t*)c->jpg[4].p, pitch));
  }
  uhdr_hip_jpeg_coefficients_t bc;
  memset(&bc, 0, sizeof bc);
  for (int i = 0; i < 3; i++) {
    bc.coef[i] = sb.coef[i];
    bc.blocks_w[i] = sb.blocks_w[i];
    bc.blocks_h[i] = sb.blocks_h[i];
    memcpy(bc.qtable[i], base->qtable[i], sizeof bc.qtable[i]);
  }
  uhdr_error_info_t ap = uhdr_hip_apply_gainmap_coef_dev(c, &bc, sb.w, sb.h, base_cg, &gm, md, output_ct, output_format, max_display_boost, dest);
  if (ap.error_code != UHDR_CODEC_UNSUPPORTED_FEATURE) return ap;
  // a geometry the coefficient-input kernel does not take: the planes after all (three IDCT launches), then the operator
  size_t ppitch[3], poff[3], total = 0;
  for (int i = 0; i < 3; i++) {
    ppitch[i] = ((size_t)sb.blocks_w[i] * 8 + 63) & ~(size_t)63;
    poff[i] = total;
    total += ppitch[i] * (size_t)sb.blocks_h[i] * 8;
  }
  UHDR_TRY(ensure(c->jpg[0], total));
  uhdr_raw_image_t bi;
  memset(&bi, 0, sizeof bi);
  bi.fmt = UHDR_IMG_FMT_12bppYCbCr420;
  bi.cg = base_cg;
  bi.ct = UHDR_CT_SRGB;
  bi.range = UHDR_CR_FULL_RANGE;
  bi.w = sb.w;
  bi.h = sb.h;
  for (int i = 0; i < 3; i++) {
    uint8_t* d = (uint8_t*)c->jpg[0].p + poff[i];
    UHDR_TRY(uhdr_hip_idct_dequant_dev(c, sb.coef[i], sb.blocks_w[i], sb.blocks_h[i], base->qtable[i], d, ppitch[i]));
    bi.planes[i] = d;
    bi.stride[i] = (unsigned int)ppitch[i];
  }
  return uhdr_hip_apply_gainmap_dev(c, &bi, &gm, md, output_ct, output_format, max_display_boost, dest, 0, 0);
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
