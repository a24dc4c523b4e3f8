/*
 * uhdr_hip.h -- C ABI of libuhdr_hip.so: the MI355X (gfx950) implementation of libultrahdr's
 * per-pixel gain-map hot path.  Plain C, plain pointers and sizes; no torch / HIP types leak
 * through this boundary (streams travel as void*).
 *
 * Each entry point is the drop-in for one reference operator (paths under /root/reference):
 *
 *   uhdr_hip_apply_gainmap              UltraHdr::applyGainMap        lib/include/ultrahdr/ultrahdrcommon.h:531-534
 *                                                                     (impl lib/src/jpegr.cpp:1533-1831; replaces the GLES
 *                                                                      dispatch seam at jpegr.cpp:1633-1649)
 *   uhdr_hip_generate_gainmap           UltraHdr::generateGainMap     ultrahdrcommon.h:507-510 (impl jpegr.cpp:530-1058)
 *   uhdr_hip_tone_map                   UltraHdr::toneMap             ultrahdrcommon.h:482      (impl jpegr.cpp:1985-2222)
 *   uhdr_hip_convert_yuv                UltraHdr::convertYuv          ultrahdrcommon.h:545-546 (impl jpegr.cpp:436-518)
 *   uhdr_hip_convert_raw_input_to_ycbcr convert_raw_input_to_ycbcr    lib/include/ultrahdr/gainmapmath.h:604-605
 *                                                                     (impl lib/src/gainmapmath.cpp:1291-1482)
 *   uhdr_hip_fdct_quant                 the FDCT+quantize stage libjpeg runs inside
 *                                       JpegEncoderHelper::compressImage  lib/include/ultrahdr/jpegencoderhelper.h:57-58
 *                                                                     (call sites lib/src/jpegencoderhelper.cpp:187-198,297)
 *   uhdr_hip_copy_raw_image_dev         copy_raw_image                lib/src/gainmapmath.cpp:1492-1613
 *   uhdr_hip_jpeg_rgb_to_ycc            libjpeg's JCS_RGB -> YCbCr for 3-channel gain maps  lib/src/jpegencoderhelper.cpp:165-167, 212-225
 *   uhdr_hip_idct_dequant,
 *   uhdr_hip_jpeg_ycc_to_rgb            the dequantize + IDCT (+ colour conversion) stage libjpeg runs inside
 *                                       JpegDecoderHelper::decompressImage  lib/src/jpegdecoderhelper.cpp:169-535
 *   MI355X extensions without a single reference counterpart (each documented at its declaration):
 *   uhdr_hip_apply_gainmap_batch_dev (n frames, one launch), uhdr_hip_generate_gainmap_pass1_dev / _finalize /
 *   _pass2_dev (two-pass generation split at its only exchange step, for row stripes across GPUs),
 *   uhdr_hip_encode_api0_fused_dev (toneMap + generateGainMap + convert_raw_input_to_ycbcr in one pass),
 *   uhdr_hip_fdct_quant_rgb_dev (colour conversion + FDCT of a 3-channel map in one pass),
 *   uhdr_hip_idct_dequant_rgb_dev (its decode-side mirror: dequant + IDCT + colour conversion in one pass),
 *   uhdr_hip_apply_gainmap_coef_dev (applyGainMap on a base image still in coefficient form: IDCT inside the kernel),
 *   uhdr_hip_huffman_encode_dev + uhdr_hip_jpeg_assemble (baseline Huffman entropy coding, without restart markers or one
 *   restart interval per wavefront, and the file wrapper around it), uhdr_hip_huffman_decode_dev (its inverse: the
 *   self-synchronising parallel decoder, or one interval per lane), uhdr_hip_jpeg_parse (host: the headers of a
 *   baseline JPEG file, in the form those entry points take)
 *
 * Same argument meaning and error behaviour as the reference: uhdr_error_info_t is returned by
 * value, UHDR_CODEC_OK == 0, strides are in PIXELS, outputs go into caller-provided images.
 * Host-memory variants are synchronous and stage through the context's device buffers; *_dev variants take DEVICE plane pointers (data already in HBM) and
 * only enqueue work on the context's stream.
 */
#ifndef UHDR_HIP_H
#define UHDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- types shared with the reference's public header ------------------------------------------
 * When ultrahdr_api.h (the reference's public header) is already included, its definitions are
 * used as they are; otherwise layout-identical ones are declared here (ultrahdr_api.h:108-283). */
#ifndef ULTRAHDR_API_H
typedef enum uhdr_img_fmt {
  UHDR_IMG_FMT_UNSPECIFIED = -1,
  UHDR_IMG_FMT_24bppYCbCrP010 = 0,
  UHDR_IMG_FMT_12bppYCbCr420 = 1,
  UHDR_IMG_FMT_8bppYCbCr400 = 2,
  UHDR_IMG_FMT_32bppRGBA8888 = 3,
  UHDR_IMG_FMT_64bppRGBAHalfFloat = 4,
  UHDR_IMG_FMT_32bppRGBA1010102 = 5,
  UHDR_IMG_FMT_24bppYCbCr444 = 6,
  UHDR_IMG_FMT_16bppYCbCr422 = 7,
  UHDR_IMG_FMT_16bppYCbCr440 = 8,
  UHDR_IMG_FMT_12bppYCbCr411 = 9,
  UHDR_IMG_FMT_10bppYCbCr410 = 10,
  UHDR_IMG_FMT_24bppRGB888 = 11,
  UHDR_IMG_FMT_30bppYCbCr444 = 12
} uhdr_img_fmt_t;
typedef enum uhdr_color_gamut {
  UHDR_CG_UNSPECIFIED = -1, UHDR_CG_BT_709 = 0, UHDR_CG_DISPLAY_P3 = 1, UHDR_CG_BT_2100 = 2
} uhdr_color_gamut_t;
typedef enum uhdr_color_transfer {
  UHDR_CT_UNSPECIFIED = -1, UHDR_CT_LINEAR = 0, UHDR_CT_HLG = 1, UHDR_CT_PQ = 2, UHDR_CT_SRGB = 3
} uhdr_color_transfer_t;
typedef enum uhdr_color_range {
  UHDR_CR_UNSPECIFIED = -1, UHDR_CR_LIMITED_RANGE = 0, UHDR_CR_FULL_RANGE = 1
} uhdr_color_range_t;
typedef enum uhdr_enc_preset { UHDR_USAGE_REALTIME, UHDR_USAGE_BEST_QUALITY } uhdr_enc_preset_t;
typedef enum uhdr_codec_err {
  UHDR_CODEC_OK,
  UHDR_CODEC_ERROR,
  UHDR_CODEC_UNKNOWN_ERROR,
  UHDR_CODEC_INVALID_PARAM,
  UHDR_CODEC_MEM_ERROR,
  UHDR_CODEC_INVALID_OPERATION,
  UHDR_CODEC_UNSUPPORTED_FEATURE,
  UHDR_CODEC_LIST_END
} uhdr_codec_err_t;
typedef struct uhdr_error_info {
  uhdr_codec_err_t error_code;
  int has_detail;
  char detail[256];
} uhdr_error_info_t;
#define UHDR_PLANE_PACKED 0
#define UHDR_PLANE_Y 0
#define UHDR_PLANE_U 1
#define UHDR_PLANE_UV 1
#define UHDR_PLANE_V 2
typedef struct uhdr_raw_image {
  uhdr_img_fmt_t fmt;
  uhdr_color_gamut_t cg;
  uhdr_color_transfer_t ct;
  uhdr_color_range_t range;
  unsigned int w, h;
  void* planes[3];
  unsigned int stride[3]; /* pixels */
} uhdr_raw_image_t;
typedef struct uhdr_gainmap_metadata {
  float max_content_boost[3];
  float min_content_boost[3];
  float gamma[3];
  float offset_sdr[3];
  float offset_hdr[3];
  float hdr_capacity_min;
  float hdr_capacity_max;
  int use_base_cg;
} uhdr_gainmap_metadata_t;
#endif /* ULTRAHDR_API_H */

/* Encoder knobs: the UltraHdr constructor arguments that reach the hot path plus
 * generateGainMap's two optional flags (ultrahdrcommon.h:450-457, 507-510).  "unset" sentinels
 * are the reference's own: FLT_MIN / FLT_MAX / -1.0f. */
typedef struct uhdr_hip_encode_cfg {
  int map_dimension_scale_factor; /* mMapDimensionScaleFactor        (C-API default 1, Android 4) */
  int use_multi_channel_gainmap;  /* mUseMultiChannelGainMap         (C-API default 1)            */
  float gamma;                    /* mGamma                          (default 1.0)                */
  int preset;                     /* uhdr_enc_preset_t: REALTIME = one pass, BEST_QUALITY = two   */
  float min_content_boost;        /* mMinContentBoost, FLT_MIN = unset                            */
  float max_content_boost;        /* mMaxContentBoost, FLT_MAX = unset                            */
  float target_disp_peak_nits;    /* mTargetDispPeakBrightness, -1 = unset                        */
  int sdr_is_601;
  int use_luminance;
} uhdr_hip_encode_cfg_t;

typedef struct uhdr_hip_ctx uhdr_hip_ctx_t; /* one per (device, stream); not thread-safe, like a codec handle */

/* ---- context ------------------------------------------------------------------------------- */
/* device < 0: current device.  Fails (NULL + message in *err, may be NULL) when no gfx950-class
 * GPU is usable: there is no CPU fallback behind this ABI. */
uhdr_hip_ctx_t* uhdr_hip_create(int device, uhdr_error_info_t* err);
void uhdr_hip_destroy(uhdr_hip_ctx_t* ctx);
/* run on a caller-owned hipStream_t (e.g. torch's current stream); NULL = the context's own */
uhdr_error_info_t uhdr_hip_set_stream(uhdr_hip_ctx_t* ctx, void* hip_stream);
/* the hipStream_t the context currently enqueues on (its own non-blocking stream unless uhdr_hip_set_stream changed
 * it): lets a caller order its own streams against the library's with events instead of host synchronisation */
void* uhdr_hip_get_stream(uhdr_hip_ctx_t* ctx);
uhdr_error_info_t uhdr_hip_synchronize(uhdr_hip_ctx_t* ctx);
const char* uhdr_hip_version(void);
int uhdr_hip_device_count(void);

/* ---- stage operators, HOST buffers (drop-in for the UltraHdr:: methods) ----------------------- */
uhdr_error_info_t uhdr_hip_apply_gainmap(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* sdr_intent,
                                         const uhdr_raw_image_t* gainmap_img,
                                         const uhdr_gainmap_metadata_t* gainmap_metadata,
                                         uhdr_color_transfer_t output_ct,
                                         uhdr_img_fmt_t output_format, float max_display_boost,
                                         uhdr_raw_image_t* dest);
/* gainmap_img: caller provides planes[0] and stride[0] (>= w/scale; the reference aligns to 64);
 * fmt / w / h / colour aspects are filled in like the reference's freshly allocated image. */
uhdr_error_info_t uhdr_hip_generate_gainmap(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* sdr_intent,
                                            const uhdr_raw_image_t* hdr_intent,
                                            const uhdr_hip_encode_cfg_t* cfg,
                                            uhdr_gainmap_metadata_t* gainmap_metadata,
                                            uhdr_raw_image_t* gainmap_img);
uhdr_error_info_t uhdr_hip_tone_map(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* hdr_intent,
                                    uhdr_raw_image_t* sdr_intent);
uhdr_error_info_t uhdr_hip_convert_yuv(uhdr_hip_ctx_t* ctx, uhdr_raw_image_t* image,
                                       uhdr_color_gamut_t src_encoding,
                                       uhdr_color_gamut_t dst_encoding);
/* dst: caller provides planes/strides; dst->fmt is set to what the reference would allocate */
uhdr_error_info_t uhdr_hip_convert_raw_input_to_ycbcr(uhdr_hip_ctx_t* ctx,
                                                      const uhdr_raw_image_t* src,
                                                      int chroma_sampling_enabled,
                                                      uhdr_raw_image_t* dst);

/* ---- stage operators, DEVICE buffers (planes[] are device pointers) -------------------------- */
/* Row-stripe sharding: sdr/dest may describe only this rank's rows of a taller image; the gain
 * map is whole and replicated.  y0 = global row of the stripe's first row, full_height = height
 * of the whole image.  Whole image: y0 = 0, full_height = 0. */
uhdr_error_info_t uhdr_hip_apply_gainmap_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* sdr_intent,
                                             const uhdr_raw_image_t* gainmap_img,
                                             const uhdr_gainmap_metadata_t* gainmap_metadata,
                                             uhdr_color_transfer_t output_ct,
                                             uhdr_img_fmt_t output_format, float max_display_boost,
                                             uhdr_raw_image_t* dest, unsigned int y0,
                                             unsigned int full_height);
/* Batch decode (BASELINE config 5): n frames of identical geometry / formats / colour aspects that
 * share one metadata block (a burst or a video-like sequence), arrays of n descriptors with
 * DEVICE plane pointers, ONE kernel launch.  Combinations the batch kernel does not cover run
 * frame by frame with identical results. */
uhdr_error_info_t uhdr_hip_apply_gainmap_batch_dev(uhdr_hip_ctx_t* ctx, unsigned int n,
                                                   const uhdr_raw_image_t* sdr_intents,
                                                   const uhdr_raw_image_t* gainmap_imgs,
                                                   const uhdr_gainmap_metadata_t* gainmap_metadata,
                                                   uhdr_color_transfer_t output_ct,
                                                   uhdr_img_fmt_t output_format, float max_display_boost,
                                                   uhdr_raw_image_t* dests);
/* one-pass (REALTIME) generation, or the whole two-pass sequence on one device */
uhdr_error_info_t uhdr_hip_generate_gainmap_dev(uhdr_hip_ctx_t* ctx,
                                                const uhdr_raw_image_t* sdr_intent,
                                                const uhdr_raw_image_t* hdr_intent,
                                                const uhdr_hip_encode_cfg_t* cfg,
                                                uhdr_gainmap_metadata_t* gainmap_metadata,
                                                uhdr_raw_image_t* gainmap_img);
/* two-pass generation split at its only exchange step (jpegr.cpp:932-938: the min/max merge):
 *   pass1: per map pixel gain -> gain_log2_dev (device floats, w/scale * h/scale * (3|1): an OPAQUE plane that only pass2
 *          reads -- since round 4 it holds the gain ratio (hdr + eps) / (sdr + eps), not its log2: min / max commute with the
 *          monotone log2 and pass 2's byte is a step function of the ratio, see csrc/generate_gainmap.hip),
 *          per-channel min/max of this stripe's LOG2 gains, exactly the reference's six floats
 *          -> minmax_dev[6] = {min0,min1,min2,max0,max1,max2} (device floats; initialise nothing, the call does)
 *   <all-reduce minmax_dev across ranks: MIN on [0..2], MAX on [3..5]>
 *   finalize: clamp / hints / epsilon guard + metadata fill (jpegr.cpp:969-986, 1031-1048), host
 *   pass2: affine map to u8 (jpegr.cpp:992-1013): per-channel step tables built on the device from the final range, then a
 *          table lookup per sample (a user gamma != 1 or a range too narrow for a table: per-sample evaluation) */
uhdr_error_info_t uhdr_hip_generate_gainmap_pass1_dev(uhdr_hip_ctx_t* ctx,
                                                      const uhdr_raw_image_t* sdr_intent,
                                                      const uhdr_raw_image_t* hdr_intent,
                                                      const uhdr_hip_encode_cfg_t* cfg,
                                                      float* gain_log2_dev, float* minmax_dev,
                                                      int* use_base_cg);
uhdr_error_info_t uhdr_hip_generate_gainmap_finalize(const uhdr_hip_encode_cfg_t* cfg,
                                                     uhdr_color_transfer_t hdr_ct, int use_base_cg,
                                                     float minmax[6],
                                                     uhdr_gainmap_metadata_t* gainmap_metadata);
uhdr_error_info_t uhdr_hip_generate_gainmap_pass2_dev(uhdr_hip_ctx_t* ctx,
                                                      const float* gain_log2_dev,
                                                      const float minmax[6],
                                                      const uhdr_hip_encode_cfg_t* cfg,
                                                      uhdr_raw_image_t* gainmap_img);
/* ---- multi-GPU: one process per GPU, images sharded by row stripe -----------------------------
 * Every stage of the path is stripe-local except two-pass generateGainMap's per-channel min / max merge
 * (jpegr.cpp:932-938).  uhdr_hip_generate_gainmap_striped_dev runs the whole two-pass sequence on this rank's stripe:
 *   pass 1 -> ONE ncclAllReduce(min) over {min0..2, -max0..2} (RCCL over xGMI, on the context's stream) ->
 *   finalisation of the range on the device (jpegr.cpp:969-986) -> pass 2
 * as one stream-ordered sequence; the host synchronises once, at the end, to fill the metadata (identical on every rank).
 * sdr / hdr describe this rank's rows (a multiple of lcm(2, scale) rows, except the last stripe); gainmap_img this rank's
 * rows of the map (planes[0] / stride[0] from the caller).  A stripe too short for one map row launches nothing and
 * contributes the merge's identity.  Without a communicator (uhdr_hip_comm_init not called) the same sequence runs
 * for a single stripe = the whole image.  A rank whose arguments are rejected (or whose pass 1 fails to launch) STILL takes
 * part in the exchange, with the identity, and returns its error afterwards: one bad descriptor never leaves the other
 * ranks waiting in the collective.
 * Communicator set-up is the usual NCCL bootstrap: rank 0 calls uhdr_hip_comm_unique_id, the application sends the
 * 128 bytes to the other ranks (torch.distributed, MPI, a file ...), every rank calls uhdr_hip_comm_init.  RCCL is
 * bound at run time (the copy already in the process, else librccl.so.1); the library does not link against it. */
#define UHDR_HIP_COMM_ID_BYTES 128
int uhdr_hip_comm_unique_id(unsigned char id[UHDR_HIP_COMM_ID_BYTES]); /* 0, or -1 when RCCL is unavailable */
uhdr_error_info_t uhdr_hip_comm_init(uhdr_hip_ctx_t* ctx, const unsigned char id[UHDR_HIP_COMM_ID_BYTES], int rank, int nranks);
void uhdr_hip_comm_destroy(uhdr_hip_ctx_t* ctx);
int uhdr_hip_comm_size(uhdr_hip_ctx_t* ctx); /* ncclCommCount of the context's communicator, 0 without one */
int uhdr_hip_comm_rank(uhdr_hip_ctx_t* ctx);
/* The exchange steps over a caller-provided transport instead of RCCL (an MPI / gloo / shared-memory relay where RCCL is not
 * an option -- e.g. several ranks on one GPU -- or a test double).  The library calls the functions in stream order with
 * DEVICE pointers and its own hipStream_t (as void*); they return 0 on success.  all_reduce_min_f32 is required (it is the
 * one collective of the hot path), the gathers are optional (NULL: the corresponding uhdr_hip_comm_*_dev call reports
 * UHDR_CODEC_UNSUPPORTED_FEATURE). */
typedef struct uhdr_hip_comm_ops {
  void* user;
  int (*all_reduce_min_f32)(void* user, float* buf, size_t n, void* hip_stream);                                    /* in place */
  int (*all_gather)(void* user, const void* send, void* recv, size_t bytes_per_rank, void* hip_stream);
  int (*gather_v)(void* user, const void* send, size_t send_bytes, void* recv, const size_t* counts, int root, void* hip_stream);
} uhdr_hip_comm_ops_t;
uhdr_error_info_t uhdr_hip_comm_init_custom(uhdr_hip_ctx_t* ctx, const uhdr_hip_comm_ops_t* ops, int rank, int nranks);
/* The path's ONE collective by itself: elementwise minimum of n floats across the communicator's ranks, in place on a DEVICE
 * buffer, enqueued on the context's stream (RCCL: ncclAllReduce(ncclMin); custom transport: all_reduce_min_f32).  The striped
 * entry points issue exactly this over {min0..2, -max0..2} between their passes (jpegr.cpp:932-938's mutex merge).  Without a
 * communicator the call does nothing. */
uhdr_error_info_t uhdr_hip_comm_all_reduce_min_dev(uhdr_hip_ctx_t* ctx, float* buf, size_t n);
/* Data movement between the ranks' devices on the context's stream (RCCL over xGMI by default): what replaces the
 * reference's threads writing their stripes into one buffer (jpegr.cpp:845-864).  all_gather: `bytes_per_rank` from every
 * rank, in rank order, on every rank.  gather: rank r's counts[r] bytes land at recv + sum(counts[0..r)) on `root` (stripes
 * of unequal height, per-stripe entropy-coded streams); counts is a host array, the same on every rank. */
uhdr_error_info_t uhdr_hip_comm_all_gather_dev(uhdr_hip_ctx_t* ctx, const void* send, void* recv, size_t bytes_per_rank);
uhdr_error_info_t uhdr_hip_comm_gather_dev(uhdr_hip_ctx_t* ctx, const void* send, size_t send_bytes, void* recv, const size_t* counts, int root);
uhdr_error_info_t uhdr_hip_generate_gainmap_striped_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* sdr_stripe,
                                                        const uhdr_raw_image_t* hdr_stripe, const uhdr_hip_encode_cfg_t* cfg,
                                                        uhdr_gainmap_metadata_t* gainmap_metadata, uhdr_raw_image_t* gainmap_stripe);
uhdr_error_info_t uhdr_hip_tone_map_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* hdr_intent,
                                        uhdr_raw_image_t* sdr_intent);
uhdr_error_info_t uhdr_hip_convert_yuv_dev(uhdr_hip_ctx_t* ctx, uhdr_raw_image_t* image,
                                           uhdr_color_gamut_t src_encoding,
                                           uhdr_color_gamut_t dst_encoding);
uhdr_error_info_t uhdr_hip_convert_raw_input_to_ycbcr_dev(uhdr_hip_ctx_t* ctx,
                                                          const uhdr_raw_image_t* src,
                                                          int chroma_sampling_enabled,
                                                          uhdr_raw_image_t* dst);

/* ---- image effects ---------------------------------------------------------------
 * apply_rotate / apply_mirror / apply_crop / apply_resize of the reference's effects chain (lib/src/editorhelper.cpp:
 * 210-520; resize = the chain's nearest-sample resize_buffer, :77-87), bit-exact element remaps.
 *   effect 0 rotate:  p0 = degrees clockwise (90, 180, 270)            dst is h x w for 90 / 270
 *   effect 1 mirror:  p0 = uhdr_mirror_direction_t (0 vertical, 1 horizontal)
 *   effect 2 crop:    p0 = left, p1 = top; the crop size is dst->w x dst->h
 *   effect 3 resize:  to dst->w x dst->h
 * dst: caller provides fmt (== src->fmt), w, h, planes and strides (the reference allocates with strides aligned to 64).
 * Formats: every raw format of the library (P010, YCbCr 4:2:0 / 4:4:4, 10-bit 4:4:4, Y400, RGBA8888, RGBA1010102,
 * RGBA-F16).  Host (no suffix) or device (_dev) plane pointers. */
uhdr_error_info_t uhdr_hip_apply_effect(uhdr_hip_ctx_t* ctx, int effect, int p0, int p1, const uhdr_raw_image_t* src, uhdr_raw_image_t* dst);
uhdr_error_info_t uhdr_hip_apply_effect_dev(uhdr_hip_ctx_t* ctx, int effect, int p0, int p1, const uhdr_raw_image_t* src, uhdr_raw_image_t* dst);

/* ---- JPEG DCT/quantize stage ----------------------------------------------------------------- */
/* Quant table libjpeg builds for jpeg_set_quality(quality, TRUE): natural (row-major) order. */
void uhdr_hip_jpeg_quant_table(int quality, int is_chroma, uint16_t qtable[64]);
/* Host utility (no GPU needed): the table the HLG / PQ decode tail runs on.  thresholds[c], c = 1..1023,
 * is the smallest clamped linear value whose 10-bit output code (jpegr.cpp:1775-1805: [pow 1/1.2,]
 * OETF LUT, colorToRgba1010102 quantisation) is >= c, computed with the host's libm; 2.0f = the code is
 * never reached.  Returns 0, or -1 for a transfer other than HLG / PQ. */
int uhdr_hip_oetf_code_thresholds(uhdr_color_transfer_t ct, float thresholds[1024]);
/* Host utility (no GPU needed): runs the encode kernels' table-driven float64 math (csrc/exact_math.h,
 * the same source the device compiles) on the host so it can be compared with libm without a GPU.
 *   fn 0: srgbOetf(in)                      (gainmapmath.cpp:139-148, powf replaced)
 *   fn 1: (float)log2((double)in)           (gainmapmath.cpp:767, 774, double log2 replaced)
 *   fn 2: division by a library constant: in[0] = b, out[0] = 1/b, out[i] = in[i] / b by the kernels'
 *         reciprocal-multiply-and-correct sequence (csrc/device_math.h div_const), i >= 1
 *   fn 3: division by an arbitrary per-call divisor through its float64 reciprocal: in[0] = b,
 *         out[i] = (float)((double)in[i] * (1.0 / (double)b)) (csrc/device_math.h div_by_rcp64), i >= 1
 *   fn 5: srgbOetf(in) through the direct pow table (round 4: csrc/exact_math.h srgb_oetf_direct, what the kernels run now)
 *   fn 4: (a, b) pairs: out[2i] = a / b through a float64 reciprocal refined from a deliberately
 *         2-ulp-off float seed (csrc/device_math.h rcp64_of_f32, the tone mapper's shared-divisor divisions)
 * Returns 0, or -1 for an unknown fn. */
int uhdr_hip_exact_math_eval(int fn, const float* in, float* out, size_t n);
/* Device self-test (needs the GPU): exhaustive sweeps of the instruction-level shortcuts of the encode kernels
 * (csrc/encode_core.h, csrc/selftest.hip) -- what the hardware's v_rcp_f32 / v_cvt_rpi_i32_f32 return cannot be a CPU test.
 * out[8] receives counters; the caller asserts.
 *   which 0: v_cvt_rpi_i32_f32(x) == floor(x + 0.5) for every float of [0, 2^23]      out = {mismatches, floats swept}
 *   which 1: refined reciprocal == RN(1 / b) for every normal float (both signs)      out = {mismatches, floats swept, raw v_rcp_f32 not correctly rounded}
 *   which 2: Markstein quotient == IEEE a / b on random pairs, biased exponents in [arg0, arg1], RNG seed `seed`
 *                                                                                      out = {mismatches, pairs}
 *   which 3: srgbOetf through the direct pow table, LDS form == generic form == round-1 form, every float of [0, 1]
 *                                                                                      out = {LDS != generic, generic != round-1, floats swept}
 *   which 4: the device-built ratio -> byte step table of two-pass generateGainMap for the FINAL range mm = {min0..2, max0..2}
 *            (log2 gains, as uhdr_hip_generate_gainmap_finalize returns it), channel arg0 of arg1 channels, against the
 *            per-sample evaluation over the table's whole domain and 2^20 bit patterns beyond either end
 *                                                                                      out = {mismatches, ratios swept, table entries, 1 if the range got no table}
 * The reference has no counterpart: these pin the arithmetic of jpegr.cpp:753-1013, 1945-1983 and gainmapmath.cpp:127-148 as evaluated here. */
uhdr_error_info_t uhdr_hip_selftest(uhdr_hip_ctx_t* ctx, int which, unsigned int arg0, unsigned int arg1, unsigned int seed, const float mm[6],
                                    unsigned long long out[8]);
/* Host utility (no GPU needed): evaluates one of the kernels' step tables (monotone float -> code functions stored as
 * bucket tables in LDS: csrc/host_tables.cpp build_step_table) exactly as the device does, so that tests can compare it
 * with the composite it stands for.
 *   which 0: toneMap's sRGB byte  put8(srgbOetf(clamp01(x)))         (jpegr.cpp:1976-1977, gainmapmath.cpp:139-148, 538-552)
 *   which 1: encodeGain's byte for gain x with min / max boost a / b, gamma 1  (gainmapmath.cpp:758-771)
 *   which 2 / 3: the HLG / PQ decode tail's 10-bit code               (jpegr.cpp:1775-1805)
 *   which 4 / 5: the same including the nit scaling in front of it, code(clamp01((x * 203) / peak))
 *   which 6: a synthetic staircase on [0, 1] for tests: code 0 below a (its bit pattern rounded down to a bucket start of
 *            2^15 patterns), one more at a and every b bit patterns (likewise rounded) after it -- steps exactly on bucket starts
 * info (may be NULL) receives {verified exact, entries, shift, first bucket}.  Returns 0; 1 when the table could not be
 * verified exact for these parameters (the kernels then keep the arithmetic evaluation); -1 for a bad argument. */
int uhdr_hip_step_table_eval(int which, float a, float b, const float* in, uint32_t* out, size_t n, uint32_t info[4]);
/* islow 8x8 FDCT + quantize of one 8-bit plane.  Reads blocks_w*8 x blocks_h*8 samples (the
 * caller pads to the MCU grid exactly as jpegencoderhelper.cpp:246-309 does); writes blocks in
 * raster order, 64 int16 each in natural order = libjpeg's JBLOCK layout, ready for
 * jpeg_write_coefficients().  plane / coef are host (no suffix) or device (_dev) pointers. */
uhdr_error_info_t uhdr_hip_fdct_quant(uhdr_hip_ctx_t* ctx, const uint8_t* plane, size_t stride,
                                      int blocks_w, int blocks_h, const uint16_t qtable[64],
                                      int16_t* coef);
uhdr_error_info_t uhdr_hip_fdct_quant_dev(uhdr_hip_ctx_t* ctx, const uint8_t* plane, size_t stride,
                                          int blocks_w, int blocks_h, const uint16_t qtable[64],
                                          int16_t* coef);

/* MI355X extension: the three full-image loops JpegR::encodeJPEGR API-0 runs back to back (lib/src/jpegr.cpp:202-251:
 * toneMap, generateGainMap on the tone-mapped image, convert_raw_input_to_ycbcr for the base JPEG) in ONE pass over the
 * HDR image: 4 B/px in, 3 + 3 B/px out instead of 26 B/px of traffic.  Device images only.  hdr: RGBA1010102 or
 * RGBA-F16; cfg->map_dimension_scale_factor must be 1; sdr_rgba (may be NULL, or planes[0] NULL) optionally receives
 * the RGBA8888 SDR rendition; base_ycc receives YCbCr 4:4:4 planes; md / gainmap as uhdr_hip_generate_gainmap_dev.
 * Outputs are bit-identical to uhdr_hip_tone_map_dev -> uhdr_hip_generate_gainmap_dev ->
 * uhdr_hip_convert_raw_input_to_ycbcr_dev(.., 0, ..). */
uhdr_error_info_t uhdr_hip_encode_api0_fused_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* hdr,
                                                 const uhdr_hip_encode_cfg_t* cfg, uhdr_raw_image_t* sdr_rgba,
                                                 uhdr_raw_image_t* base_ycc, uhdr_gainmap_metadata_t* metadata,
                                                 uhdr_raw_image_t* gainmap);

/* copy_raw_image(src, dst) (lib/src/gainmapmath.cpp:1492-1613) between device images: strided plane copies
 * for equal formats, RGB888 -> RGBA8888 (alpha 0xff), RGBA8888 -> Y400 (R byte); same error codes
 * (UHDR_CODEC_MEM_ERROR for a size mismatch, UHDR_CODEC_UNSUPPORTED_FEATURE for other format pairs).
 * The host <-> device direction of this function is what the host-buffer entry points do internally. */
uhdr_error_info_t uhdr_hip_copy_raw_image_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* src,
                                              uhdr_raw_image_t* dst);

/* MI355X extension for 3-channel gain maps: uhdr_hip_jpeg_rgb_to_ycc followed by uhdr_hip_fdct_quant of Y (luma
 * table), Cb and Cr (chroma table) -- what libjpeg does to a JCS_RGB image at 4:4:4 (jpegencoderhelper.cpp:165-167,
 * 212-225) -- in one pass: 3 (4) B/px in, 6 B/px out instead of 15 B/px.  Device image, RGB888 or RGBA8888, w and h
 * multiples of 8, rows 8- (16-) byte aligned; three coefficient arrays of (w/8)*(h/8) JBLOCKs, bit-identical to the
 * two-step route. */
uhdr_error_info_t uhdr_hip_fdct_quant_rgb_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* rgb,
                                              const uint16_t qtable_luma[64], const uint16_t qtable_chroma[64],
                                              int16_t* coef_y, int16_t* coef_cb, int16_t* coef_cr);

/* ---- JPEG decode stage (the step immediately before applyGainMap) ---------------
 * Inverse of uhdr_hip_fdct_quant: dequantize + libjpeg's JDCT_ISLOW 8x8 inverse DCT + range limit of
 * coefficient blocks as jpeg_read_coefficients() yields them (JBLOCK layout, raster block order), so that
 * only the Huffman decode stays on the CPU and the 8-bit planes JpegDecoderHelper::decompressImage
 * (jpegdecoderhelper.cpp:169-535, dct_method = JDCT_ISLOW at :283) would produce never visit host memory.
 * Writes blocks_w*8 x blocks_h*8 samples (plane must have room for the block padding).  coef must be
 * 16-byte aligned.  Host (no suffix) or device (_dev) pointers. */
uhdr_error_info_t uhdr_hip_idct_dequant(uhdr_hip_ctx_t* ctx, const int16_t* coef, int blocks_w,
                                        int blocks_h, const uint16_t qtable[64], uint8_t* plane,
                                        size_t stride);
uhdr_error_info_t uhdr_hip_idct_dequant_dev(uhdr_hip_ctx_t* ctx, const int16_t* coef, int blocks_w,
                                            int blocks_h, const uint16_t qtable[64], uint8_t* plane,
                                            size_t stride);
/* libjpeg's colour conversions around a 3-channel gain map.  rgb_to_ycc = jccolor.c rgb_ycc_convert,
 * what jpeg_write_scanlines applies when the encoder hands the map over as JCS_RGB
 * (jpegencoderhelper.cpp:165-167, 212-225): src UHDR_IMG_FMT_24bppRGB888 or 32bppRGBA8888 (alpha
 * ignored), dst UHDR_IMG_FMT_24bppYCbCr444 planes for uhdr_hip_fdct_quant.  ycc_to_rgb = jdcolor.c
 * ycc_rgb_convert, what jpeg_read_scanlines applies when the decoder asks for RGB / RGBA output
 * (jpegdecoderhelper.cpp:400-470): src 24bppYCbCr444, dst RGB888 or RGBA8888 (alpha 255).
 * libjpeg_variant selects the green-channel constants: 0 = libjpeg 6b / libjpeg-turbo (0.71414,
 * 0.34414; the reference pins libjpeg-turbo), 1 = IJG 9 (0.714136286, 0.344136286); they differ for 59
 * of the 65536 (Cb, Cr) pairs.  rgb_to_ycc has no variant: both constant sets give identical bytes. */
uhdr_error_info_t uhdr_hip_jpeg_rgb_to_ycc(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* rgb,
                                           uhdr_raw_image_t* ycc);
uhdr_error_info_t uhdr_hip_jpeg_rgb_to_ycc_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* rgb,
                                               uhdr_raw_image_t* ycc);
uhdr_error_info_t uhdr_hip_jpeg_ycc_to_rgb(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* ycc,
                                           int libjpeg_variant, uhdr_raw_image_t* rgb);
uhdr_error_info_t uhdr_hip_jpeg_ycc_to_rgb_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* ycc,
                                               int libjpeg_variant, uhdr_raw_image_t* rgb);

/* MI355X extension, the decode-side mirror of uhdr_hip_fdct_quant_rgb_dev: three uhdr_hip_idct_dequant calls (Y with
 * the luma table, Cb and Cr with the chroma table) followed by uhdr_hip_jpeg_ycc_to_rgb -- what libjpeg does when
 * JpegDecoderHelper asks for RGB / RGBA scanlines of a 3-channel (4:4:4) gain-map JPEG (jpegdecoderhelper.cpp:400-470)
 * -- in one pass: 6 B/px in, 3 (4) B/px out instead of 16 B/px over four launches.  Device pointers; three coefficient
 * arrays of blocks_w * blocks_h JBLOCKs, 16-byte aligned; rgb is a device image (RGB888 or RGBA8888, alpha 255) with
 * ceil(w/8) == blocks_w and ceil(h/8) == blocks_h: only the w x h valid pixels are stored.  Bit-identical to the
 * four-step route. */
uhdr_error_info_t uhdr_hip_idct_dequant_rgb_dev(uhdr_hip_ctx_t* ctx, const int16_t* coef_y,
                                                const int16_t* coef_cb, const int16_t* coef_cr,
                                                int blocks_w, int blocks_h,
                                                const uint16_t qtable_luma[64],
                                                const uint16_t qtable_chroma[64], int libjpeg_variant,
                                                uhdr_raw_image_t* rgb);

/* MI355X extension, SURVEY.md 8f-1 as worded ("GPU dequant + IDCT fused into apply_gainmap"): applyGainMap on a
 * 4:2:0 base image that is still in coefficient form -- what jpeg_read_coefficients() yields after the CPU's Huffman
 * decode (JBLOCK arrays of width_in_blocks x height_in_blocks blocks per component, compact, raster block order,
 * DEVICE pointers, 16-byte aligned; quantval tables in natural order).  The kernel dequantizes and inverse-transforms
 * (JDCT_ISLOW, as jpegdecoderhelper.cpp:283 selects) each 128 x 16 pixel tile in LDS and applies the gain map from
 * there: the 8-bit Y / Cb / Cr planes never exist in memory (3 B/px of coefficients in instead of 3 in + 1.5 out +
 * 1.5 in over four launches).  Result == uhdr_hip_idct_dequant_dev x 3 followed by uhdr_hip_apply_gainmap_dev on the
 * w x h image, bit for bit.  base_cg is the colour gamut of the base image (what uhdr_raw_image_t::cg would carry);
 * gainmap_img is a decoded device image as for uhdr_hip_apply_gainmap_dev.  Covers the cases of the library's 2x2-quad
 * kernel (even w and h, w >= 128, destination rows 16-byte aligned, map at scale 1 or an even scale <= 8 with
 * gamma 1); UHDR_CODEC_UNSUPPORTED_FEATURE otherwise (decode with uhdr_hip_idct_dequant_dev, then apply). */
typedef struct uhdr_hip_jpeg_coefficients {
  const int16_t* coef[3];  /* Y, Cb, Cr */
  int blocks_w[3];         /* jpeg_component_info::width_in_blocks */
  int blocks_h[3];         /* jpeg_component_info::height_in_blocks */
  uint16_t qtable[3][64];  /* comp_info[c].quant_table->quantval */
} uhdr_hip_jpeg_coefficients_t;
uhdr_error_info_t uhdr_hip_apply_gainmap_coef_dev(uhdr_hip_ctx_t* ctx,
                                                  const uhdr_hip_jpeg_coefficients_t* base,
                                                  unsigned int w, unsigned int h,
                                                  uhdr_color_gamut_t base_cg,
                                                  const uhdr_raw_image_t* gainmap_img,
                                                  const uhdr_gainmap_metadata_t* gainmap_metadata,
                                                  uhdr_color_transfer_t output_ct,
                                                  uhdr_img_fmt_t output_format, float max_display_boost,
                                                  uhdr_raw_image_t* dest);

/* ---- JPEG entropy stage (the step after uhdr_hip_fdct_quant) ---------------------------------
 * Baseline Huffman coding of quantized coefficient blocks with the Annex K tables -- what libjpeg does behind
 * JpegEncoderHelper::compressImage (jpegencoderhelper.cpp:131-244: jpeg_set_defaults, optimize_coding off).  libjpeg's
 * single sequential pass chains every block to its predecessor through the DC prediction.  Two parallel forms:
 * restart_interval == 0 -- the stream the reference itself writes (jpegencoderhelper.cpp:187-201 never sets
 * cinfo.restart_interval): nothing in encoding is serial on a device that holds all coefficients (a block's DC
 * difference needs its predecessor's DC VALUE, its bit position is a prefix sum of code lengths, byte stuffing a prefix
 * sum of 0xFF counts): lengths -> scan -> emit -> stuff, three passes over all blocks; the bytes are libjpeg's.
 * restart_interval > 0 -- JPEG's own restart intervals (T.81 B.2.4.4): every restart_interval MCUs the stream is
 * byte-aligned, an RSTn marker is written and the predictors reset, so one wavefront encodes one interval (<= 64 blocks:
 * restart_interval <= 10 for 4:2:0, <= 21 for 4:4:4, <= 64 for one component); this is the form that shards over ranks.
 * Parity policy: the entropy-coded data is byte-identical to what libjpeg emits for the same coefficients with
 * cinfo.restart_interval set to the same value (dummy blocks at the right / bottom edge included); with intervals that
 * adds a DRI segment and the RSTn markers relative to the reference's files, the decoded coefficients are identical.
 * coef[c]: DEVICE JBLOCK arrays (as uhdr_hip_fdct_quant_dev writes them) of blocks_w[c] x blocks_h[c] blocks, between
 * the component's real grid (jpeg_component_info::width_in_blocks / height_in_blocks) and the MCU-padded grid; blocks
 * an MCU needs beyond the array are libjpeg's dummy blocks.  One component: non-interleaved scan (an MCU is a block).
 * out: DEVICE buffer; receives everything between the SOS header and EOI, RSTn markers included; *out_bytes its
 * length (UHDR_CODEC_MEM_ERROR with the needed size in *out_bytes when out_capacity is too small).  Synchronous.
 * Coefficients must be in the baseline range (as an 8-bit FDCT produces): UHDR_CODEC_INVALID_PARAM otherwise. */
typedef struct uhdr_hip_jpeg_scan {
  int num_components;     /* 1 or 3 */
  const int16_t* coef[3];
  int blocks_w[3];
  int blocks_h[3];
  int h_samp[3];          /* sampling factors, 1 or 2 (ignored for one component) */
  int v_samp[3];
  unsigned int w, h;      /* image dimensions in pixels */
  int restart_interval;   /* in MCUs; 0 = no restart markers */
} uhdr_hip_jpeg_scan_t;
uhdr_error_info_t uhdr_hip_huffman_encode_dev(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_scan_t* scan,
                                              uint8_t* out, size_t out_capacity, size_t* out_bytes);
/* The inverse (what libjpeg's jdhuff.c does behind JpegDecoderHelper::decompressImage, jpegdecoderhelper.cpp:169-535):
 * entropy-coded data -> quantized coefficient blocks, ready for uhdr_hip_idct_dequant_dev / uhdr_hip_apply_gainmap_coef_dev.
 * A stream with restart markers is a sequence of independent intervals; the markers are located on the device and every
 * interval is decoded by its own lane.  A stream without them (restart_interval 0 -- every file the reference writes) is
 * ONE interval: it is decoded by the self-synchronising parallel decoder (huffman_decode_sync.hip: sub-sequences of the
 * bit stream are decoded speculatively and re-decoded until every hand-over state agrees with its predecessor's end
 * state), falling back to a single lane only for streams shorter than 4 KiB or if the search does not settle.  Restart
 * files whose intervals are long (>= 320 bytes on average: one lane per interval would leave the device idle) take the
 * same decoder with the markers spliced out; one that is not what its headers say is handed to the interval decoder,
 * which words the error.
 * scan->coef[c]: DEVICE arrays of blocks_w[c] x blocks_h[c] JBLOCKs, WRITTEN by this call (dummy blocks of edge MCUs are
 * dropped); data: DEVICE pointer to the bytes between the SOS header and EOI; tables: the file's DHT content in the order
 * DC luma (Tc 0, Th 0), AC luma, DC chroma, AC chroma -- NULL selects the Annex K tables; component 0 uses the luma pair,
 * the others the chroma pair.  Synchronous.  UHDR_CODEC_INVALID_PARAM for malformed data (undefined code, run past the
 * end of a block, marker count / numbering that does not fit the restart interval). */
typedef struct uhdr_hip_huff_tables {
  uint8_t bits[4][17]; /* BITS: bits[t][l] = number of codes of length l (index 0 unused) */
  uint8_t vals[4][256]; /* HUFFVAL */
} uhdr_hip_huff_tables_t;
uhdr_error_info_t uhdr_hip_huffman_decode_dev(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_scan_t* scan,
                                              const uhdr_hip_huff_tables_t* tables, const uint8_t* data,
                                              size_t data_bytes);
/* Round 5: the TWO scans of one UltraHDR file (base image and gain map) coded / decoded concurrently -- same arguments, results
 * and error behaviour as two calls of uhdr_hip_huffman_encode_dev / _decode_dev (scan a on the context's stream from the calling
 * thread, scan b on an auxiliary stream of the same device from a second thread; the first error is returned).  The entropy
 * stages are latency- and occupancy-bound for long stretches, which the two streams overlap (4K API-1 pair: see DESIGN.md 5.5).
 * Synchronous; the context's stream is synchronised first (the inputs of both scans were produced on it). */
uhdr_error_info_t uhdr_hip_huffman_encode2_dev(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_scan_t* scan_a, uint8_t* out_a, size_t capacity_a,
                                               size_t* bytes_a, const uhdr_hip_jpeg_scan_t* scan_b, uint8_t* out_b, size_t capacity_b,
                                               size_t* bytes_b);
uhdr_error_info_t uhdr_hip_huffman_decode2_dev(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_scan_t* scan_a,
                                               const uhdr_hip_huff_tables_t* tables_a, const uint8_t* data_a, size_t data_bytes_a,
                                               const uhdr_hip_jpeg_scan_t* scan_b, const uhdr_hip_huff_tables_t* tables_b,
                                               const uint8_t* data_b, size_t data_bytes_b);
/* Host helper (no device work): wraps entropy-coded data (HOST pointer) into a complete baseline JFIF file -- SOI,
 * APP0, DQT (natural-order tables as uhdr_hip_jpeg_quant_table returns them; component 0 uses qtable_luma, the others
 * qtable_chroma), SOF0, DHT (Annex K), DRI, SOS, data, EOI, the marker order of libjpeg's jcmarker.c.  scan->coef is
 * not read.  Returns the file size, 0 when out_capacity is too small or the description is invalid. */
size_t uhdr_hip_jpeg_assemble(const uhdr_hip_jpeg_scan_t* scan, const uint16_t qtable_luma[64],
                              const uint16_t qtable_chroma[64], const uint8_t* scan_data, size_t scan_bytes,
                              uint8_t* out, size_t out_capacity);

/* Host helper (no device work), the counterpart of uhdr_hip_jpeg_assemble: reads the headers of a baseline JPEG file --
 * what jpeg_read_header does for JpegDecoderHelper (jpegdecoderhelper.cpp:212-222) -- and returns what the device decode
 * path needs: the scan description (component grids = jpeg_component_info::width_in_blocks / height_in_blocks, sampling
 * factors, restart interval; coef pointers left NULL for the caller's device arrays), the Huffman tables in
 * uhdr_hip_huffman_decode_dev's order, each component's quantization table in natural order, and where the entropy-coded
 * data sits inside the file.  Accepts what libjpeg writes with default settings (hence every base image / gain map of an
 * UltraHDR file): SOF0, 8 bit, 1 or 3 components in one scan, sampling factors 1 or 2, components 1 and 2 sharing their
 * tables.  Returns 0, or a negative number naming the first thing it does not handle (-8: not baseline sequential). */
typedef struct uhdr_hip_jpeg_header {
  uhdr_hip_jpeg_scan_t scan;
  uhdr_hip_huff_tables_t tables;
  uint16_t qtable[3][64];
  size_t scan_offset; /* first byte after the SOS header */
  size_t scan_bytes;  /* up to (not including) the marker that ends the entropy-coded data */
} uhdr_hip_jpeg_header_t;
int uhdr_hip_jpeg_parse(const uint8_t* file, size_t size, uhdr_hip_jpeg_header_t* out);

/* JpegDecoderHelper::decompressImage (jpegdecoderhelper.cpp:169-535) for one baseline JPEG whose headers are already
 * parsed -- by uhdr_hip_jpeg_parse, or by libjpeg's jpeg_read_header as in facade/uhdr_hip_jpeg_seam.cpp, which fills the
 * same struct from jpeg_decompress_struct.  Entropy decode (uhdr_hip_huffman_decode_dev), dequantization + JDCT_ISLOW
 * IDCT (uhdr_hip_idct_dequant_dev) and, for RGB output, ycc_rgb_convert (uhdr_hip_idct_dequant_rgb_dev) all run on the
 * device: the compressed bytes go up, the decoded samples come down, nothing else crosses PCIe.
 * hdr: scan geometry (coef pointers ignored), Huffman tables, per-component quantization tables; scan_offset / scan_bytes
 * ignored.  scan_data: HOST pointer to the first byte after the SOS header; scan_bytes: bytes available from there (the
 * rest of the file is fine: the data is cut at the first marker that is not RSTn).
 * out_channels 0: planar output as libjpeg's raw-data mode gives it -- plane i gets min(hstride[i], blocks_w[i]*8) columns
 * of min(vstride[i], blocks_h[i]*8) rows (HOST pointers, row pitch hstride[i] bytes), i.e. hstride / vstride describe the
 * caller's buffer like JpegDecoderHelper's mPlaneHStride / mPlaneVStride.  out_channels 3 / 4: packed RGB888 / RGBA8888
 * (alpha 255) of a 3-component 4:4:4 file into planes[0], hstride[0] in PIXELS; libjpeg_variant as for
 * uhdr_hip_jpeg_ycc_to_rgb.  UHDR_CODEC_UNSUPPORTED_FEATURE: RGB output of a subsampled file (libjpeg's fancy upsampling).
 * Synchronous. */
uhdr_error_info_t uhdr_hip_jpeg_decode_scan(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_header_t* hdr,
                                            const uint8_t* scan_data, size_t scan_bytes, int out_channels,
                                            int libjpeg_variant, uint8_t* const planes[3],
                                            const unsigned int hstride[3], const unsigned int vstride[3]);

/* The encode-side mirror: JpegEncoderHelper::compressImage's sample -> entropy-coded-data part
 * (jpegencoderhelper.cpp:131-309) on the device.  FDCT + quantization (uhdr_hip_fdct_quant_dev; for a packed RGB gain map
 * uhdr_hip_fdct_quant_rgb_dev, i.e. rgb_ycc_convert included) feed uhdr_hip_huffman_encode_dev without the coefficient
 * blocks leaving HBM: the samples go up, the compressed bytes come down.
 * scan: geometry (coef pointers ignored); restart_interval as for uhdr_hip_huffman_encode_dev: 0 (no markers: the
 * reference's bytes, what the facade uses) or 1..64 / blocks per MCU (RSTn markers, see that function's parity policy).
 * qtable[c]: natural order.
 * rgb_channels 0: planes[c] = HOST plane of blocks_w[c]*8 x blocks_h[c]*8 samples (the caller pads to whole blocks as
 * jpegencoderhelper.cpp:246-309 does), strides in bytes; 3 / 4: planes[0] = HOST packed RGB888 / RGBA8888 image of
 * scan->w x scan->h pixels (multiples of 8), strides[0] in PIXELS, for a 3-component 4:4:4 scan.
 * out (HOST) receives everything between the SOS header and EOI; *out_bytes its length (UHDR_CODEC_MEM_ERROR with the
 * needed size in *out_bytes when out_capacity is too small).  Wrap it with uhdr_hip_jpeg_assemble.  Synchronous. */
uhdr_error_info_t uhdr_hip_jpeg_encode_scan(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_scan_t* scan,
                                            const uint16_t qtable[3][64], const uint8_t* const planes[3],
                                            const unsigned int strides[3], int rgb_channels, uint8_t* out,
                                            size_t out_capacity, size_t* out_bytes);
/* The same with the IMAGE's own planes -- JpegEncoderHelper::compressImage's contract (jpegencoderhelper.cpp:101-309): plane c
 * holds ceil(w * h_samp[c] / max_h) x ceil(h * v_samp[c] / max_v) samples (scan->blocks_w / _h = its REAL blocks, ceil(.. / 8)),
 * strides in bytes as the caller has them.  Whatever partial edge blocks need beyond those samples is made up on the device
 * exactly as the helper does on the host (jpegencoderhelper.cpp:246-309): with a stride that covers the block-aligned width
 * the columns behind the plane width are the caller's own bytes there and the rows below the plane height are 0 (component 0)
 * / 128; with a shorter stride the columns are 0 / 128 and the rows below repeat the helper's stale scratch rows (the rows one
 * MCU row up; zeros in the first MCU row).  Packed RGB (rgb_channels 3 / 4, any w x h): the last column / row is replicated,
 * which is what libjpeg's scanline pipeline does (jcsample.c expand_right_edge, jcprepct.c expand_bottom_edge).  Dummy blocks
 * that complete edge MCUs are the entropy coder's business (uhdr_hip_huffman_encode_dev).  A 1920x1080 4:2:0 base image
 * (960x540 chroma planes) or the 960x540 map of a 4K frame therefore encode on the device, byte-identical to the reference. */
uhdr_error_info_t uhdr_hip_jpeg_encode_image(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_scan_t* scan,
                                            const uint16_t qtable[3][64], const uint8_t* const planes[3],
                                            const unsigned int strides[3], int rgb_channels, uint8_t* out,
                                            size_t out_capacity, size_t* out_bytes);

/* ---- device-resident handoff between host-buffer calls -----------------------------------------------------------
 * JpegR::decodeJPEGR (jpegr.cpp:1467-1530) decodes the base image and the gain map into JpegDecoderHelper buffers and hands
 * exactly those buffers to applyGainMap.  Between _begin and _end, uhdr_hip_jpeg_decode_scan keeps what it wrote to the
 * caller's planes on the device (the two most recent images), and a host-buffer entry point that is handed an input image
 * with the same plane pointers, strides and format reads that copy instead of uploading the planes again.  Likewise on the
 * way out (JpegR::encodeJPEGR, jpegr.cpp:253-316): an 8-bit image a host-buffer entry point has just produced (gain map,
 * converted base image) stays on the device too, and uhdr_hip_jpeg_encode_scan, handed those host planes, reads it there.
 * The caller promises not to write to those host buffers in between (call _begin again to drop the copies if it does; the
 * library drops a copy itself whenever it rewrites the host buffer); outside a _begin/_end pair nothing is kept.  The
 * facade opens one pair per uhdr_encode / uhdr_decode. */
void uhdr_hip_resident_begin(uhdr_hip_ctx_t* ctx);
void uhdr_hip_resident_end(uhdr_hip_ctx_t* ctx);

/* ---- lazy downloads inside a resident session (round 4) ---------------------------------------------------------------
 * decodeJPEGR's two decoded images have exactly two readers: applyGainMap (jpegr.cpp:1527), which on this path reads the
 * device copies, and copy_raw_image(&gainmap, gainmap_img) (jpegr.cpp:1490), which fills the image uhdr_get_decoded_gainmap_
 * image (ultrahdr_api.cpp:2031-2043) hands out -- rarely asked for, 33 MB for a full-resolution RGBA map.  So:
 *   _lazy(ctx, 1)   until _lazy(ctx, 0) or _end: an image the handoff keeps on the device -- what uhdr_hip_jpeg_decode_scan decodes,
 *                   and an 8-bit image a host-buffer entry point produces (the facade switches it on around generateGainMap: the
 *                   map's only reader is the compressImage that follows, jpegr.cpp:253-257) -- is NOT written to the
 *                   caller's planes; the device copy IS the image.  Library entry points handed those planes read the device
 *                   copy, and the library writes the planes back by itself before it would read them from the host or reuse
 *                   the slot.  The caller calls _flush before anything ELSE reads them (the facade: before any CPU stage).
 *   _flush          writes every unwritten image back to its host planes and performs every adopted copy, now.
 *   _adopt(src,dst) stands for copy_raw_image(src, dst) of an unwritten single-plane image (8bppYCbCr400, RGB888, RGBA8888;
 *                   same size, and same format or copy_raw_image's RGB888 -> RGBA8888 with alpha 255 -- what a build against
 *                   IJG libjpeg decodes a three-channel map to): returns 1 when the copy is now the library's to make -- the device
 *                   copy then outlives _end, and _materialize (or _flush, inside the session) makes it; 0 when src is not
 *                   such an image and the caller copies on the host as before.  Only pixel data: cg / ct / range of dst are
 *                   the caller's.  One copy can be pending per context; dst must stay allocated until _materialize or _forget.
 *   _materialize    makes the pending copy (device -> dst planes); no-op when nothing is pending.
 *   _forget         dst is going away (uhdr_reset_decoder, ~uhdr_codec_private): nothing is pending any more.
 * _end discards unwritten images without writing them back: by ending the session the caller says nobody reads those host
 * planes any more (decodeJPEGR's JpegDecoderHelper locals are gone by then).  _begin forgets a pending copy. */
void uhdr_hip_resident_lazy(uhdr_hip_ctx_t* ctx, int on);
uhdr_error_info_t uhdr_hip_resident_flush(uhdr_hip_ctx_t* ctx);
int uhdr_hip_resident_adopt(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* src, const uhdr_raw_image_t* dst);
uhdr_error_info_t uhdr_hip_resident_materialize(uhdr_hip_ctx_t* ctx);
void uhdr_hip_resident_forget(uhdr_hip_ctx_t* ctx);

/* ---- API-1 encode chain without its round trips (MI355X extension, round 4) ---------------------------------------------
 * JpegR::encodeJPEGR API-1 (jpegr.cpp:253-316) = generateGainMap -> compress the map -> convertYuv of the base copy
 * (jpegr.cpp:436-518) -> compress the base.  This entry point runs the sample -> coefficient part of all of it in FOUR
 * launches on device-resident images: pass 1 of two-pass generation (gain ratios + extrema), the range / step-table kernel,
 * the map's pass 2 fused with libjpeg's rgb_ycc_convert and its three FDCT / quantize transforms (the 8-bit map is written
 * only if `gm` is given), and convertYuv fused with the FDCT / quantize of Y, Cb and Cr (the converted planes never exist).
 * Coefficient blocks and metadata are bit-identical to uhdr_hip_generate_gainmap_dev + uhdr_hip_convert_yuv_dev +
 * uhdr_hip_fdct_quant_dev x 3 + uhdr_hip_fdct_quant_rgb_dev; feed them to uhdr_hip_huffman_encode_dev.
 * sdr: UHDR_IMG_FMT_12bppYCbCr420, w and h multiples of 16 (read only); hdr as for generateGainMap; cfg: two-pass presets,
 * gamma 1, a scale factor that leaves map dimensions that are multiples of 8.  base_encoding: convertYuv's destination
 * encoding (UHDR_CG_DISPLAY_P3 = BT.601 in encodeJPEGR; UHDR_CG_UNSPECIFIED or sdr->cg: no conversion).  qt_*[0] luma,
 * [1] chroma tables, natural order.  blocks: DEVICE buffers, 16-byte aligned -- base_coef[0] (w/8)*(h/8) JBLOCKs, [1] [2]
 * (w/16)*(h/16); map_coef[c] (map_w/8)*(map_h/8) for the map's 1 or 3 components.  gm may be null.
 * UHDR_CODEC_UNSUPPORTED_FEATURE names what falls outside (the operators then do the job).  Synchronises once (metadata). */
typedef struct uhdr_hip_api1_blocks {
  int16_t* base_coef[3];
  int16_t* map_coef[3];
} uhdr_hip_api1_blocks_t;
uhdr_error_info_t uhdr_hip_encode_api1_fused_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                                 const uhdr_hip_encode_cfg_t* cfg, uhdr_color_gamut_t base_encoding,
                                                 const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                                 const uhdr_hip_api1_blocks_t* blocks, uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gm);

/* ---- JpegR::encodeJPEGR API-1 behind ONE entry point (round 5; what the facade's seam at encodeJPEGR binds) ------------------
 * jpegr.cpp:253-316 = generateGainMap -> compressGainMap -> convertYuv(base, BT.601) -> compressImage(base) -> appendGainMap.
 * Everything of it that touches samples runs here as one device sequence on HOST images: the two raw intents go up once (a
 * few threads feed a pinned ring, the DMA engine drains it), uhdr_hip_encode_api1_fused_dev leaves all coefficient blocks in
 * HBM, uhdr_hip_huffman_encode_dev (restart_interval 0: the reference's own bytes) codes the base image's interleaved 4:2:0
 * scan and the map's 4:4:4 (or one-component) scan, and only those two byte strings come down.  The 8-bit gain map and the
 * converted base planes never exist; the container around the scans (SOI .. SOS headers, ICC, MPF, XMP / ISO metadata --
 * jpegr.cpp:1100-1300) stays the caller's.
 * sdr / hdr / cfg / base_encoding / qt_base / qt_map: as for uhdr_hip_encode_api1_fused_dev, with HOST plane pointers.
 * gainmap_desc (may be NULL): receives fmt / w / h / cg / ct / range of the gain-map image generateGainMap would have
 * allocated (jpegr.cpp:714-716); its planes are not touched.  base_scan / map_scan: HOST buffers; *_bytes = sizes written
 * (UHDR_CODEC_MEM_ERROR with the needed size in the corresponding *_bytes when a capacity is too small -- call again).
 * UHDR_CODEC_UNSUPPORTED_FEATURE: a combination the fused chain does not take (dimensions not multiples of 16, one-pass
 * preset, gamma != 1 ...) -- nothing was uploaded; use the per-stage entry points.  Synchronous. */
uhdr_error_info_t uhdr_hip_encode_api1_scans(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                             const uhdr_hip_encode_cfg_t* cfg, uhdr_color_gamut_t base_encoding,
                                             const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                             uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc,
                                             uint8_t* base_scan, size_t base_capacity, size_t* base_bytes,
                                             uint8_t* map_scan, size_t map_capacity, size_t* map_bytes);

/* JpegR::encodeJPEGR API-0 (lib/src/jpegr.cpp:179-244) for the HDR intents whose tone-mapped rendition is RGBA8888 -- RGBA1010102 and RGBA
 * half float -- in ONE call (round 6): host intent in, the two entropy-coded scans out (host buffers).  uhdr_hip_encode_api0_fused_dev (tone
 * map + one-pass gain map + RGB -> YCbCr 4:4:4) + FDCT / quantize + both scans Huffman-coded; what the facade binds at encodeJPEGR API-0.
 * cfg: preset is overridden to UHDR_USAGE_REALTIME and use_luminance to 0, as the reference does on this path; scale factor 1 and
 * dimensions that are multiples of 8 (UHDR_CODEC_UNSUPPORTED_FEATURE otherwise: the per-stage operators take those).  *sdr_cg: the colour
 * gamut of the tone-mapped rendition (for the base file's ICC profile).  qt_base / qt_map: {luma, chroma} tables. */
uhdr_error_info_t uhdr_hip_encode_api0_scans(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* hdr, const uhdr_hip_encode_cfg_t* cfg,
                                             const uint16_t qt_base[2][64], const uint16_t qt_map[2][64], uhdr_gainmap_metadata_t* md,
                                             uhdr_raw_image_t* gainmap_desc, uhdr_color_gamut_t* sdr_cg, uint8_t* base_scan,
                                             size_t base_capacity, size_t* base_bytes, uint8_t* map_scan, size_t map_capacity,
                                             size_t* map_bytes);

/* The same on DEVICE-resident intents into DEVICE buffers, and its inverse (round 6): one entry point per direction of the API-1
 * round trip, for callers whose images live in HBM (a transcoding service; bench.py's headline).
 * uhdr_hip_encode_api1_scans_dev: JpegR::encodeJPEGR API-1 (jpegr.cpp:253-316) without the container -- sdr / hdr are device images,
 * base_scan / map_scan device buffers; everything else as uhdr_hip_encode_api1_scans.  Synchronous (the byte counts come back).
 * uhdr_hip_decode_api1_scans_dev: JpegR::decodeJPEGR behind its container parsing (jpegr.cpp:1469-1531) -- the two JPEG streams'
 * headers (uhdr_hip_jpeg_parse, or filled by the caller: scan geometry, DQT, DHT -- all zeros selects the Annex K tables; scan.coef is ignored) and their entropy-coded
 * bytes in device memory (between the SOS header and the marker that ends them) -> both scans entropy-decoded concurrently ->
 * the gain map's dequant + IDCT (+ ycc -> rgb, alpha 255, for a three-channel map) -> applyGainMap (base image's dequant + IDCT inside
 * the kernel) -> dest (device image).  base: a 4:2:0 three-component scan; map: one component, or three at 4:4:4.  base_cg / map_cg:
 * what the decoded images' uhdr_raw_image_t::cg would carry (the ICC profile's gamut; the gain map's from the metadata's use_base_cg
 * logic upstream); libjpeg_variant: the ycc -> rgb constants of a three-channel map, as for uhdr_hip_jpeg_ycc_to_rgb (0: libjpeg-turbo, which
 * the reference pins).  The entropy stage is synchronous; the two sample-domain launches behind it are only enqueued. */
uhdr_error_info_t uhdr_hip_encode_api1_scans_dev(uhdr_hip_ctx_t* ctx, const uhdr_raw_image_t* sdr, const uhdr_raw_image_t* hdr,
                                                 const uhdr_hip_encode_cfg_t* cfg, uhdr_color_gamut_t base_encoding,
                                                 const uint16_t qt_base[2][64], const uint16_t qt_map[2][64],
                                                 uhdr_gainmap_metadata_t* md, uhdr_raw_image_t* gainmap_desc,
                                                 uint8_t* base_scan, size_t base_capacity, size_t* base_bytes,
                                                 uint8_t* map_scan, size_t map_capacity, size_t* map_bytes);
uhdr_error_info_t uhdr_hip_decode_api1_scans_dev(uhdr_hip_ctx_t* ctx, const uhdr_hip_jpeg_header_t* base, const uint8_t* base_data,
                                                 size_t base_bytes, uhdr_color_gamut_t base_cg, const uhdr_hip_jpeg_header_t* map,
                                                 const uint8_t* map_data, size_t map_bytes, uhdr_color_gamut_t map_cg,
                                                 int libjpeg_variant, const uhdr_gainmap_metadata_t* md, uhdr_color_transfer_t output_ct,
                                                 uhdr_img_fmt_t output_format, float max_display_boost, uhdr_raw_image_t* dest);

/* ---- handing a context to its next user (round 6; what the facade's context pool calls when a codec lets go of one) --------------------
 * Forgets everything that belonged to the previous user: the device-resident image copies, a write-back error that was latched for
 * "the next call" (uhdr_hip_ctx::sticky), the entropy decoder's size hint, the counters of uhdr_hip_get_stats.  Device scratch,
 * coefficient buffers and the pinned staging ring are kept for the next call only up to keep_bytes in total (largest buffers freed
 * first; 0 frees them all): one 16K encode must not pin gigabytes of HBM for the life of the process.  Returns the device the context
 * is bound to (a pool hands a context only to a caller on that device), -1 for a null context. */
int uhdr_hip_recycle(uhdr_hip_ctx_t* ctx, size_t keep_bytes);
int uhdr_hip_current_device(void); /* the calling thread's current device (what uhdr_hip_create(-1) would bind to); -1 without one */

/* ---- which route did the entropy stage take? ---------------------------------------------------------------------
 * Counters of the context since its creation.  A scan the device declines (entropy_decode_declined: a marker-less stream so
 * dense that the parallel decoder does not settle, or Huffman tables outside its two-level form) is returned to the caller
 * with UHDR_CODEC_UNSUPPORTED_FEATURE -- behind the libuhdr.so facade libjpeg then decodes it on the CPU; this is the place
 * where that shows without a trace. */
typedef struct uhdr_hip_stats {
  unsigned long long entropy_decode_parallel;     /* self-synchronising parallel decode: marker-less scans, restart files with long intervals */
  unsigned long long entropy_decode_intervals;    /* restart-interval scans decoded one lane per interval */
  unsigned long long entropy_decode_single_lane;  /* marker-less scans small enough for one lane */
  unsigned long long entropy_decode_declined;     /* handed back to the caller (see above) */
  unsigned long long entropy_encode_stream;       /* marker-less scans written (the reference's bytes) */
  unsigned long long entropy_encode_intervals;    /* restart-interval scans written */
  unsigned long long resident_hits;               /* host images found on the device (uhdr_hip_resident_begin) instead of uploaded */
  /* two-pass generateGainMap (round 4): pass 2 maps gain ratios to bytes through per-channel step tables built on the device
   * from the exact per-sample evaluation; a channel whose range admits no such table (narrower than ~4e-4 log2 units, not
   * monotone) -- or a user gamma != 1 -- is evaluated per sample instead.  Both give the reference's bytes; this counts which ran. */
  unsigned long long generate_channels_tabled;    /* channels of two-pass calls mapped through a step table */
  unsigned long long generate_channels_per_sample;/* channels of two-pass calls evaluated per sample */
  /* lazy downloads (uhdr_hip_resident_lazy) */
  unsigned long long lazy_downloads_skipped;      /* decoded images left on the device instead of written to the caller's planes */
  unsigned long long lazy_downloads_done;         /* of those, written back / copied out after all (_flush, _materialize, slot reuse) */
  /* wall-clock time the library itself spent in its most recent call of these two entry points, in nanoseconds (round 5): what a
   * C / C++ caller such as the facade pays, without the overhead of whatever binding drives the library */
  unsigned long long last_jpeg_decode_scan_ns;
  unsigned long long last_encode_api1_scans_ns;
} uhdr_hip_stats_t;
void uhdr_hip_get_stats(uhdr_hip_ctx_t* ctx, uhdr_hip_stats_t* out);

/* ---- where did the stages of a drop-in call run? (round 6) -----------------------------------------------------------
 * Process-wide tallies, one row per stage name, kept by the library for whatever sits above it: the libuhdr.so facade reports
 * every stage of an accelerated uhdr_encode / uhdr_decode here (facade/uhdr_hip_seam.cpp) -- on the device, or handed back to
 * the reference's CPU code -- and the whole accelerated call as the stage "uhdr_call".  Tests, bench.py and applications read
 * the table instead of parsing a stderr trace.  Thread-safe; rows come back in the order the stages were first seen since the
 * last reset (first_seq = that position, counted over all notes).  UHDR_HIP_SEAM_STATS_FILE=<path> in the environment: the table
 * is also written there as JSON when the process exits (for a process one cannot call into, e.g. the reference's ultrahdr_app). */
typedef struct uhdr_hip_seam_stage {
  char name[40];
  unsigned long long device_calls;     /* the device produced the stage's result */
  unsigned long long reference_calls;  /* left to the reference's CPU code */
  unsigned long long first_seq;
  double device_ms;                    /* wall time of the device calls, summed */
  double last_ms;                      /* wall time of the most recent call (either kind) */
} uhdr_hip_seam_stage_t;
void uhdr_hip_seam_note(const char* stage, int on_device, double ms);
int uhdr_hip_seam_stats(uhdr_hip_seam_stage_t* out, int capacity); /* -> number of rows (may exceed capacity) */
void uhdr_hip_seam_stats_reset(void);

/* ---- timing hook for bench.py ------------------------------------------------------------------
 * HIP events recorded on the context's stream around every kernel launch of the named family
 * ("apply_gainmap", "generate_gainmap", ...) since the last reset; returns the number of launches
 * and their summed duration in milliseconds (synchronises the stream). */
void uhdr_hip_profile_enable(uhdr_hip_ctx_t* ctx, int enable);
int uhdr_hip_profile_read(uhdr_hip_ctx_t* ctx, const char* family, double* total_ms, int reset);
/* the same, launch by launch: the first min(n, capacity) durations go to ms[] in launch order; returns n */
int uhdr_hip_profile_read_list(uhdr_hip_ctx_t* ctx, const char* family, double* ms, int capacity, int reset);
/* Section marker for external profilers (round 5): launches the empty kernel `uhdr_profile_mark_kernel` on the context's stream.
 * A rocprofv3 kernel trace / counter collection of a process that runs several cases is cut at these launches, so that durations
 * and HBM counters can be reported per (kernel, case) instead of per kernel name (tools/qbench.py, tools/read_prof.py). */
void uhdr_hip_profile_mark(uhdr_hip_ctx_t* ctx);

#ifdef __cplusplus
}
#endif
#endif /* UHDR_HIP_H */
