/* TEST INFRASTRUCTURE ONLY -- CPU oracle ("port") for the gain-map hot path.
 *
 * A plain-C restatement of what the reference computes on the hot path, written from the
 * reference's behaviour (NOT copied): every function cites the reference file:line it follows
 * (paths relative to /root/reference).  It is pinned three ways by tests/: (1) against the
 * known-answer vectors of the reference's own tests/gainmapmath_test.cpp, (2) against the real
 * reference compiled into oracle/_ref (bit-for-bit on random images), (3) against committed golden
 * fixtures generated from (2).  Parity status: PINNED (see DESIGN.md "Oracle").
 *
 * Arithmetic model = the reference as built by its own flags on x86-64 (-O3 -march=x86-64
 * -ffp-contract=fast => SSE2 scalar float, no FMA).  One subtlety is replicated deliberately:
 * gainmapmath.cpp calls pow/log/log2/exp/exp2 *unqualified* inside namespace ultrahdr without a
 * using-directive, so overload resolution picks the C library's DOUBLE functions (verified by
 * disassembling the built object: it imports pow, log, log2, exp, exp2 -- and powf only where the
 * source says std::pow / powf).  jpegr.cpp has `using namespace std`, so its own log2/exp2 calls
 * on floats are log2f/exp2f.  Each site below states which one applies.
 *
 * Build: oracle/Makefile (`make port`), -ffp-contract=off -march=x86-64.
 */
#include "uhdr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float r, g, b; } color_t; /* Color, gainmapmath.h:53-66 (r,g,b aliases y,u,v) */

static const float kSdrWhiteNits = 203.0f; /* gainmapmath.h:44 */
static const float kHlgMaxNits = 1000.0f;  /* gainmapmath.h:46 */
static const float kPqMaxNits = 10000.0f;  /* gainmapmath.h:48 */
static const float kSdrOffset = 1e-7f, kHdrOffset = 1e-7f; /* gainmapmath.h:549-550 */
#define K_MAX_LINEAR (10000.0f / 203.0f)   /* gainmapmath.h:570 */

/* ---------------------------------------------------------------------------------------------
 * colour constants: gainmapmath.cpp:86-105, 156-175, 187-227.  Derived ones are float
 * expressions evaluated left to right, exactly as the static initialisers there.
 * ------------------------------------------------------------------------------------------- */
static const float kSrgbR = 0.212639f, kSrgbG = 0.715169f, kSrgbB = 0.072192f;
static const float kP3R = 0.2289746f, kP3G = 0.6917385f, kP3B = 0.0792869f;
static const float kP3YR = 0.299f, kP3YG = 0.587f, kP3YB = 0.114f;
static const float kP3Cb = 1.772f, kP3Cr = 1.402f;
static const float kBt2100R = 0.2627f, kBt2100G = 0.677998f, kBt2100B = 0.059302f;

typedef struct { float cr, gcb, gcr, cb; } yuv2rgb_t;
typedef struct { float yr, yg, yb, cb, cr; } rgb2yuv_t;

static yuv2rgb_t yuv2rgb_coeffs(int cg) {
  yuv2rgb_t k;
  if (cg == UO_CG_709) { /* gainmapmath.cpp:94,104-105 */
    float cb = 2 * (1 - kSrgbB), cr = 2 * (1 - kSrgbR);
    k.cb = cb; k.cr = cr;
    k.gcb = kSrgbB * cb / kSrgbG;
    k.gcr = kSrgbR * cr / kSrgbG;
  } else if (cg == UO_CG_P3) { /* gainmapmath.cpp:164,174-175 */
    k.cb = kP3Cb; k.cr = kP3Cr;
    k.gcb = kP3YB * kP3Cb / kP3YG;
    k.gcr = kP3YR * kP3Cr / kP3YG;
  } else { /* gainmapmath.cpp:194,226-227 */
    float cb = 2 * (1 - kBt2100B), cr = 2 * (1 - kBt2100R);
    k.cb = cb; k.cr = cr;
    k.gcb = kBt2100B * cb / kBt2100G;
    k.gcr = kBt2100R * cr / kBt2100G;
  }
  return k;
}

static rgb2yuv_t rgb2yuv_coeffs(int cg) {
  rgb2yuv_t k;
  if (cg == UO_CG_709) {
    k.yr = kSrgbR; k.yg = kSrgbG; k.yb = kSrgbB;
    k.cb = 2 * (1 - kSrgbB); k.cr = 2 * (1 - kSrgbR);
  } else if (cg == UO_CG_P3) {
    k.yr = kP3YR; k.yg = kP3YG; k.yb = kP3YB; k.cb = kP3Cb; k.cr = kP3Cr;
  } else {
    k.yr = kBt2100R; k.yg = kBt2100G; k.yb = kBt2100B;
    k.cb = 2 * (1 - kBt2100B); k.cr = 2 * (1 - kBt2100R);
  }
  return k;
}

static float clamp01(float v) { return (v < 0.0f) ? 0.0f : (v > 1.0f) ? 1.0f : v; } /* gainmapmath.h:561 */
static float clip_neg(float v) { return (v < 0.0f) ? 0.0f : v; }                     /* gainmapmath.h:552 */
static float clamp_linear(float v) {                                                 /* gainmapmath.h:572 */
  return (v < 0.0f) ? 0.0f : (v > K_MAX_LINEAR) ? K_MAX_LINEAR : v;
}

/* srgbYuvToRgb / p3YuvToRgb / bt2100YuvToRgb: gainmapmath.cpp:107-111, 177-181, 229-233 */
static color_t yuv_to_rgb(color_t e, const yuv2rgb_t* k) {
  color_t o;
  o.r = clamp01(e.r + k->cr * e.b);
  o.g = clamp01(e.r - k->gcb * e.g - k->gcr * e.b);
  o.b = clamp01(e.r + k->cb * e.g);
  return o;
}
/* srgbRgbToYuv / p3RgbToYuv / bt2100RgbToYuv: gainmapmath.cpp:96-99, 166-169, 196-199 */
static color_t rgb_to_yuv(color_t e, const rgb2yuv_t* k) {
  float y = k->yr * e.r + k->yg * e.g + k->yb * e.b;
  color_t o = {y, (e.b - y) / k->cb, (e.r - y) / k->cr};
  return o;
}
/* luminance: gainmapmath.cpp:88, 158, 189 */
static float luminance(color_t e, int cg) {
  if (cg == UO_CG_709) return kSrgbR * e.r + kSrgbG * e.g + kSrgbB * e.b;
  if (cg == UO_CG_P3) return kP3R * e.r + kP3G * e.g + kP3B * e.b;
  return kBt2100R * e.r + kBt2100G * e.g + kBt2100B * e.b;
}

/* gamut 3x3: gainmapmath.cpp:603-621 */
static const float kBt709ToP3[9] = {0.822462f, 0.177537f, 0.000001f, 0.033194f, 0.966807f,
                                    -0.000001f, 0.017083f, 0.072398f, 0.91052f};
static const float kBt709ToBt2100[9] = {0.627404f, 0.329282f, 0.043314f, 0.069097f, 0.919541f,
                                        0.011362f, 0.016392f, 0.088013f, 0.895595f};
static const float kP3ToBt709[9] = {1.22494f, -0.22494f, 0.0f, -0.042057f, 1.042057f,
                                    0.0f, -0.019638f, -0.078636f, 1.098274f};
static const float kP3ToBt2100[9] = {0.753833f, 0.198597f, 0.04757f, 0.045744f, 0.941777f,
                                     0.012479f, -0.00121f, 0.017601f, 0.983608f};
static const float kBt2100ToBt709[9] = {1.660491f, -0.587641f, -0.07285f, -0.124551f, 1.1329f,
                                        -0.008349f, -0.018151f, -0.100579f, 1.11873f};
static const float kBt2100ToP3[9] = {1.343578f, -0.282179f, -0.061399f, -0.065298f, 1.075788f,
                                     -0.01049f, 0.002822f, -0.019598f, 1.016777f};
/* getGamutConversionFn(dst, src): gainmapmath.cpp:1087-1129; NULL = identity, (void*)-1 = none */
static const float* gamut_matrix(int dst, int src, int* ok) {
  *ok = 1;
  if (dst < 0 || dst > 2 || src < 0 || src > 2) { *ok = 0; return NULL; }
  if (dst == src) return NULL;
  if (dst == UO_CG_709) return src == UO_CG_P3 ? kP3ToBt709 : kBt2100ToBt709;
  if (dst == UO_CG_P3) return src == UO_CG_709 ? kBt709ToP3 : kBt2100ToP3;
  return src == UO_CG_709 ? kBt709ToBt2100 : kP3ToBt2100;
}
static color_t gamut_conv(color_t e, const float* m) {
  if (!m) return e;
  color_t o = {m[0] * e.r + m[1] * e.g + m[2] * e.b, m[3] * e.r + m[4] * e.g + m[5] * e.b,
               m[6] * e.r + m[7] * e.g + m[8] * e.b};
  return o;
}

/* YUV-encoding 3x3: gainmapmath.cpp:638-674 */
static const float kYuv709To601[9] = {1.0f, 0.101579f, 0.196076f, 0.0f, 0.989854f, -0.110653f, 0.0f, -0.072453f, 0.983398f};
static const float kYuv709To2100[9] = {1.0f, -0.016969f, 0.096312f, 0.0f, 0.995306f, -0.051192f, 0.0f, 0.011507f, 1.002637f};
static const float kYuv601To709[9] = {1.0f, -0.118188f, -0.212685f, 0.0f, 1.018640f, 0.114618f, 0.0f, 0.075049f, 1.025327f};
static const float kYuv601To2100[9] = {1.0f, -0.128245f, -0.115879, 0.0f, 1.010016f, 0.061592f, 0.0f, 0.086969f, 1.029350f};
static const float kYuv2100To709[9] = {1.0f, 0.018149f, -0.095132f, 0.0f, 1.004123f, 0.051267f, 0.0f, -0.011524f, 0.996782f};
static const float kYuv2100To601[9] = {1.0f, 0.117887f, 0.105521f, 0.0f, 0.995211f, -0.059549f, 0.0f, -0.084085f, 0.976518f};

/* ---------------------------------------------------------------------------------------------
 * transfer functions.  DOUBLE libm where the reference's unqualified call resolves to it.
 * ------------------------------------------------------------------------------------------- */
/* gainmapmath.cpp:114-120: pow() is the double one; result narrowed on return. */
static float srgb_inv_oetf(float e) {
  if (e <= 0.04045f) return e / 12.92f;
  return (float)pow((double)((e + 0.055f) / 1.055f), (double)2.4f);
}
/* gainmapmath.cpp:139-148: std::pow(float,float) = powf */
static float srgb_oetf(float e) {
  if (e <= 0.0031308f) return 12.92f * e;
  return (1.0f + 0.055f) * powf(e, 1.0f / 2.4f) - 0.055f;
}
static const float kHlgA = 0.17883277f, kHlgB = 0.28466892f, kHlgC = 0.55991073f;
/* gainmapmath.cpp:238-244: sqrt(float)->float rounding == sqrtf; log() is double and drags the
 * whole a*log(..)+c expression to double. */
static float hlg_oetf(float e) {
  if (e <= 1.0f / 12.0f) return sqrtf(3.0f * e);
  return (float)((double)kHlgA * log((double)(12.0f * e - kHlgB)) + (double)kHlgC);
}
/* gainmapmath.cpp:259-265: pow(x, 2.0f) is double (gcc folds it to x*x in double); exp() double */
static float hlg_inv_oetf(float e) {
  if (e <= 0.5f) return (float)(((double)e * (double)e) / (double)3.0f);
  return (float)((exp((double)((e - kHlgC) / kHlgA)) + (double)kHlgB) / (double)12.0f);
}
/* gainmapmath.cpp:309-316: everything double */
static float pq_oetf(float e) {
  const float m1 = 2610.0f / 16384.0f, m2 = 2523.0f / 4096.0f * 128.0f;
  const float c1 = 3424.0f / 4096.0f, c2 = 2413.0f / 4096.0f * 32.0f, c3 = 2392.0f / 4096.0f * 32.0f;
  if (e <= 0.0f) return 0.0f;
  double p = pow((double)e, (double)m1);
  return (float)pow(((double)c1 + (double)c2 * p) / (1 + (double)c3 * p), (double)m2);
}
/* gainmapmath.cpp:330-333: val narrowed to float between the two double pow()s */
static float pq_inv_oetf(float e) {
  const float m1 = 2610.0f / 16384.0f, m2 = 2523.0f / 4096.0f * 128.0f;
  const float c1 = 3424.0f / 4096.0f, c2 = 2413.0f / 4096.0f * 32.0f, c3 = 2392.0f / 4096.0f * 32.0f;
  float val = (float)pow((double)e, (double)(1 / m2));
  float num = val - c1;
  if (!(num > 0.0f)) num = 0.0f; /* (std::max)(val - c1, 0.0f) */
  return (float)pow((double)(num / (c2 - c3 * val)), (double)(1 / m1));
}

/* LookUpTable: gainmapmath.h:345-357 -- entry i = f((float)i / (float)(N-1)) */
#define N_SRGB 1024
#define N_INV 4096
#define N_OETF 65536
static float g_lut_srgb[N_SRGB], g_lut_hlg_inv[N_INV], g_lut_pq_inv[N_INV];
static float g_lut_hlg[N_OETF], g_lut_pq[N_OETF];
static int g_luts_ready = 0;
static void init_luts(void) {
  if (g_luts_ready) return;
  for (int i = 0; i < N_SRGB; i++) g_lut_srgb[i] = srgb_inv_oetf((float)i / (float)(N_SRGB - 1));
  for (int i = 0; i < N_INV; i++) {
    g_lut_hlg_inv[i] = hlg_inv_oetf((float)i / (float)(N_INV - 1));
    g_lut_pq_inv[i] = pq_inv_oetf((float)i / (float)(N_INV - 1));
  }
  for (int i = 0; i < N_OETF; i++) {
    g_lut_hlg[i] = hlg_oetf((float)i / (float)(N_OETF - 1));
    g_lut_pq[i] = pq_oetf((float)i / (float)(N_OETF - 1));
  }
  __sync_synchronize();
  g_luts_ready = 1;
}
/* index rule shared by all LUTs: gainmapmath.cpp:127-129 etc.  float product, double +0.5,
 * truncate, clip. */
static inline int lut_index(float x, int n) {
  int v = (int)((double)(x * (float)(n - 1)) + 0.5);
  return v < 0 ? 0 : (v > n - 1 ? n - 1 : v);
}
static inline float srgb_inv_oetf_lut(float e) { return g_lut_srgb[lut_index(e, N_SRGB)]; }

/* hlgOotfApprox / hlgInverseOotfApprox: gainmapmath.cpp:293-306, std::pow => powf */
static color_t hlg_ootf_approx(color_t e) {
  color_t o = {powf(e.r, 1.2f), powf(e.g, 1.2f), powf(e.b, 1.2f)};
  return o;
}
static color_t hlg_inv_ootf_approx(color_t e) {
  color_t o = {powf(e.r, 1.0f / 1.2f), powf(e.g, 1.0f / 1.2f), powf(e.b, 1.0f / 1.2f)};
  return o;
}

static color_t inv_oetf(color_t e, int ct) { /* getInverseOetfFn: gainmapmath.cpp:1159-1185 */
  const float* t; int n;
  switch (ct) {
    case UO_CT_HLG: t = g_lut_hlg_inv; n = N_INV; break;
    case UO_CT_PQ: t = g_lut_pq_inv; n = N_INV; break;
    case UO_CT_SRGB: t = g_lut_srgb; n = N_SRGB; break;
    default: return e;
  }
  color_t o = {t[lut_index(e.r, n)], t[lut_index(e.g, n)], t[lut_index(e.b, n)]};
  return o;
}
static float ref_peak_nits(int ct) { /* gainmapmath.cpp:20-34 */
  switch (ct) {
    case UO_CT_LINEAR: return kPqMaxNits;
    case UO_CT_HLG: return kHlgMaxNits;
    case UO_CT_PQ: return kPqMaxNits;
    case UO_CT_SRGB: return kSdrWhiteNits;
  }
  return -1.0f;
}

/* ---------------------------------------------------------------------------------------------
 * half floats: gainmapmath.h:160-216
 * ------------------------------------------------------------------------------------------- */
static uint16_t float_to_half(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t b = u + 0x00001000u;
  const int32_t e = (int32_t)((b & 0x7F800000u) >> 23);
  const uint32_t m = b & 0x007FFFFFu;
  uint32_t out = (b & 0x80000000u) >> 16;
  if (e > 112) out |= (((uint32_t)(e - 112) << 10) & 0x7C00u) | (m >> 13);
  if (e < 113 && e > 101) out |= (((0x007FF000u + m) >> (125 - e)) + 1) >> 1;
  if (e > 143) out |= 0x7FFFu;
  return (uint16_t)out;
}
static float half_to_float(uint16_t h) {
  uint32_t mant = h & 0x3ffu, ex = (h >> 10) & 0x1fu, sign = h >> 15, o;
  float f;
  if (ex == 0) {
    const uint32_t magic = 126u << 23;
    float mf, of;
    o = magic + mant;
    memcpy(&of, &o, 4); memcpy(&mf, &magic, 4);
    of -= mf;
    memcpy(&o, &of, 4);
  } else {
    o = mant << 13;
    o |= (ex == 0x1f) ? (255u << 23) : ((127 - 15 + ex) << 23);
  }
  o |= sign << 31;
  memcpy(&f, &o, 4);
  return f;
}
/* sanitizePixel: gainmapmath.h:580-593 */
static float sanitize(float v) {
  if (isfinite(v)) return clamp_linear(v);
  if (isinf(v)) return v > 0 ? K_MAX_LINEAR : 0.0f;
  return 0.0f;
}
/* colorToRgba1010102 / colorToRgbaF16: gainmapmath.cpp:1279-1289 */
static uint32_t to_1010102(color_t e) {
  float r = e.r * 1023 + 0.5f, g = e.g * 1023 + 0.5f, b = e.b * 1023 + 0.5f;
  uint32_t ri = (uint32_t)(r < 0.0f ? 0.0f : (r > 1023.0f ? 1023.0f : r));
  uint32_t gi = (uint32_t)(g < 0.0f ? 0.0f : (g > 1023.0f ? 1023.0f : g));
  uint32_t bi = (uint32_t)(b < 0.0f ? 0.0f : (b > 1023.0f ? 1023.0f : b));
  return ri | (gi << 10) | (bi << 20) | (0x3u << 30);
}
static uint64_t to_f16(color_t e) {
  return (uint64_t)float_to_half(e.r) | ((uint64_t)float_to_half(e.g) << 16) |
         ((uint64_t)float_to_half(e.b) << 32) | ((uint64_t)float_to_half(1.0f) << 48);
}

/* ---------------------------------------------------------------------------------------------
 * pixel fetch: gainmapmath.cpp:354-492
 * ------------------------------------------------------------------------------------------- */
static color_t get_pixel(const uo_image_t* im, size_t x, size_t y) {
  color_t c = {0, 0, 0};
  switch (im->fmt) {
    case UO_FMT_YUV444: case UO_FMT_YUV422: case UO_FMT_YUV420: {
      int hf = im->fmt == UO_FMT_YUV444 ? 1 : 2, vf = im->fmt == UO_FMT_YUV420 ? 2 : 1;
      const uint8_t* yp = (const uint8_t*)im->planes[0];
      const uint8_t* up = (const uint8_t*)im->planes[1];
      const uint8_t* vp = (const uint8_t*)im->planes[2];
      uint8_t yy = yp[x + y * im->stride[0]];
      uint8_t uu = up[x / hf + (y / vf) * im->stride[1]];
      uint8_t vv = vp[x / hf + (y / vf) * im->stride[2]];
      c.r = (float)yy * (1 / 255.0f);
      c.g = (float)(uu - 128) * (1 / 255.0f);
      c.b = (float)(vv - 128) * (1 / 255.0f);
      return c;
    }
    case UO_FMT_Y400: {
      const uint8_t* yp = (const uint8_t*)im->planes[0];
      c.r = (float)yp[x + y * im->stride[0]] * (1 / 255.0f);
      return c;
    }
    case UO_FMT_P010: case UO_FMT_YUV444_10: {
      uint16_t yy, uu, vv;
      if (im->fmt == UO_FMT_P010) {
        const uint16_t* yp = (const uint16_t*)im->planes[0];
        const uint16_t* cp = (const uint16_t*)im->planes[1];
        size_t ui = (y >> 1) * im->stride[1] + (x & ~(size_t)1);
        yy = yp[y * im->stride[0] + x] >> 6;
        uu = cp[ui] >> 6;
        vv = cp[ui + 1] >> 6;
      } else {
        yy = ((const uint16_t*)im->planes[0])[y * im->stride[0] + x];
        uu = ((const uint16_t*)im->planes[1])[y * im->stride[1] + x];
        vv = ((const uint16_t*)im->planes[2])[y * im->stride[2] + x];
      }
      if (im->range == UO_CR_FULL) {
        c.r = (float)yy / 1023.0f;
        c.g = (float)uu / 1023.0f - 0.5f;
        c.b = (float)vv / 1023.0f - 0.5f;
      } else {
        c.r = (float)(yy - 64) * (1 / 876.0f);
        c.g = (float)(uu - 64) * (1 / 896.0f) - 0.5f;
        c.b = (float)(vv - 64) * (1 / 896.0f) - 0.5f;
      }
      return c;
    }
    case UO_FMT_RGB888: {
      const uint8_t* p = (const uint8_t*)im->planes[0] + x * 3 + y * (size_t)im->stride[0] * 3;
      c.r = (float)p[0] / 255.0f; c.g = (float)p[1] / 255.0f; c.b = (float)p[2] / 255.0f;
      return c;
    }
    case UO_FMT_RGBA8888: {
      uint32_t v = ((const uint32_t*)im->planes[0])[x + y * (size_t)im->stride[0]];
      c.r = (float)(v & 0xff) / 255.0f;
      c.g = (float)((v >> 8) & 0xff) / 255.0f;
      c.b = (float)((v >> 16) & 0xff) / 255.0f;
      return c;
    }
    case UO_FMT_RGBA1010102: {
      uint32_t v = ((const uint32_t*)im->planes[0])[x + y * (size_t)im->stride[0]];
      c.r = (float)(v & 0x3ff) / 1023.0f;
      c.g = (float)((v >> 10) & 0x3ff) / 1023.0f;
      c.b = (float)((v >> 20) & 0x3ff) / 1023.0f;
      return c;
    }
    case UO_FMT_RGBAF16: {
      uint64_t v = ((const uint64_t*)im->planes[0])[x + y * (size_t)im->stride[0]];
      c.r = sanitize(half_to_float((uint16_t)(v & 0xffff)));
      c.g = sanitize(half_to_float((uint16_t)((v >> 16) & 0xffff)));
      c.b = sanitize(half_to_float((uint16_t)((v >> 32) & 0xffff)));
      return c;
    }
  }
  return c;
}
/* samplePixels: gainmapmath.cpp:494-504 (dy outer, dx inner, then one divide per channel) */
static color_t sample_pixels(const uo_image_t* im, size_t s, size_t x, size_t y) {
  color_t e = {0.0f, 0.0f, 0.0f};
  for (size_t dy = 0; dy < s; ++dy)
    for (size_t dx = 0; dx < s; ++dx) {
      color_t p = get_pixel(im, x * s + dx, y * s + dy);
      e.r += p.r; e.g += p.g; e.b += p.b;
    }
  float d = (float)(s * s);
  e.r /= d; e.g /= d; e.b /= d;
  return e;
}
static int is_rgb_fmt(int fmt) { /* isPixelFormatRgb: gainmapmath.cpp:1274-1277 */
  return fmt == UO_FMT_RGBAF16 || fmt == UO_FMT_RGBA8888 || fmt == UO_FMT_RGBA1010102;
}

/* ---------------------------------------------------------------------------------------------
 * gain encode / apply: gainmapmath.cpp:753-855, gainmapmath.h:452-495
 * ------------------------------------------------------------------------------------------- */
/* encodeGain: log2() double; the normalisation is double arithmetic narrowed to float; powf */
static uint8_t encode_gain(float y_sdr, float y_hdr, float min_boost, float max_boost,
                           float gamma, float log2min, float log2max) {
  float gain = 1.0f;
  if (y_sdr > 0.0f) gain = y_hdr / y_sdr;
  if (gain < min_boost) gain = min_boost;
  if (gain > max_boost) gain = max_boost;
  float n = (float)((log2((double)gain) - (double)log2min) / (double)(log2max - log2min));
  float ng = powf(n, gamma);
  return (uint8_t)(ng * 255.0f);
}
/* computeGain: gainmapmath.cpp:773-782 (log2 double, narrowed) */
static float compute_gain(float sdr, float hdr) {
  float gain = (float)log2((double)((hdr + kHdrOffset) / (sdr + kSdrOffset)));
  if (sdr < 2.f / 255.0f) gain = gain < 2.3f ? gain : 2.3f; /* (std::min)(gain, 2.3f) */
  return gain;
}
/* affineMapGain: gainmapmath.cpp:784-789 (pow double; CLIP3 in float; truncating return) */
static uint8_t affine_map_gain(float g, float mn, float mx, float gamma) {
  float m = (g - mn) / (mx - mn);
  if (gamma != 1.0f) m = (float)pow((double)m, (double)gamma);
  m *= 255;
  float t = m + 0.5f;
  t = (t < 0) ? 0 : ((t > 255) ? 255 : t);
  return (uint8_t)t;
}
/* GainLUT: gainmapmath.h:452-495.  log2/exp2 are the double ones (header code, no using-directive);
 * logBoost is narrowed to float; logBoost*weight is a float product fed to double exp2. */
typedef struct { float tab[3][1024]; float gamma_inv[3]; int single; } gain_lut_t;
static int md_identical(const uo_metadata_t* m) { /* ultrahdrcommon.h:218-226 */
  return m->max_content_boost[0] == m->max_content_boost[1] && m->max_content_boost[0] == m->max_content_boost[2] &&
         m->min_content_boost[0] == m->min_content_boost[1] && m->min_content_boost[0] == m->min_content_boost[2] &&
         m->gamma[0] == m->gamma[1] && m->gamma[0] == m->gamma[2] &&
         m->offset_sdr[0] == m->offset_sdr[1] && m->offset_sdr[0] == m->offset_sdr[2] &&
         m->offset_hdr[0] == m->offset_hdr[1] && m->offset_hdr[0] == m->offset_hdr[2];
}
static void gain_lut_init(gain_lut_t* l, const uo_metadata_t* md, float weight) {
  l->single = md_identical(md);
  for (int i = 0; i < (l->single ? 1 : 3); i++) {
    l->gamma_inv[i] = 1.0f / md->gamma[i];
    for (int idx = 0; idx < 1024; idx++) {
      float value = (float)idx / (float)1023;
      float log_boost = (float)(log2((double)md->min_content_boost[i]) * (double)(1.0f - value) +
                                log2((double)md->max_content_boost[i]) * (double)value);
      l->tab[i][idx] = (float)exp2((double)(log_boost * weight));
    }
  }
  if (l->single) {
    l->gamma_inv[1] = l->gamma_inv[2] = l->gamma_inv[0];
    memcpy(l->tab[1], l->tab[0], sizeof l->tab[0]);
    memcpy(l->tab[2], l->tab[0], sizeof l->tab[0]);
  }
}
static float gain_factor(const gain_lut_t* l, float gain, int ch) {
  if (l->gamma_inv[ch] != 1.0f) gain = (float)pow((double)gain, (double)l->gamma_inv[ch]);
  return l->tab[ch][lut_index(gain, 1024)];
}
/* exact applyGain with weight: gainmapmath.cpp:799-805 / 830-846 (all-double log/exp) */
static float apply_gain_exact_factor(float gain, const uo_metadata_t* md, int ch, float weight) {
  if (md->gamma[ch] != 1.0f) gain = (float)pow((double)gain, (double)(1.0f / md->gamma[ch]));
  float log_boost = (float)(log2((double)md->min_content_boost[ch]) * (double)(1.0f - gain) +
                            log2((double)md->max_content_boost[ch]) * (double)gain);
  return (float)exp2((double)(log_boost * weight));
}

/* ShepardsIDW::fillShepardsIDW: gainmapmath.cpp:39-80 (sqrtf; float arithmetic) */
static void fill_idw(float* w, int s, int inc_r, int inc_b) {
  for (int y = 0; y < s; y++)
    for (int x = 0; x < s; x++) {
      float px = ((float)x) / s, py = ((float)y) / s;
      int cx = (int)floorf(px), cy = (int)floorf(py);
      int nx = cx + inc_r, ny = cy + inc_b;
      int idx = y * s * 4 + x * 4;
#define DIST(x1, x2, y1, y2) sqrtf((((y2) - (y1)) * ((y2) - (y1))) + ((x2) - (x1)) * ((x2) - (x1)))
      float d1 = DIST(px, (float)cx, py, (float)cy);
      if (d1 == 0) {
        w[idx] = 1.f; w[idx + 1] = 0.f; w[idx + 2] = 0.f; w[idx + 3] = 0.f;
      } else {
        float w1 = 1.f / d1;
        float w2 = 1.f / DIST(px, (float)cx, py, (float)ny);
        float w3 = 1.f / DIST(px, (float)nx, py, (float)cy);
        float w4 = 1.f / DIST(px, (float)nx, py, (float)ny);
        float tot = w1 + w2 + w3 + w4;
        w[idx] = w1 / tot; w[idx + 1] = w2 / tot; w[idx + 2] = w3 / tot; w[idx + 3] = w4 / tot;
      }
#undef DIST
    }
}
typedef struct { int s; float *w, *wnr, *wnb, *wc; } idw_t;
static void idw_init(idw_t* t, int s) {
  size_t n = (size_t)s * s * 4;
  t->s = s;
  t->w = (float*)malloc(4 * n * sizeof(float));
  t->wnr = t->w + n; t->wnb = t->wnr + n; t->wc = t->wnb + n;
  fill_idw(t->w, s, 1, 1); fill_idw(t->wnr, s, 0, 1); fill_idw(t->wnb, s, 1, 0); fill_idw(t->wc, s, 0, 0);
}
static void idw_free(idw_t* t) { free(t->w); }

static inline size_t szmin(size_t a, size_t b) { return a < b ? a : b; }
/* pythDistance: gainmapmath.cpp:866-868 -- pow(x,2.0f) double => x*x in double, sqrt double */
static float pyth(float dx, float dy) {
  return (float)sqrt((double)dx * (double)dx + (double)dy * (double)dy);
}

/* one sampler for the four reference variants (gainmapmath.cpp:871-1080).  ch = 1 (Y400) or 3;
 * bpp = bytes per map pixel (1, 3 or 4).  Returns gain per channel in out[]. */
static void sample_map(const uo_image_t* map, float msf, size_t x, size_t y, const idw_t* idw,
                       int use_table, int nch, int bpp, float out[3]) {
  const uint8_t* data = (const uint8_t*)map->planes[0];
  size_t stride = map->stride[0];
  size_t xl, xu, yl, yu;
  float xm = 0, ym = 0;
  if (use_table) {
    size_t s = (size_t)msf;
    xl = x / s; yl = y / s;
  } else {
    xm = (float)x / msf; ym = (float)y / msf;
    xl = (size_t)floorf(xm); yl = (size_t)floorf(ym);
  }
  xu = xl + 1; yu = yl + 1;
  xl = szmin(xl, map->w - 1); xu = szmin(xu, map->w - 1);
  yl = szmin(yl, map->h - 1); yu = szmin(yu, map->h - 1);
  float e1[3], e2[3], e3[3], e4[3];
  for (int c = 0; c < nch; c++) {
    e1[c] = (float)data[(xl + yl * stride) * bpp + c] / 255.0f;
    e2[c] = (float)data[(xl + yu * stride) * bpp + c] / 255.0f;
    e3[c] = (float)data[(xu + yl * stride) * bpp + c] / 255.0f;
    e4[c] = (float)data[(xu + yu * stride) * bpp + c] / 255.0f;
  }
  if (use_table) {
    size_t s = (size_t)msf;
    const float* w = idw->w;
    if (xl == xu && yl == yu) w = idw->wc;
    else if (xl == xu) w = idw->wnr;
    else if (yl == yu) w = idw->wnb;
    w += (y % s) * s * 4 + (x % s) * 4;
    for (int c = 0; c < nch; c++) out[c] = e1[c] * w[0] + e2[c] * w[1] + e3[c] * w[2] + e4[c] * w[3];
    return;
  }
  float d1 = pyth(xm - (float)xl, ym - (float)yl);
  if (d1 == 0.0f) { for (int c = 0; c < nch; c++) out[c] = e1[c]; return; }
  float d2 = pyth(xm - (float)xl, ym - (float)yu);
  if (d2 == 0.0f) { for (int c = 0; c < nch; c++) out[c] = e2[c]; return; }
  float d3 = pyth(xm - (float)xu, ym - (float)yl);
  if (d3 == 0.0f) { for (int c = 0; c < nch; c++) out[c] = e3[c]; return; }
  float d4 = pyth(xm - (float)xu, ym - (float)yu);
  if (d4 == 0.0f) { /* quirk: the 1-channel sampler returns e2 here (gainmapmath.cpp:908) */
    for (int c = 0; c < nch; c++) out[c] = (nch == 1) ? e2[c] : e4[c];
    return;
  }
  float w1 = 1.0f / d1, w2 = 1.0f / d2, w3 = 1.0f / d3, w4 = 1.0f / d4;
  float tot = w1 + w2 + w3 + w4;
  for (int c = 0; c < nch; c++)
    out[c] = e1[c] * (w1 / tot) + e2[c] * (w2 / tot) + e3[c] * (w3 / tot) + e4[c] * (w4 / tot);
}

static int validate_md(const uo_metadata_t* m) { /* ultrahdr_api.cpp:431-503 */
  if (!m) return UO_INVALID_PARAM;
  for (int i = 0; i < 3; i++) {
    if (!isfinite(m->min_content_boost[i]) || !isfinite(m->max_content_boost[i]) ||
        !isfinite(m->offset_sdr[i]) || !isfinite(m->offset_hdr[i]) || !isfinite(m->hdr_capacity_min) ||
        !isfinite(m->hdr_capacity_max) || !isfinite(m->gamma[i]))
      return UO_INVALID_PARAM;
    if (m->max_content_boost[i] < m->min_content_boost[i]) return UO_INVALID_PARAM;
    if (m->min_content_boost[i] <= 0.0f) return UO_INVALID_PARAM;
    if (m->gamma[i] <= 0.0f) return UO_INVALID_PARAM;
    if (m->offset_sdr[i] < 0.0f || m->offset_hdr[i] < 0.0f) return UO_INVALID_PARAM;
    if (m->hdr_capacity_max <= m->hdr_capacity_min) return UO_INVALID_PARAM;
    if (m->hdr_capacity_min < 1.0f) return UO_INVALID_PARAM;
  }
  return UO_OK;
}

/* ---------------------------------------------------------------------------------------------
 * UltraHdr::applyGainMap: jpegr.cpp:1533-1831
 * ------------------------------------------------------------------------------------------- */
int uo_apply_gainmap(const uo_image_t* sdr, const uo_image_t* gm, const uo_metadata_t* md,
                     int out_ct, int out_fmt, float max_display_boost, uo_image_t* dest) {
  (void)out_fmt;
  init_luts();
  if (!dest || !dest->planes[0]) return UO_INVALID_PARAM;
  if (dest->stride[0] < dest->w) return UO_INVALID_PARAM;
  if (out_ct != UO_CT_LINEAR && out_ct != UO_CT_HLG && out_ct != UO_CT_PQ) return UO_INVALID_PARAM;
  if ((out_ct == UO_CT_LINEAR && dest->fmt != UO_FMT_RGBAF16) ||
      (out_ct != UO_CT_LINEAR && dest->fmt != UO_FMT_RGBA1010102))
    return UO_INVALID_PARAM;
  int st = validate_md(md);
  if (st) return st;
  if (sdr->fmt != UO_FMT_YUV444 && sdr->fmt != UO_FMT_YUV422 && sdr->fmt != UO_FMT_YUV420 &&
      sdr->fmt != UO_FMT_RGB888 && sdr->fmt != UO_FMT_RGBA8888)
    return UO_UNSUPPORTED;
  if (gm->fmt != UO_FMT_Y400 && gm->fmt != UO_FMT_RGB888 && gm->fmt != UO_FMT_RGBA8888)
    return UO_UNSUPPORTED;

  int sdr_cg = sdr->cg == UO_CG_UNSPEC ? UO_CG_709 : sdr->cg;
  int hdr_cg = gm->cg == UO_CG_UNSPEC ? sdr_cg : gm->cg;
  dest->cg = hdr_cg;
  int ok;
  const float* m = gamut_matrix(hdr_cg, sdr_cg, &ok);
  if (!ok) return UO_ERROR;
  const float* hdr_m = md->use_base_cg ? m : NULL;
  const float* sdr_m = md->use_base_cg ? NULL : m;

  { /* aspect-ratio guard: jpegr.cpp:1651-1671.  The resize fallback is outside the hot path. */
    float pa = (float)sdr->w / sdr->h, ga = (float)gm->w / gm->h;
    if (fabsf(pa - ga) / pa > 0.01f) return UO_UNSUPPORTED;
  }
  float msf = (float)sdr->w / gm->w;
  int msf_rnd = (int)roundf(msf);
  if (msf_rnd < 1) msf_rnd = 1;
  idw_t idw;
  idw_init(&idw, msf_rnd);
  float display_boost = max_display_boost < md->hdr_capacity_max ? max_display_boost : md->hdr_capacity_max;
  float weight;
  if (display_boost != md->hdr_capacity_max) { /* log2f here: jpegr.cpp has using namespace std */
    weight = (log2f(display_boost) - log2f(md->hdr_capacity_min)) /
             (log2f(md->hdr_capacity_max) - log2f(md->hdr_capacity_min));
    weight = (weight < 0.0f) ? 0.0f : (weight > 1.0f) ? 1.0f : weight;
  } else {
    weight = 1.0f;
  }
  gain_lut_t* lut = (gain_lut_t*)malloc(sizeof(gain_lut_t));
  gain_lut_init(lut, md, weight);

  const yuv2rgb_t p3 = yuv2rgb_coeffs(UO_CG_P3); /* always BT.601: jpegr.cpp:1723 */
  const int sdr_rgb = is_rgb_fmt(sdr->fmt);      /* NB: RGB888 is not "rgb" here (gainmapmath.cpp:1274) */
  const int use_table = (msf == floorf(msf));
  const int nch = gm->fmt == UO_FMT_Y400 ? 1 : 3;
  const int bpp = gm->fmt == UO_FMT_Y400 ? 1 : (gm->fmt == UO_FMT_RGBA8888 ? 4 : 3);

  for (size_t y = 0; y < sdr->h; ++y) {
    for (size_t x = 0; x < sdr->w; ++x) {
      color_t g = get_pixel(sdr, x, y);
      if (!sdr_rgb) g = yuv_to_rgb(g, &p3);
      color_t lin = {srgb_inv_oetf_lut(g.r), srgb_inv_oetf_lut(g.g), srgb_inv_oetf_lut(g.b)};
      lin = gamut_conv(lin, sdr_m);
      float gain[3];
      sample_map(gm, msf, x, y, &idw, use_table, nch, bpp, gain);
      if (nch == 1) gain[1] = gain[2] = gain[0];
      color_t hdr;
      if (nch == 1) { /* applyGainLUT(Color,float): channel-0 metadata for all: gainmapmath.cpp:807-810 */
        float f = gain_factor(lut, gain[0], 0);
        hdr.r = ((lin.r + md->offset_sdr[0]) * f) - md->offset_hdr[0];
        hdr.g = ((lin.g + md->offset_sdr[0]) * f) - md->offset_hdr[0];
        hdr.b = ((lin.b + md->offset_sdr[0]) * f) - md->offset_hdr[0];
      } else {
        hdr.r = ((lin.r + md->offset_sdr[0]) * gain_factor(lut, gain[0], 0)) - md->offset_hdr[0];
        hdr.g = ((lin.g + md->offset_sdr[1]) * gain_factor(lut, gain[1], 1)) - md->offset_hdr[1];
        hdr.b = ((lin.b + md->offset_sdr[2]) * gain_factor(lut, gain[2], 2)) - md->offset_hdr[2];
      }
      size_t idx = x + y * (size_t)dest->stride[0];
      if (out_ct == UO_CT_LINEAR) {
        hdr = gamut_conv(hdr, hdr_m);
        hdr.r = clamp_linear(hdr.r); hdr.g = clamp_linear(hdr.g); hdr.b = clamp_linear(hdr.b);
        ((uint64_t*)dest->planes[0])[idx] = to_f16(hdr);
      } else {
        float peak = out_ct == UO_CT_HLG ? kHlgMaxNits : kPqMaxNits;
        hdr.r = hdr.r * kSdrWhiteNits / peak; /* two ops: (x*203)/peak */
        hdr.g = hdr.g * kSdrWhiteNits / peak;
        hdr.b = hdr.b * kSdrWhiteNits / peak;
        hdr = gamut_conv(hdr, hdr_m);
        hdr.r = clamp01(hdr.r); hdr.g = clamp01(hdr.g); hdr.b = clamp01(hdr.b);
        const float* t = g_lut_pq;
        if (out_ct == UO_CT_HLG) { hdr = hlg_inv_ootf_approx(hdr); t = g_lut_hlg; }
        color_t o = {t[lut_index(hdr.r, N_OETF)], t[lut_index(hdr.g, N_OETF)], t[lut_index(hdr.b, N_OETF)]};
        ((uint32_t*)dest->planes[0])[idx] = to_1010102(o);
      }
    }
  }
  free(lut);
  idw_free(&idw);
  return UO_OK;
}

/* ---------------------------------------------------------------------------------------------
 * UltraHdr::generateGainMap: jpegr.cpp:530-1058  (built with UHDR_WRITE_ISO only => no XMP merge)
 * ------------------------------------------------------------------------------------------- */
/* ext_gbuf/ext_minmax non-NULL: stop after pass 1 of the two-pass mode and hand out the float
 * log2-gain plane and {min0,min1,min2,max0,max1,max2} (used by the row-stripe tests). */
static int gen_run(const uo_image_t* sdr, const uo_image_t* hdr, const uo_encode_cfg_t* cfg,
                   uo_metadata_t* md, uo_image_t* gm, float* ext_gbuf, float* ext_minmax) {
  init_luts();
  if (sdr->fmt != UO_FMT_YUV444 && sdr->fmt != UO_FMT_YUV422 && sdr->fmt != UO_FMT_YUV420 &&
      sdr->fmt != UO_FMT_RGBA8888)
    return UO_UNSUPPORTED;
  if (hdr->fmt != UO_FMT_P010 && hdr->fmt != UO_FMT_YUV444_10 && hdr->fmt != UO_FMT_RGBA1010102 &&
      hdr->fmt != UO_FMT_RGBAF16)
    return UO_UNSUPPORTED;
  if (hdr->ct < UO_CT_LINEAR || hdr->ct > UO_CT_SRGB) return UO_UNSUPPORTED;
  if (hdr->cg < 0 || hdr->cg > 2 || sdr->cg < 0 || sdr->cg > 2) return UO_UNSUPPORTED;
  float hdr_white_nits = ref_peak_nits(hdr->ct);

  const float *hdr_m = NULL, *sdr_m = NULL;
  int use_sdr_cg = 1, ok;
  if (sdr->cg != hdr->cg) { /* jpegr.cpp:608-637 with kWriteXmpMetadata == false */
    use_sdr_cg = !(hdr->cg == UO_CG_2100 || (hdr->cg == UO_CG_P3 && sdr->cg != UO_CG_2100));
    if (use_sdr_cg) hdr_m = gamut_matrix(sdr->cg, hdr->cg, &ok);
    else sdr_m = gamut_matrix(hdr->cg, sdr->cg, &ok);
  }
  md->use_base_cg = use_sdr_cg;
  yuv2rgb_t sdr_y2r = yuv2rgb_coeffs(cfg->sdr_is_601 ? UO_CG_P3 : sdr->cg);
  yuv2rgb_t hdr_y2r = yuv2rgb_coeffs(hdr->cg);
  const int lum_cg = sdr->cg; /* luminanceFn = getLuminanceFn(sdr_intent->cg), used for BOTH images */

  int scale = cfg->scale;
  unsigned mw = sdr->w / scale, mh = sdr->h / scale;
  if (mw == 0 || mh == 0) { /* jpegr.cpp:696-706 */
    int s = (int)(sdr->w < sdr->h ? sdr->w : sdr->h);
    s = (s >= 8) ? (s / 8) : 1;
    scale = s;
    mw = sdr->w / scale; mh = sdr->h / scale;
  }
  const int multi = cfg->multichannel != 0;
  gm->fmt = multi ? UO_FMT_RGB888 : UO_FMT_Y400;
  gm->cg = hdr->cg; gm->ct = hdr->ct; gm->range = hdr->range;
  gm->w = mw; gm->h = mh;
  if (!ext_gbuf && gm->stride[0] < mw) return UO_INVALID_PARAM;
  uint8_t* out = (uint8_t*)gm->planes[0];
  const size_t ostride = gm->stride[0];

  const int hdr_rgb = is_rgb_fmt(hdr->fmt), sdr_rgb = is_rgb_fmt(sdr->fmt);
  const float hdr_nits = hdr->ct == UO_CT_LINEAR ? kSdrWhiteNits : hdr_white_nits;
  const int two_pass = cfg->preset != UO_PRESET_REALTIME;
  const float gamma = cfg->gamma;

  float log2min = 0, log2max = 0;
  float* gbuf = NULL;
  float gmin[3] = {127.0f, 127.0f, 127.0f}, gmax[3] = {-128.0f, -128.0f, -128.0f};
  if (!two_pass) { /* jpegr.cpp:724-737 */
    for (int i = 0; i < 3; i++) {
      md->max_content_boost[i] = hdr_white_nits / kSdrWhiteNits;
      md->min_content_boost[i] = 1.0f;
      md->gamma[i] = gamma;
      md->offset_sdr[i] = 0.0f;
      md->offset_hdr[i] = 0.0f;
    }
    md->hdr_capacity_min = 1.0f;
    md->hdr_capacity_max = cfg->target_nits != -1.0f ? cfg->target_nits / kSdrWhiteNits : md->max_content_boost[0];
    log2min = log2f(md->min_content_boost[0]);
    log2max = log2f(md->max_content_boost[0]);
  } else {
    gbuf = ext_gbuf ? ext_gbuf : (float*)malloc((size_t)mw * mh * sizeof(float) * (multi ? 3 : 1));
    if (!gbuf) return UO_MEM_ERROR;
  }

  for (size_t y = 0; y < mh; ++y) {
    for (size_t x = 0; x < mw; ++x) {
      color_t s = sample_pixels(sdr, scale, x, y);
      if (!sdr_rgb) s = yuv_to_rgb(s, &sdr_y2r);
      color_t sl = {srgb_inv_oetf_lut(s.r), srgb_inv_oetf_lut(s.g), srgb_inv_oetf_lut(s.b)};
      sl = gamut_conv(sl, sdr_m);
      sl.r = clip_neg(sl.r); sl.g = clip_neg(sl.g); sl.b = clip_neg(sl.b);

      color_t h = sample_pixels(hdr, scale, x, y);
      if (!hdr_rgb) h = yuv_to_rgb(h, &hdr_y2r);
      color_t hl = inv_oetf(h, hdr->ct);
      if (hdr->ct == UO_CT_HLG) hl = hlg_ootf_approx(hl);
      hl = gamut_conv(hl, hdr_m);
      hl.r = clip_neg(hl.r); hl.g = clip_neg(hl.g); hl.b = clip_neg(hl.b);

      if (multi) {
        float sn[3] = {sl.r * kSdrWhiteNits, sl.g * kSdrWhiteNits, sl.b * kSdrWhiteNits};
        float hn[3] = {hl.r * hdr_nits, hl.g * hdr_nits, hl.b * hdr_nits};
        if (!two_pass) {
          size_t idx = (x + y * ostride) * 3;
          for (int c = 0; c < 3; c++)
            out[idx + c] = encode_gain(sn[c], hn[c], md->min_content_boost[c], md->max_content_boost[c],
                                       md->gamma[c], log2min, log2max);
        } else {
          size_t idx = (x + y * mw) * 3;
          for (int c = 0; c < 3; c++) {
            float v = compute_gain(sn[c], hn[c]);
            gbuf[idx + c] = v;
            gmin[c] = v < gmin[c] ? v : gmin[c];
            gmax[c] = gmax[c] < v ? v : gmax[c];
          }
        }
      } else {
        float sy, hy;
        if (cfg->use_luminance) {
          sy = luminance(sl, lum_cg) * kSdrWhiteNits;
          hy = luminance(hl, lum_cg) * hdr_nits;
        } else {
          sy = fmaxf(sl.r, fmaxf(sl.g, sl.b)) * kSdrWhiteNits;
          hy = fmaxf(hl.r, fmaxf(hl.g, hl.b)) * hdr_nits;
        }
        if (!two_pass) {
          out[x + y * ostride] = encode_gain(sy, hy, md->min_content_boost[0], md->max_content_boost[0],
                                             md->gamma[0], log2min, log2max);
        } else {
          float v = compute_gain(sy, hy);
          gbuf[x + y * mw] = v;
          gmin[0] = v < gmin[0] ? v : gmin[0];
          gmax[0] = gmax[0] < v ? v : gmax[0];
        }
      }
    }
  }
  if (!two_pass) return UO_OK;
  if (ext_gbuf) {
    for (int i = 0; i < 3; i++) { ext_minmax[i] = gmin[i]; ext_minmax[3 + i] = gmax[i]; }
    return UO_OK;
  }

  const int nch = multi ? 3 : 1;
  for (int i = 0; i < nch; i++) { /* jpegr.cpp:969-986 */
    gmin[i] = gmin[i] < -14.3f ? -14.3f : (gmin[i] > 15.6f ? 15.6f : gmin[i]);
    gmax[i] = gmax[i] < -14.3f ? -14.3f : (gmax[i] > 15.6f ? 15.6f : gmax[i]);
    if (cfg->max_boost != FLT_MAX) { float s = log2f(cfg->max_boost); gmax[i] = gmax[i] < s ? gmax[i] : s; }
    if (cfg->min_boost != FLT_MIN) { float s = log2f(cfg->min_boost); gmin[i] = gmin[i] < s ? s : gmin[i]; }
    if (fabsf(gmax[i] - gmin[i]) < FLT_EPSILON) gmax[i] += 0.1f;
  }
  for (size_t y = 0; y < mh; ++y) /* jpegr.cpp:992-1013 */
    for (size_t i = 0; i < (size_t)mw * nch; i++)
      out[y * ostride * nch + i] = affine_map_gain(gbuf[y * mw * nch + i], gmin[i % nch], gmax[i % nch], gamma);
  free(gbuf);
  for (int i = 0; i < 3; i++) { /* jpegr.cpp:1031-1048 (exp2f) */
    int k = multi ? i : 0;
    md->max_content_boost[i] = exp2f(gmax[k]);
    md->min_content_boost[i] = exp2f(gmin[k]);
    md->gamma[i] = gamma;
    md->offset_sdr[i] = kSdrOffset;
    md->offset_hdr[i] = kHdrOffset;
  }
  md->hdr_capacity_min = 1.0f;
  md->hdr_capacity_max = cfg->target_nits != -1.0f ? cfg->target_nits / kSdrWhiteNits : hdr_white_nits / kSdrWhiteNits;
  return UO_OK;
}

int uo_generate_gainmap(const uo_image_t* sdr, const uo_image_t* hdr, const uo_encode_cfg_t* cfg,
                        uo_metadata_t* md, uo_image_t* gm) {
  return gen_run(sdr, hdr, cfg, md, gm, NULL, NULL);
}
int uo_generate_gainmap_pass1(const uo_image_t* sdr, const uo_image_t* hdr, const uo_encode_cfg_t* cfg,
                              float* gain_log2, float minmax[6], int* use_base_cg) {
  uo_metadata_t md;
  uo_image_t gm;
  memset(&gm, 0, sizeof gm);
  uo_encode_cfg_t c = *cfg;
  c.preset = UO_PRESET_BEST_QUALITY;
  int rc = gen_run(sdr, hdr, &c, &md, &gm, gain_log2, minmax);
  *use_base_cg = md.use_base_cg;
  return rc;
}
/* pass 2 of jpegr.cpp:992-1013 on an externally reduced min/max (already clamped/finalized) */
void uo_generate_gainmap_pass2(const float* gain_log2, const float minmax[6], float gamma, int nch,
                               unsigned mw, unsigned mh, uint8_t* out, size_t out_stride) {
  for (size_t y = 0; y < mh; ++y)
    for (size_t i = 0; i < (size_t)mw * nch; i++)
      out[y * out_stride * nch + i] =
          affine_map_gain(gain_log2[y * mw * nch + i], minmax[i % nch], minmax[3 + i % nch], gamma);
}

/* ---------------------------------------------------------------------------------------------
 * UltraHdr::toneMap + globalTonemap: jpegr.cpp:1945-2222
 * ------------------------------------------------------------------------------------------- */
static uint8_t scale_to_8bit(float v) { /* jpegr.cpp:1979-1983 */
  int i = (int)roundf(v * 255.0f);
  return (uint8_t)(i < 0 ? 0 : (i > 255 ? 255 : i));
}
static uint8_t put8(float v) { /* put*Pixel: *255, +0.5, clip, truncate (gainmapmath.cpp:538-596) */
  v *= 255.0f; v += 0.5f;
  v = (v < 0.0f) ? 0.0f : (v > 255.0f) ? 255.0f : v;
  return (uint8_t)v;
}
int uo_tone_map(const uo_image_t* hdr, uo_image_t* sdr) {
  init_luts();
  if (hdr->fmt != UO_FMT_P010 && hdr->fmt != UO_FMT_YUV444_10 && hdr->fmt != UO_FMT_RGBA1010102 &&
      hdr->fmt != UO_FMT_RGBAF16)
    return UO_UNSUPPORTED;
  if (hdr->fmt == UO_FMT_P010 && sdr->fmt != UO_FMT_YUV420) return UO_UNSUPPORTED;
  if (hdr->fmt == UO_FMT_YUV444_10 && sdr->fmt != UO_FMT_YUV444) return UO_UNSUPPORTED;
  if ((hdr->fmt == UO_FMT_RGBA1010102 || hdr->fmt == UO_FMT_RGBAF16) && sdr->fmt != UO_FMT_RGBA8888)
    return UO_UNSUPPORTED;
  if (hdr->cg < 0 || hdr->cg > 2) return UO_UNSUPPORTED;
  if (hdr->ct < UO_CT_LINEAR || hdr->ct > UO_CT_SRGB) return UO_UNSUPPORTED;
  float hdr_white_nits = ref_peak_nits(hdr->ct);
  sdr->cg = UO_CG_P3; sdr->ct = UO_CT_SRGB; sdr->range = UO_CR_FULL;
  int ok;
  const float* gm = gamut_matrix(UO_CG_P3, hdr->cg, &ok);
  const yuv2rgb_t y2r = yuv2rgb_coeffs(hdr->cg);
  const rgb2yuv_t p3 = rgb2yuv_coeffs(UO_CG_P3);
  const int f = hdr->fmt == UO_FMT_P010 ? 2 : 1;
  const int hdr_rgb = is_rgb_fmt(hdr->fmt), sdr_rgb = is_rgb_fmt(sdr->fmt);
  const int is_norm = hdr->ct != UO_CT_LINEAR;
  const float headroom = hdr_white_nits / kSdrWhiteNits;
  uint8_t *yp = (uint8_t*)sdr->planes[0], *up = (uint8_t*)sdr->planes[1], *vp = (uint8_t*)sdr->planes[2];

  for (size_t y = 0; y < hdr->h; y += f) {
    for (size_t x = 0; x < hdr->w; x += f) {
      float su = 0.0f, sv = 0.0f;
      for (int i = 0; i < f; i++) {
        for (int j = 0; j < f; j++) {
          color_t g = get_pixel(hdr, x + j, y + i);
          if (!hdr_rgb) g = yuv_to_rgb(g, &y2r);
          color_t l = inv_oetf(g, hdr->ct);
          if (hdr->ct == UO_CT_HLG) l = hlg_ootf_approx(l);
          /* globalTonemap: jpegr.cpp:1951-1977 */
          float c[3] = {l.r, l.g, l.b};
          if (is_norm) { c[0] *= headroom; c[1] *= headroom; c[2] *= headroom; }
          float mx = c[0];
          if (c[1] > mx) mx = c[1];
          if (c[2] > mx) mx = c[2];
          float ms = 1.0f + mx / (headroom * headroom); /* ReinhardMap: jpegr.cpp:1945-1949 */
          ms /= 1.0f + mx;
          ms = ms * mx;
          color_t o;
          o.r = c[0] > 0.0f ? c[0] * ms / mx : 0.0f;
          o.g = c[1] > 0.0f ? c[1] * ms / mx : 0.0f;
          o.b = c[2] > 0.0f ? c[2] * ms / mx : 0.0f;
          o = gamut_conv(o, gm);
          o.r = clamp01(o.r); o.g = clamp01(o.g); o.b = clamp01(o.b);
          color_t og = {srgb_oetf(o.r), srgb_oetf(o.g), srgb_oetf(o.b)};
          if (sdr_rgb) {
            uint32_t r0 = put8(og.r), g0 = put8(og.g), b0 = put8(og.b);
            ((uint32_t*)sdr->planes[0])[(x + j) + (y + i) * (size_t)sdr->stride[0]] =
                r0 | (g0 << 8) | (b0 << 16) | (255u << 24);
          } else {
            color_t yuv = rgb_to_yuv(og, &p3);
            yuv.g += 0.5f; yuv.b += 0.5f; /* r += 0.0f is the identity except for -0.0f, harmless */
            yuv.r += 0.0f;
            if (sdr->fmt != UO_FMT_YUV420) {
              yp[(x + j) + (y + i) * (size_t)sdr->stride[0]] = put8(yuv.r);
              up[(x + j) + (y + i) * (size_t)sdr->stride[1]] = put8(yuv.g);
              vp[(x + j) + (y + i) * (size_t)sdr->stride[2]] = put8(yuv.b);
            } else {
              yp[(y + i) * (size_t)sdr->stride[0] + x + j] = scale_to_8bit(yuv.r);
              su += yuv.g; sv += yuv.b;
            }
          }
        }
      }
      if (sdr->fmt == UO_FMT_YUV420) {
        su /= (float)(f * f); sv /= (float)(f * f);
        up[x / f + (y / f) * (size_t)sdr->stride[1]] = scale_to_8bit(su);
        vp[x / f + (y / f) * (size_t)sdr->stride[2]] = scale_to_8bit(sv);
      }
    }
  }
  return UO_OK;
}

/* ---------------------------------------------------------------------------------------------
 * UltraHdr::convertYuv -> transformYuv420/444: jpegr.cpp:436-518, gainmapmath.cpp:676-748
 * ------------------------------------------------------------------------------------------- */
static color_t yuv_mat(color_t e, const float* c) {
  color_t o = {e.r * c[0] + e.g * c[1] + e.b * c[2], e.r * c[3] + e.g * c[4] + e.b * c[5],
               e.r * c[6] + e.g * c[7] + e.b * c[8]};
  return o;
}
static uint8_t st8(float v) { /* static_cast<uint8_t>(CLIP3(v, 0, 255)) with v float */
  v = (v < 0) ? 0 : ((v > 255) ? 255 : v);
  return (uint8_t)v;
}
int uo_convert_yuv(uo_image_t* im, int src, int dst) {
  const float* c = NULL;
  if (src < 0 || src > 2 || dst < 0 || dst > 2) return UO_INVALID_PARAM;
  if (src == dst) return UO_OK;
  if (src == UO_CG_709) c = dst == UO_CG_P3 ? kYuv709To601 : kYuv709To2100;
  else if (src == UO_CG_P3) c = dst == UO_CG_709 ? kYuv601To709 : kYuv601To2100;
  else c = dst == UO_CG_709 ? kYuv2100To709 : kYuv2100To601;
  uint8_t *yp = (uint8_t*)im->planes[0], *up = (uint8_t*)im->planes[1], *vp = (uint8_t*)im->planes[2];
  if (im->fmt == UO_FMT_YUV420) {
    for (size_t y = 0; y < im->h / 2; ++y)
      for (size_t x = 0; x < im->w / 2; ++x) {
        color_t p1 = yuv_mat(get_pixel(im, x * 2, y * 2), c);
        color_t p2 = yuv_mat(get_pixel(im, x * 2 + 1, y * 2), c);
        color_t p3 = yuv_mat(get_pixel(im, x * 2, y * 2 + 1), c);
        color_t p4 = yuv_mat(get_pixel(im, x * 2 + 1, y * 2 + 1), c);
        float nu = (((p1.g + p2.g) + p3.g) + p4.g) / 4.0f;
        float nv = (((p1.b + p2.b) + p3.b) + p4.b) / 4.0f;
        yp[x * 2 + y * 2 * im->stride[0]] = st8(p1.r * 255.0f + 0.5f);
        yp[x * 2 + 1 + y * 2 * im->stride[0]] = st8(p2.r * 255.0f + 0.5f);
        yp[x * 2 + (y * 2 + 1) * im->stride[0]] = st8(p3.r * 255.0f + 0.5f);
        yp[x * 2 + 1 + (y * 2 + 1) * im->stride[0]] = st8(p4.r * 255.0f + 0.5f);
        up[x + y * im->stride[1]] = st8(nu * 255.0f + 128.0f + 0.5f);
        vp[x + y * im->stride[2]] = st8(nv * 255.0f + 128.0f + 0.5f);
      }
  } else if (im->fmt == UO_FMT_YUV444) {
    for (size_t y = 0; y < im->h; ++y)
      for (size_t x = 0; x < im->w; ++x) {
        color_t p = yuv_mat(get_pixel(im, x, y), c);
        yp[x + y * im->stride[0]] = st8(p.r * 255.0f + 0.5f);
        up[x + y * im->stride[1]] = st8(p.g * 255.0f + 128.0f + 0.5f);
        vp[x + y * im->stride[2]] = st8(p.b * 255.0f + 128.0f + 0.5f);
      }
  } else {
    return UO_UNSUPPORTED;
  }
  return UO_OK;
}

/* ---------------------------------------------------------------------------------------------
 * convert_raw_input_to_ycbcr: gainmapmath.cpp:1291-1482 (RGB variants; full range only)
 * ------------------------------------------------------------------------------------------- */
static float clipf(float v, float hi) { return (v < 0.0f) ? 0.0f : (v > hi) ? hi : v; }
int uo_convert_raw_input_to_ycbcr(const uo_image_t* src, int chroma, uo_image_t* dst) {
  if (src->fmt != UO_FMT_RGBA1010102 && src->fmt != UO_FMT_RGBA8888 && src->fmt != UO_FMT_RGB888)
    return UO_UNSUPPORTED;
  if (src->cg < 0 || src->cg > 2) return UO_UNSUPPORTED;
  const rgb2yuv_t k = rgb2yuv_coeffs(src->cg);
  dst->cg = src->cg; dst->ct = src->ct; dst->range = UO_CR_FULL; dst->w = src->w; dst->h = src->h;
  if (src->fmt == UO_FMT_RGBA1010102) {
    const uint32_t* rgb = (const uint32_t*)src->planes[0];
    size_t ss = src->stride[0];
    if (chroma) {
      dst->fmt = UO_FMT_P010;
      uint16_t* yd = (uint16_t*)dst->planes[0];
      uint16_t* ud = (uint16_t*)dst->planes[1];
      for (size_t i = 0; i < dst->h; i += 2)
        for (size_t j = 0; j < dst->w; j += 2) {
          color_t p[4];
          const size_t off[4] = {ss * i + j, ss * i + j + 1, ss * (i + 1) + j, ss * (i + 1) + j + 1};
          for (int q = 0; q < 4; q++) {
            uint32_t v = rgb[off[q]];
            color_t c = {(float)(v & 0x3ff), (float)((v >> 10) & 0x3ff), (float)((v >> 20) & 0x3ff)};
            c.r /= 1023.0f; c.g /= 1023.0f; c.b /= 1023.0f;
            p[q] = rgb_to_yuv(c, &k);
            p[q].r = clipf((p[q].r * 1023.0f) + 0.5f, 1023.0f);
          }
          size_t ds = dst->stride[0];
          yd[ds * i + j] = (uint16_t)((uint16_t)p[0].r << 6);
          yd[ds * i + j + 1] = (uint16_t)((uint16_t)p[1].r << 6);
          yd[ds * (i + 1) + j] = (uint16_t)((uint16_t)p[2].r << 6);
          yd[ds * (i + 1) + j + 1] = (uint16_t)((uint16_t)p[3].r << 6);
          float u = (p[0].g + p[1].g + p[2].g + p[3].g) / 4;
          float v = (p[0].b + p[1].b + p[2].b + p[3].b) / 4;
          u = clipf((u * 1023.0f) + 512.0f + 0.5f, 1023.0f);
          v = clipf((v * 1023.0f) + 512.0f + 0.5f, 1023.0f);
          ud[dst->stride[1] * (i / 2) + j] = (uint16_t)((uint16_t)u << 6);
          ud[dst->stride[1] * (i / 2) + j + 1] = (uint16_t)((uint16_t)v << 6);
        }
    } else {
      dst->fmt = UO_FMT_YUV444_10;
      uint16_t *yd = (uint16_t*)dst->planes[0], *ud = (uint16_t*)dst->planes[1], *vd = (uint16_t*)dst->planes[2];
      for (size_t i = 0; i < dst->h; i++)
        for (size_t j = 0; j < dst->w; j++) {
          uint32_t v = rgb[ss * i + j];
          color_t c = {(float)(v & 0x3ff), (float)((v >> 10) & 0x3ff), (float)((v >> 20) & 0x3ff)};
          c.r /= 1023.0f; c.g /= 1023.0f; c.b /= 1023.0f;
          color_t p = rgb_to_yuv(c, &k);
          yd[dst->stride[0] * i + j] = (uint16_t)clipf((p.r * 1023.0f) + 0.5f, 1023.0f);
          ud[dst->stride[1] * i + j] = (uint16_t)clipf((p.g * 1023.0f) + 512.0f + 0.5f, 1023.0f);
          vd[dst->stride[2] * i + j] = (uint16_t)clipf((p.b * 1023.0f) + 512.0f + 0.5f, 1023.0f);
        }
    }
    return UO_OK;
  }
  uint8_t *yd = (uint8_t*)dst->planes[0], *ud = (uint8_t*)dst->planes[1], *vd = (uint8_t*)dst->planes[2];
  if (chroma) {
    dst->fmt = UO_FMT_YUV420;
    for (size_t i = 0; i < dst->h; i += 2)
      for (size_t j = 0; j < dst->w; j += 2) {
        color_t p[4] = {get_pixel(src, j, i), get_pixel(src, j + 1, i), get_pixel(src, j, i + 1),
                        get_pixel(src, j + 1, i + 1)};
        for (int q = 0; q < 4; q++) {
          p[q] = rgb_to_yuv(p[q], &k);
          p[q].r = clipf(p[q].r * 255.0f + 0.5f, 255.0f);
        }
        yd[dst->stride[0] * i + j] = (uint8_t)p[0].r;
        yd[dst->stride[0] * i + j + 1] = (uint8_t)p[1].r;
        yd[dst->stride[0] * (i + 1) + j] = (uint8_t)p[2].r;
        yd[dst->stride[0] * (i + 1) + j + 1] = (uint8_t)p[3].r;
        float u = (p[0].g + p[1].g + p[2].g + p[3].g) / 4;
        float v = (p[0].b + p[1].b + p[2].b + p[3].b) / 4;
        ud[dst->stride[1] * (i / 2) + (j / 2)] = (uint8_t)clipf(u * 255.0f + 0.5f + 128.0f, 255.0f);
        vd[dst->stride[2] * (i / 2) + (j / 2)] = (uint8_t)clipf(v * 255.0f + 0.5f + 128.0f, 255.0f);
      }
  } else {
    dst->fmt = UO_FMT_YUV444;
    for (size_t i = 0; i < dst->h; i++)
      for (size_t j = 0; j < dst->w; j++) {
        color_t p = rgb_to_yuv(get_pixel(src, j, i), &k);
        yd[dst->stride[0] * i + j] = (uint8_t)clipf(p.r * 255.0f + 0.5f, 255.0f);
        ud[dst->stride[1] * i + j] = (uint8_t)clipf(p.g * 255.0f + 0.5f + 128.0f, 255.0f);
        vd[dst->stride[2] * i + j] = (uint8_t)clipf(p.b * 255.0f + 0.5f + 128.0f, 255.0f);
      }
  }
  return UO_OK;
}

/* ---------------------------------------------------------------------------------------------
 * JPEG DCT/quantize stage.  The arithmetic lives in libjpeg (libjpeg-turbo 3.1.0 pinned by the
 * reference, CMakeLists.txt:519-521; NOT under /root/reference) and is restated here from the
 * published algorithm: Loeffler-Ligtenberg-Moschytz "islow" integer FDCT (jfdctint.c:
 * CONST_BITS 13, PASS1_BITS 2, row pass then column pass, round-to-nearest descale), Annex-K
 * base tables scaled by jpeg_quality_scaling() with force_baseline, and jcdctmgr.c's
 * round-half-away integer division by (quant << 3).  Reference call sites:
 * lib/src/jpegencoderhelper.cpp:187-198 (jpeg_set_quality(q, TRUE), JDCT_ISLOW).
 * Pinned against libjpeg itself via jpeg_read_coefficients (tests/test_oracle_vs_ref.py).
 * ------------------------------------------------------------------------------------------- */
static const uint8_t kStdLuma[64] = {
    16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
    14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
    49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t kStdChroma[64] = {
    17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
    47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
    99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
void uo_jpeg_quant_table(int quality, int is_chroma, uint16_t qt[64]) {
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  const uint8_t* base = is_chroma ? kStdChroma : kStdLuma;
  for (int i = 0; i < 64; i++) {
    long t = ((long)base[i] * scale + 50L) / 100L;
    if (t <= 0L) t = 1L;
    if (t > 255L) t = 255L; /* force_baseline */
    qt[i] = (uint16_t)t;
  }
}
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n)-1))) >> (n))
static void fdct_islow_1d(const int32_t in[8], int32_t out[8], int pass) {
  const int sh = pass == 0 ? 13 - 2 : 13 + 2;
  int32_t t0 = in[0] + in[7], t7 = in[0] - in[7], t1 = in[1] + in[6], t6 = in[1] - in[6];
  int32_t t2 = in[2] + in[5], t5 = in[2] - in[5], t3 = in[3] + in[4], t4 = in[3] - in[4];
  int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  if (pass == 0) {
    out[0] = (t10 + t11) * 4;
    out[4] = (t10 - t11) * 4;
  } else {
    out[0] = DESCALE(t10 + t11, 2);
    out[4] = DESCALE(t10 - t11, 2);
  }
  int32_t z1 = (t12 + t13) * FIX_0_541196100;
  out[2] = DESCALE(z1 + t13 * FIX_0_765366865, sh);
  out[6] = DESCALE(z1 + t12 * (-FIX_1_847759065), sh);
  z1 = t4 + t7;
  int32_t z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  int32_t z5 = (z3 + z4) * FIX_1_175875602;
  t4 *= FIX_0_298631336; t5 *= FIX_2_053119869; t6 *= FIX_3_072711026; t7 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
  z3 += z5; z4 += z5;
  out[7] = DESCALE(t4 + z1 + z3, sh);
  out[5] = DESCALE(t5 + z2 + z4, sh);
  out[3] = DESCALE(t6 + z2 + z3, sh);
  out[1] = DESCALE(t7 + z1 + z4, sh);
}
void uo_fdct_quant_plane(const uint8_t* plane, size_t stride, int bw, int bh, const uint16_t qt[64],
                         int16_t* coef) {
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      int32_t ws[64], in[8], o[8];
      for (int r = 0; r < 8; r++) {
        const uint8_t* p = plane + (size_t)(by * 8 + r) * stride + (size_t)bx * 8;
        for (int c = 0; c < 8; c++) in[c] = (int32_t)p[c] - 128;
        fdct_islow_1d(in, o, 0);
        for (int c = 0; c < 8; c++) ws[r * 8 + c] = o[c];
      }
      for (int c = 0; c < 8; c++) {
        for (int r = 0; r < 8; r++) in[r] = ws[r * 8 + c];
        fdct_islow_1d(in, o, 1);
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = o[r];
      }
      int16_t* out = coef + ((size_t)by * bw + bx) * 64;
      for (int i = 0; i < 64; i++) {
        int32_t q = (int32_t)qt[i] << 3, t = ws[i];
        if (t < 0) {
          t = -t; t += q >> 1;
          t = t >= q ? t / q : 0;
          t = -t;
        } else {
          t += q >> 1;
          t = t >= q ? t / q : 0;
        }
        out[i] = (int16_t)t;
      }
    }
}
/* jccolor.c rgb_ycc_convert (16-bit fixed point).  Constants: the 6b / libjpeg-turbo set
 * (0.29900 0.58700 0.11400 / 0.16874 0.33126 0.50000 / 0.41869 0.08131). */
void uo_jpeg_rgb_to_ycc(const uint8_t* rgb, size_t stride_px, int w, int h, uint8_t* y, uint8_t* cb,
                        uint8_t* cr, size_t os) {
#define FIX16(x) ((int32_t)((x) * 65536.0 + 0.5))
  const int32_t half = 1 << 15, off = 128 << 16;
  for (int j = 0; j < h; j++)
    for (int i = 0; i < w; i++) {
      const uint8_t* p = rgb + ((size_t)j * stride_px + i) * 3;
      int32_t r = p[0], g = p[1], b = p[2];
      y[j * os + i] = (uint8_t)((FIX16(0.29900) * r + FIX16(0.58700) * g + FIX16(0.11400) * b + half) >> 16);
      cb[j * os + i] = (uint8_t)(((-FIX16(0.16874)) * r + (-FIX16(0.33126)) * g + FIX16(0.50000) * b + off + half - 1) >> 16);
      cr[j * os + i] = (uint8_t)((FIX16(0.50000) * r + (-FIX16(0.41869)) * g + (-FIX16(0.08131)) * b + off + half - 1) >> 16);
    }
#undef FIX16
}

/* ---------------------------------------------------------------------------------------------
 * copy_raw_image(src, dst) (gainmapmath.cpp:1492-1613): strided plane copies for equal formats,
 * RGB888 -> RGBA8888 (alpha 0xff) and RGBA8888 -> Y400 (takes the R byte).  Colour aspects are
 * copied before the format check (:1505-1507); P010 / 4:2:0 copy h/2 chroma rows of w (resp. w/2)
 * samples (:1519-1523, 1540-1547), i.e. the last chroma row / column of an odd image is not copied.
 * ------------------------------------------------------------------------------------------- */
int uo_copy_raw_image(const uo_image_t* src, uo_image_t* dst) {
  if (dst->w != src->w || dst->h != src->h) return UO_MEM_ERROR;
  dst->cg = src->cg; dst->ct = src->ct; dst->range = src->range;
  const size_t w = src->w, h = src->h;
  if (dst->fmt == src->fmt) {
    if (src->fmt == UO_FMT_P010) {
      for (size_t i = 0; i < h; i++)
        memcpy((uint8_t*)dst->planes[0] + i * dst->stride[0] * 2, (const uint8_t*)src->planes[0] + i * src->stride[0] * 2, w * 2);
      for (size_t i = 0; i < h / 2; i++)
        memcpy((uint8_t*)dst->planes[1] + i * dst->stride[1] * 2, (const uint8_t*)src->planes[1] + i * src->stride[1] * 2, w * 2);
      return UO_OK;
    }
    if (src->fmt == UO_FMT_YUV420) {
      for (size_t i = 0; i < h; i++)
        memcpy((uint8_t*)dst->planes[0] + i * dst->stride[0], (const uint8_t*)src->planes[0] + i * src->stride[0], w);
      for (size_t i = 0; i < h / 2; i++) {
        memcpy((uint8_t*)dst->planes[1] + i * dst->stride[1], (const uint8_t*)src->planes[1] + i * src->stride[1], w / 2);
        memcpy((uint8_t*)dst->planes[2] + i * dst->stride[2], (const uint8_t*)src->planes[2] + i * src->stride[2], w / 2);
      }
      return UO_OK;
    }
    size_t bpp = 0;
    if (src->fmt == UO_FMT_Y400) bpp = 1;
    else if (src->fmt == UO_FMT_RGBA1010102 || src->fmt == UO_FMT_RGBA8888) bpp = 4;
    else if (src->fmt == UO_FMT_RGBAF16) bpp = 8;
    else if (src->fmt == UO_FMT_RGB888) bpp = 3;
    if (bpp) {
      for (size_t i = 0; i < h; i++)
        memcpy((uint8_t*)dst->planes[0] + i * dst->stride[0] * bpp, (const uint8_t*)src->planes[0] + i * src->stride[0] * bpp, w * bpp);
      return UO_OK;
    }
  } else if (src->fmt == UO_FMT_RGB888 && dst->fmt == UO_FMT_RGBA8888) {
    for (size_t i = 0; i < h; i++) {
      const uint8_t* s = (const uint8_t*)src->planes[0] + i * src->stride[0] * 3;
      uint32_t* d = (uint32_t*)dst->planes[0] + i * dst->stride[0];
      for (size_t j = 0; j < w; j++) d[j] = s[3 * j] | ((uint32_t)s[3 * j + 1] << 8) | ((uint32_t)s[3 * j + 2] << 16) | (0xffu << 24);
    }
    return UO_OK;
  } else if (src->fmt == UO_FMT_RGBA8888 && dst->fmt == UO_FMT_Y400) {
    for (size_t i = 0; i < h; i++) {
      const uint8_t* s = (const uint8_t*)src->planes[0] + i * src->stride[0] * 4;
      uint8_t* d = (uint8_t*)dst->planes[0] + i * dst->stride[0];
      for (size_t j = 0; j < w; j++) d[j] = s[4 * j];
    }
    return UO_OK;
  }
  return UO_UNSUPPORTED;
}

/* ---------------------------------------------------------------------------------------------
 * JPEG decode stage: dequantize + islow inverse DCT + range limit, and the
 * YCbCr -> RGB conversion libjpeg applies to a 3-channel gain map.  Like the forward DCT this
 * arithmetic lives in libjpeg, not in the reference tree: JpegDecoderHelper
 * (/root/reference/lib/src/jpegdecoderhelper.cpp:169-535) sets dct_method = JDCT_ISLOW and reads
 * raw data (base image, Y400 map) or scanlines (RGB map).  Restated from the published
 * Loeffler-Ligtenberg-Moschytz routine of libjpeg's jidctint.c (CONST_BITS 13, PASS1_BITS 2:
 * column pass from the dequantized coefficients into a workspace scaled by 4, then a row pass,
 * final descale by 2^18, +128, range limit) and jdcolor.c's table-driven ycc_rgb_convert.
 * ------------------------------------------------------------------------------------------- */
static void idct_islow_1d(const int32_t in[8], int32_t out[8], int pass) {
  /* pass 0: descale by CONST_BITS - PASS1_BITS; pass 1: by CONST_BITS + PASS1_BITS + 3 */
  const int sh = pass == 0 ? 13 - 2 : 13 + 2 + 3;
  int32_t z2 = in[2], z3 = in[6];
  int32_t z1 = (z2 + z3) * FIX_0_541196100;
  int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
  int32_t tmp3 = z1 + z2 * FIX_0_765366865;
  z2 = in[0]; z3 = in[4];
  int32_t tmp0 = (int32_t)((uint32_t)(z2 + z3) << 13);
  int32_t tmp1 = (int32_t)((uint32_t)(z2 - z3) << 13);
  const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
  tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
  z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
  int32_t z4 = tmp1 + tmp3;
  const int32_t z5 = (z3 + z4) * FIX_1_175875602;
  tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
  z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
  z3 += z5; z4 += z5;
  tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
  out[0] = DESCALE(tmp10 + tmp3, sh); out[7] = DESCALE(tmp10 - tmp3, sh);
  out[1] = DESCALE(tmp11 + tmp2, sh); out[6] = DESCALE(tmp11 - tmp2, sh);
  out[2] = DESCALE(tmp12 + tmp1, sh); out[5] = DESCALE(tmp12 - tmp1, sh);
  out[3] = DESCALE(tmp13 + tmp0, sh); out[4] = DESCALE(tmp13 - tmp0, sh);
}
/* libjpeg's range_limit table indexed with (x & RANGE_MASK), RANGE_MASK = 1023, centred on 128:
 * v = (x + 128) mod 1024 -> v for v <= 255, 255 up to 639, 0 above (the wrap only matters for
 * corrupt streams). */
static uint8_t idct_range_limit(int32_t x) {
  const uint32_t v = (uint32_t)(x + 128) & 1023u;
  return (uint8_t)(v <= 255 ? v : (v < 640 ? 255 : 0));
}
void uo_idct_dequant_plane(const int16_t* coef, int bw, int bh, const uint16_t qt[64], uint8_t* plane, size_t stride) {
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      const int16_t* in = coef + ((size_t)by * bw + bx) * 64;
      int32_t ws[64];
      for (int c = 0; c < 8; c++) { /* pass 1: columns */
        int32_t col[8], o[8];
        for (int r = 0; r < 8; r++) col[r] = (int32_t)in[r * 8 + c] * (int32_t)qt[r * 8 + c];
        idct_islow_1d(col, o, 0);
        for (int r = 0; r < 8; r++) ws[r * 8 + c] = o[r];
      }
      for (int r = 0; r < 8; r++) { /* pass 2: rows */
        int32_t o[8];
        idct_islow_1d(ws + r * 8, o, 1);
        uint8_t* dst = plane + (size_t)(by * 8 + r) * stride + (size_t)bx * 8;
        for (int c = 0; c < 8; c++) dst[c] = idct_range_limit(o[c]);
      }
    }
}
/* jdcolor.c ycc_rgb_convert.  variant 0: libjpeg 6b / libjpeg-turbo constants (1.40200 1.77200
 * 0.71414 0.34414) -- the reference pins libjpeg-turbo 3.1.0; variant 1: IJG 9 constants
 * (1.402 1.772 0.714136286 0.344136286) -- the library in this image, used to pin the restatement.
 * The two differ in the green term for 59 of the 65536 (Cb, Cr) pairs.  out_bpp 3 (RGB888) or 4
 * (RGBA8888, alpha 255). */
void uo_jpeg_ycc_to_rgb(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, size_t in_stride, int w, int h,
                        uint8_t* rgb, size_t out_stride_px, int out_bpp, int variant) {
#define FIX16(x) ((int32_t)((x) * 65536.0 + 0.5))
  const int32_t k_cr_r = FIX16(1.40200), k_cb_b = FIX16(1.77200);
  const int32_t k_cr_g = variant ? FIX16(0.714136286) : FIX16(0.71414);
  const int32_t k_cb_g = variant ? FIX16(0.344136286) : FIX16(0.34414);
  const int32_t half = 1 << 15;
  for (int j = 0; j < h; j++)
    for (int i = 0; i < w; i++) {
      const int32_t yy = y[j * in_stride + i], u = (int32_t)cb[j * in_stride + i] - 128, v = (int32_t)cr[j * in_stride + i] - 128;
      int32_t r = yy + ((k_cr_r * v + half) >> 16);
      int32_t g = yy + (((-k_cb_g) * u + half + (-k_cr_g) * v) >> 16);
      int32_t b = yy + ((k_cb_b * u + half) >> 16);
      uint8_t* o = rgb + ((size_t)j * out_stride_px + i) * out_bpp;
      o[0] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
      o[1] = (uint8_t)(g < 0 ? 0 : (g > 255 ? 255 : g));
      o[2] = (uint8_t)(b < 0 ? 0 : (b > 255 ? 255 : b));
      if (out_bpp == 4) o[3] = 255;
    }
#undef FIX16
}

/* ---------------------------------------------------------------------------------------------
 * scalar access for KATs (function ids mirror oracle/ref_shim.cpp)
 * ------------------------------------------------------------------------------------------- */
int uo_eval(int fn, const float* in, float* out, size_t n) {
  init_luts();
  for (size_t i = 0; i < n; i++) {
    float x = in[i];
    switch (fn) {
      case 0: out[i] = srgb_inv_oetf(x); break;
      case 1: out[i] = g_lut_srgb[lut_index(x, N_SRGB)]; break;
      case 2: out[i] = srgb_oetf(x); break;
      case 3: out[i] = hlg_oetf(x); break;
      case 4: out[i] = g_lut_hlg[lut_index(x, N_OETF)]; break;
      case 5: out[i] = hlg_inv_oetf(x); break;
      case 6: out[i] = g_lut_hlg_inv[lut_index(x, N_INV)]; break;
      case 7: out[i] = pq_oetf(x); break;
      case 8: out[i] = g_lut_pq[lut_index(x, N_OETF)]; break;
      case 9: out[i] = pq_inv_oetf(x); break;
      case 10: out[i] = g_lut_pq_inv[lut_index(x, N_INV)]; break;
      case 11: out[i] = half_to_float((uint16_t)x); break;
      case 12: out[i] = powf(x, 1.2f); break;
      case 13: out[i] = powf(x, 1.0f / 1.2f); break;
      case 14: out[i] = (float)log2((double)x); break; /* encodeGain / computeGain's log2 (gainmapmath.cpp:767, 774) */
      default: return -1;
    }
  }
  return 0;
}
/* 10-bit output code of applyGainMap's HLG / PQ tail for an already clamped v (jpegr.cpp:1783-1786,
 * 1799-1801): [hlgInverseOotfApprox,] OETF LUT, colorToRgba1010102's per-channel quantisation. */
void uo_oetf_code(int ct, const float* in, uint32_t* out, size_t n) {
  init_luts();
  for (size_t i = 0; i < n; i++) {
    float x = in[i];
    if (ct == UO_CT_HLG) x = powf(x, 1.0f / 1.2f);
    float e = (ct == UO_CT_HLG ? g_lut_hlg : g_lut_pq)[lut_index(x, N_OETF)];
    float q = e * 1023 + 0.5f;
    out[i] = (uint32_t)(q < 0.0f ? 0.0f : (q > 1023.0f ? 1023.0f : q));
  }
}
void uo_float_to_half(const float* in, uint16_t* out, size_t n) {
  for (size_t i = 0; i < n; i++) out[i] = float_to_half(in[i]);
}
uint32_t uo_color_to_rgba1010102(float r, float g, float b) { color_t c = {r, g, b}; return to_1010102(c); }
uint64_t uo_color_to_rgbaf16(float r, float g, float b) { color_t c = {r, g, b}; return to_f16(c); }
float uo_compute_gain(float sdr, float hdr) { return compute_gain(sdr, hdr); }
uint8_t uo_affine_map_gain(float g, float mn, float mx, float gamma) { return affine_map_gain(g, mn, mx, gamma); }
uint8_t uo_encode_gain(float y_sdr, float y_hdr, float min_boost, float max_boost, float gamma) {
  /* encodeGain 4-arg form: gainmapmath.cpp:753-756 -- log2() double narrowed to the float params */
  return encode_gain(y_sdr, y_hdr, min_boost, max_boost, gamma, (float)log2((double)min_boost),
                     (float)log2((double)max_boost));
}
void uo_apply_gain(const float e[3], float gain, const uo_metadata_t* md, float weight, int use_lut,
                   float out[3]) {
  float f;
  if (use_lut) {
    gain_lut_t* l = (gain_lut_t*)malloc(sizeof *l);
    gain_lut_init(l, md, weight);
    f = gain_factor(l, gain, 0);
    free(l);
  } else {
    f = apply_gain_exact_factor(gain, md, 0, weight);
  }
  for (int c = 0; c < 3; c++) out[c] = ((e[c] + md->offset_sdr[0]) * f) - md->offset_hdr[0];
}
void uo_idw_weights(int scale, int which, float* out) {
  fill_idw(out, scale, which == 0 || which == 2, which == 0 || which == 1);
}
void uo_color_fn(int fn, const float in[3], float out[3]) {
  color_t c = {in[0], in[1], in[2]}, r = c;
  int ok;
  if (fn >= 0 && fn <= 2) { yuv2rgb_t k = yuv2rgb_coeffs(fn); r = yuv_to_rgb(c, &k); }
  else if (fn >= 3 && fn <= 5) { rgb2yuv_t k = rgb2yuv_coeffs(fn - 3); r = rgb_to_yuv(c, &k); }
  else if (fn == 6) r = gamut_conv(c, gamut_matrix(UO_CG_P3, UO_CG_709, &ok));
  else if (fn == 7) r = gamut_conv(c, gamut_matrix(UO_CG_2100, UO_CG_709, &ok));
  else if (fn == 8) r = gamut_conv(c, gamut_matrix(UO_CG_709, UO_CG_P3, &ok));
  else if (fn == 9) r = gamut_conv(c, gamut_matrix(UO_CG_2100, UO_CG_P3, &ok));
  else if (fn == 10) r = gamut_conv(c, gamut_matrix(UO_CG_709, UO_CG_2100, &ok));
  else if (fn == 11) r = gamut_conv(c, gamut_matrix(UO_CG_P3, UO_CG_2100, &ok));
  else if (fn == 12) { r.r = luminance(c, 0); r.g = luminance(c, 1); r.b = luminance(c, 2); }
  out[0] = r.r; out[1] = r.g; out[2] = r.b;
}
void uo_lut(int which, float* out) {
  init_luts();
  switch (which) {
    case 0: memcpy(out, g_lut_srgb, sizeof g_lut_srgb); break;
    case 1: memcpy(out, g_lut_hlg_inv, sizeof g_lut_hlg_inv); break;
    case 2: memcpy(out, g_lut_pq_inv, sizeof g_lut_pq_inv); break;
    case 3: memcpy(out, g_lut_hlg, sizeof g_lut_hlg); break;
    case 4: memcpy(out, g_lut_pq, sizeof g_lut_pq); break;
  }
}

/* =================================================================================================
 * Baseline Huffman entropy coding of quantized coefficient blocks (the step after
 * fdct_quant).  In the reference this is libjpeg behind JpegEncoderHelper::compressImage
 * (/root/reference/lib/src/jpegencoderhelper.cpp:131-244: jpeg_set_defaults -> the Annex K tables,
 * optimize_coding off, no restart markers).  Restated from the public algorithm: ITU-T T.81 Annex C
 * (code generation), F.1.2 (DC differences, run/size AC symbols, ZRL, EOB), F.1.2.3 (byte stuffing),
 * E.1.4 / B.2.4.4 (restart intervals), as implemented by jchuff.c encode_mcu_huff / encode_one_block /
 * emit_restart and jctrans.c compress_output (dummy blocks at the right and bottom edges: AC zero, DC
 * equal to the previous block of the MCU).  Header writing follows jcmarker.c's marker order.
 * ================================================================================================= */
static const uint8_t kZigzagToNatural[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
/* T.81 Annex K.3 typical tables (jpeg_set_defaults installs exactly these) */
static const uint8_t kBitsDcLuma[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t kBitsDcChroma[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t kValDc[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t kBitsAcLuma[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t kValAcLuma[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81,
    0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18,
    0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
    0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75,
    0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99,
    0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5,
    0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t kBitsAcChroma[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t kValAcChroma[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08,
    0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25,
    0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47,
    0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74,
    0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97,
    0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4,
    0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

void uo_std_huff_table(int is_ac, int is_chroma, uint8_t bits[17], uint8_t vals[256], int* nvals) {
  const uint8_t* b = is_ac ? (is_chroma ? kBitsAcChroma : kBitsAcLuma) : (is_chroma ? kBitsDcChroma : kBitsDcLuma);
  const uint8_t* v = is_ac ? (is_chroma ? kValAcChroma : kValAcLuma) : kValDc;
  int n = 0;
  for (int i = 0; i <= 16; i++) {
    bits[i] = b[i];
    if (i) n += b[i];
  }
  memset(vals, 0, 256);
  memcpy(vals, v, (size_t)n);
  *nvals = n;
}

/* T.81 Annex C: code sizes and codes from BITS / HUFFVAL (jpeg_make_c_derived_tbl) */
typedef struct {
  uint16_t code[256];
  uint8_t len[256];
} uo_huff_enc_t;
static void make_derived(int is_ac, int is_chroma, uo_huff_enc_t* t) {
  uint8_t bits[17], vals[256];
  int n;
  uo_std_huff_table(is_ac, is_chroma, bits, vals, &n);
  memset(t, 0, sizeof *t);
  unsigned code = 0;
  int k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < bits[l]; i++, k++) {
      t->code[vals[k]] = (uint16_t)code;
      t->len[vals[k]] = (uint8_t)l;
      code++;
    }
    code <<= 1;
  }
}

typedef struct {
  uint8_t* out;
  size_t cap, n;
  uint64_t acc; /* bits not yet written, right aligned */
  int nacc;
  int overflow;
} uo_bitw_t;
static void bw_byte(uo_bitw_t* w, unsigned b) {
  if (w->n < w->cap) w->out[w->n] = (uint8_t)b; else w->overflow = 1;
  w->n++;
}
static void bw_put(uo_bitw_t* w, unsigned code, int len) { /* emit_bits: MSB first, 0xFF followed by a stuffed 0x00 */
  w->acc = (w->acc << len) | (code & ((1u << len) - 1u));
  w->nacc += len;
  while (w->nacc >= 8) {
    const unsigned b = (unsigned)(w->acc >> (w->nacc - 8)) & 0xffu;
    bw_byte(w, b);
    if (b == 0xff) bw_byte(w, 0);
    w->nacc -= 8;
  }
}
static void bw_flush(uo_bitw_t* w) { /* flush_bits: fill the partial byte with ones */
  if (w->nacc > 0) bw_put(w, 0x7f, 8 - w->nacc);
  w->acc = 0;
  w->nacc = 0;
}
static int bit_length(unsigned v) {
  int n = 0;
  while (v) { n++; v >>= 1; }
  return n;
}
static void encode_block(uo_bitw_t* w, const int16_t blk[64], int dc_only_value, int is_dummy, int* last_dc,
                         const uo_huff_enc_t* dct, const uo_huff_enc_t* act) {
  /* jchuff.c encode_one_block */
  const int dc = is_dummy ? dc_only_value : blk[0];
  int temp = dc - *last_dc, temp2 = temp;
  *last_dc = dc;
  if (temp < 0) { temp = -temp; temp2--; }
  int nbits = bit_length((unsigned)temp);
  bw_put(w, dct->code[nbits], dct->len[nbits]);
  if (nbits) bw_put(w, (unsigned)temp2, nbits);
  int r = 0;
  if (!is_dummy) {
    for (int k = 1; k < 64; k++) {
      temp = blk[kZigzagToNatural[k]];
      if (temp == 0) { r++; continue; }
      while (r > 15) { bw_put(w, act->code[0xf0], act->len[0xf0]); r -= 16; }
      temp2 = temp;
      if (temp < 0) { temp = -temp; temp2--; }
      nbits = bit_length((unsigned)temp);
      const int sym = (r << 4) + nbits;
      bw_put(w, act->code[sym], act->len[sym]);
      bw_put(w, (unsigned)temp2, nbits);
      r = 0;
    }
  } else {
    r = 63;
  }
  if (r > 0) bw_put(w, act->code[0], act->len[0]);
}

static void scan_geometry(const uo_scan_t* sc, int* mcus_per_row, int* mcu_rows) {
  if (sc->ncomp == 1) { /* non-interleaved: an MCU is one block, no dummy blocks (jcmaster.c per_scan_setup) */
    *mcus_per_row = sc->bw[0];
    *mcu_rows = sc->bh[0];
    return;
  }
  int hmax = 1, vmax = 1;
  for (int c = 0; c < sc->ncomp; c++) {
    if (sc->hs[c] > hmax) hmax = sc->hs[c];
    if (sc->vs[c] > vmax) vmax = sc->vs[c];
  }
  *mcus_per_row = (int)((sc->w + 8u * hmax - 1) / (8u * hmax));
  *mcu_rows = (int)((sc->h + 8u * vmax - 1) / (8u * vmax));
}

/* entropy-coded data of the single scan: everything between the SOS header and EOI, RSTn markers included */
size_t uo_huffman_encode_scan(const uo_scan_t* sc, uint8_t* out, size_t cap) {
  uo_huff_enc_t dct[2], act[2];
  for (int t = 0; t < 2; t++) { make_derived(0, t, &dct[t]); make_derived(1, t, &act[t]); }
  int mpr, mrows;
  scan_geometry(sc, &mpr, &mrows);
  uo_bitw_t w = {out, cap, 0, 0, 0, 0};
  int last_dc[3] = {0, 0, 0};
  int to_go = sc->restart_interval, next_rst = 0;
  for (int my = 0; my < mrows; my++) {
    for (int mx = 0; mx < mpr; mx++) {
      if (sc->restart_interval) {
        if (to_go == 0) { /* emit_restart */
          bw_flush(&w);
          bw_byte(&w, 0xff);
          bw_byte(&w, 0xd0u + (unsigned)next_rst);
          last_dc[0] = last_dc[1] = last_dc[2] = 0;
          to_go = sc->restart_interval;
          next_rst = (next_rst + 1) & 7;
        }
        to_go--;
      }
      int prev_dc = 0; /* DC of the previous block in MCU order (jctrans.c dummy rule) */
      for (int c = 0; c < sc->ncomp; c++) {
        const int hs = sc->ncomp == 1 ? 1 : sc->hs[c], vs = sc->ncomp == 1 ? 1 : sc->vs[c], tbl = c ? 1 : 0;
        for (int yi = 0; yi < vs; yi++) {
          for (int xi = 0; xi < hs; xi++) {
            const int by = my * vs + yi, bx = mx * hs + xi;
            const int real = by < sc->bh[c] && bx < sc->bw[c];
            const int16_t* blk = real ? sc->coef[c] + ((size_t)by * sc->bw[c] + bx) * 64 : NULL;
            encode_block(&w, blk, prev_dc, !real, &last_dc[c], &dct[tbl], &act[tbl]);
            prev_dc = real ? blk[0] : prev_dc;
          }
        }
      }
    }
  }
  bw_flush(&w);
  return w.overflow ? 0 : w.n;
}

/* A complete baseline JFIF file around the entropy-coded data (jcmarker.c: SOI, APP0, DQT, SOF0, DHT, DRI, SOS ... EOI).
 * qt: natural order, table 0 for component 0, table 1 for the others. */
size_t uo_jpeg_assemble(const uo_scan_t* sc, const uint16_t qt[2][64], const uint8_t* scan, size_t scan_len, uint8_t* out, size_t cap) {
  size_t n = 0;
#define PUT(b) do { if (n < cap) out[n] = (uint8_t)(b); n++; } while (0)
#define PUT16(v) do { PUT((v) >> 8); PUT((v) & 0xff); } while (0)
  PUT(0xff); PUT(0xd8);
  PUT(0xff); PUT(0xe0); PUT16(16); PUT('J'); PUT('F'); PUT('I'); PUT('F'); PUT(0); PUT(1); PUT(1); PUT(0); PUT16(1); PUT16(1); PUT(0); PUT(0);
  const int ntab = sc->ncomp > 1 ? 2 : 1;
  for (int t = 0; t < ntab; t++) {
    PUT(0xff); PUT(0xdb); PUT16(67); PUT(t);
    for (int i = 0; i < 64; i++) PUT(qt[t][kZigzagToNatural[i]]);
  }
  PUT(0xff); PUT(0xc0); PUT16(8 + 3 * sc->ncomp); PUT(8); PUT16(sc->h); PUT16(sc->w); PUT(sc->ncomp);
  for (int c = 0; c < sc->ncomp; c++) { PUT(c + 1); PUT(((sc->ncomp == 1 ? 1 : sc->hs[c]) << 4) | (sc->ncomp == 1 ? 1 : sc->vs[c])); PUT(c ? 1 : 0); }
  for (int t = 0; t < ntab; t++) {
    for (int ac = 0; ac < 2; ac++) {
      uint8_t bits[17], vals[256];
      int nv;
      uo_std_huff_table(ac, t, bits, vals, &nv);
      PUT(0xff); PUT(0xc4); PUT16(2 + 1 + 16 + nv); PUT((ac << 4) | t);
      for (int i = 1; i <= 16; i++) PUT(bits[i]);
      for (int i = 0; i < nv; i++) PUT(vals[i]);
    }
  }
  if (sc->restart_interval) { PUT(0xff); PUT(0xdd); PUT16(4); PUT16(sc->restart_interval); }
  PUT(0xff); PUT(0xda); PUT16(6 + 2 * sc->ncomp); PUT(sc->ncomp);
  for (int c = 0; c < sc->ncomp; c++) { PUT(c + 1); PUT(c ? 0x11 : 0x00); }
  PUT(0); PUT(63); PUT(0);
  for (size_t i = 0; i < scan_len; i++) PUT(scan[i]);
  PUT(0xff); PUT(0xd9);
#undef PUT
#undef PUT16
  return n <= cap ? n : 0;
}

/* ---- Huffman decoding of a scan with restart intervals (the inverse of uo_huffman_encode_scan) -------------------
 * jdhuff.c decode_mcu_huff / process_restart restated per restart interval: the intervals are located by their RSTn
 * markers (inside entropy-coded data 0xFF is followed by 0x00 or by a marker), each is decoded on its own with the DC
 * predictors starting at zero.  Tables come as DHT content (BITS / HUFFVAL): [0] DC luma, [1] AC luma, [2] DC chroma,
 * [3] AC chroma.  coef_out[c]: bw[c] x bh[c] real blocks, natural order; dummy blocks are decoded and dropped.
 * Returns 0, or a negative number for a malformed stream. */
typedef struct {
  int maxcode[18]; /* largest code of length l, -1 if none; [17] sentinel */
  int valoff[17];  /* index into vals of the first symbol of length l minus its code */
  uint8_t vals[256];
} uo_huff_dec_t;
static void make_decoder(const uint8_t bits[17], const uint8_t* vals, uo_huff_dec_t* d) {
  int code = 0, k = 0;
  memset(d, 0, sizeof *d);
  for (int l = 1; l <= 16; l++) {
    if (bits[l]) {
      d->valoff[l] = k - code;
      k += bits[l];
      code += bits[l];
      d->maxcode[l] = code - 1;
    } else {
      d->maxcode[l] = -1;
    }
    code <<= 1;
  }
  d->maxcode[17] = 0x7fffffff;
  memcpy(d->vals, vals, (size_t)k);
}
typedef struct {
  const uint8_t* p;
  const uint8_t* end;
  uint32_t acc;
  int n;
} uo_bitr_t;
static int br_bit(uo_bitr_t* r) {
  if (r->n == 0) {
    unsigned b = 0; /* past the end of the interval: zeros (jdhuff.c fills with zeros after a marker) */
    if (r->p < r->end) {
      b = *r->p++;
      if (b == 0xff && r->p < r->end && *r->p == 0) r->p++; /* stuffed zero */
    }
    r->acc = b;
    r->n = 8;
  }
  r->n--;
  return (int)((r->acc >> r->n) & 1u);
}
static int br_symbol(uo_bitr_t* r, const uo_huff_dec_t* d) {
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | br_bit(r);
    if (d->maxcode[l] >= 0 && code <= d->maxcode[l]) return d->vals[(d->valoff[l] + code) & 255];
  }
  return -1;
}
static int br_receive_extend(uo_bitr_t* r, int s) {
  int v = 0;
  for (int i = 0; i < s; i++) v = (v << 1) | br_bit(r);
  return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; /* HUFF_EXTEND */
}
int uo_huffman_decode_scan(const uo_scan_t* sc, const uint8_t dht_bits[4][17], const uint8_t dht_vals[4][256], const uint8_t* data,
                           size_t size, int16_t* coef_out[3]) {
  uo_huff_dec_t dec[4];
  for (int t = 0; t < 4; t++) make_decoder(dht_bits[t], dht_vals[t], &dec[t]);
  int mpr, mrows;
  scan_geometry(sc, &mpr, &mrows);
  const int total = mpr * mrows, ri = sc->restart_interval > 0 ? sc->restart_interval : total;
  for (int c = 0; c < sc->ncomp; c++) memset(coef_out[c], 0, (size_t)sc->bw[c] * sc->bh[c] * 64 * sizeof(int16_t));
  size_t pos = 0;
  int expect = 0;
  for (int m0 = 0; m0 < total; m0 += ri) {
    /* the interval ends at the next marker (or at the end of the data) */
    size_t e = pos;
    while (e + 1 < size && !(data[e] == 0xff && data[e + 1] != 0)) e++;
    if (e + 1 >= size) e = size;
    uo_bitr_t r = {data + pos, data + e, 0, 0};
    int last_dc[3] = {0, 0, 0};
    for (int m = m0; m < m0 + ri && m < total; m++) {
      const int my = m / mpr, mx = m % mpr;
      for (int c = 0; c < sc->ncomp; c++) {
        const int hs = sc->ncomp == 1 ? 1 : sc->hs[c], vs = sc->ncomp == 1 ? 1 : sc->vs[c];
        const uo_huff_dec_t *dct = &dec[c ? 2 : 0], *act = &dec[c ? 3 : 1];
        for (int yi = 0; yi < vs; yi++) {
          for (int xi = 0; xi < hs; xi++) {
            const int by = my * vs + yi, bx = mx * hs + xi;
            int16_t scratch[64];
            int16_t* blk = (by < sc->bh[c] && bx < sc->bw[c]) ? coef_out[c] + ((size_t)by * sc->bw[c] + bx) * 64 : scratch;
            int s = br_symbol(&r, dct);
            if (s < 0 || s > 15) return -2;
            const int diff = s ? br_receive_extend(&r, s) : 0;
            last_dc[c] += diff;
            blk[0] = (int16_t)last_dc[c];
            for (int k = 1; k < 64;) {
              const int rs = br_symbol(&r, act);
              if (rs < 0) return -3;
              const int run = rs >> 4;
              s = rs & 15;
              if (s) {
                k += run;
                if (k > 63) return -4;
                blk[kZigzagToNatural[k]] = (int16_t)br_receive_extend(&r, s);
                k++;
              } else if (run == 15) {
                k += 16;
              } else {
                break;
              }
            }
          }
        }
      }
    }
    if (e < size) { /* RSTn */
      if (data[e + 1] != 0xd0 + expect) return -5;
      expect = (expect + 1) & 7;
      pos = e + 2;
    } else {
      pos = size;
    }
  }
  return 0;
}
