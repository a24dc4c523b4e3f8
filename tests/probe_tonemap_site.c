/* The three tone-map samples that tests/probe_tonemap_hunt.py found (HIP one code below the reference): the oracle's per-pixel pipeline with glibc's powf and
   with a correctly rounded pow in srgbOetf.   cd tests && gcc -O2 -o /tmp/tm_site probe_tonemap_site.c -lm && /tmp/tm_site     (tests/test_tonemap_site.py runs it) */
#include "../oracle/uhdr_oracle.c"  /* test infrastructure: this probe runs on the CPU only */
#include <stdio.h>
static float srgb_oetf_cr(float e) { /* the same formula with a correctly rounded pow (long double, then rounded once) */
  if (e <= 0.0031308f) return 12.92f * e;
  float p = (float)powl((long double)e, (long double)(1.0f / 2.4f));
  return 1.055f * p - 0.055f;
}
static void run(int ct, int cg, int range, const uint16_t ysmp[4], uint16_t u, uint16_t v) {
  uo_image_t hdr; memset(&hdr, 0, sizeof hdr);
  uint16_t yb[4] = {ysmp[0], ysmp[1], ysmp[2], ysmp[3]}, cb[2] = {u, v};
  hdr.fmt = UO_FMT_P010; hdr.cg = cg; hdr.ct = ct; hdr.range = range; hdr.w = 2; hdr.h = 2;
  hdr.planes[0] = yb; hdr.planes[1] = cb; hdr.stride[0] = 2; hdr.stride[1] = 2;
  int ok; const float* gm = gamut_matrix(UO_CG_P3, cg, &ok);
  const yuv2rgb_t y2r = yuv2rgb_coeffs(cg); const rgb2yuv_t p3 = rgb2yuv_coeffs(UO_CG_P3);
  const float headroom = ref_peak_nits(ct) / kSdrWhiteNits; const int is_norm = ct != UO_CT_LINEAR;
  for (int variant = 0; variant < 2; variant++) {
    float su = 0, sv = 0; int yo[4];
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) {
      color_t g = get_pixel(&hdr, j, i); g = yuv_to_rgb(g, &y2r);
      color_t l = inv_oetf(g, ct); if (ct == UO_CT_HLG) l = hlg_ootf_approx(l);
      float c[3] = {l.r, l.g, l.b}; if (is_norm) { c[0] *= headroom; c[1] *= headroom; c[2] *= headroom; }
      float mx = c[0]; if (c[1] > mx) mx = c[1]; if (c[2] > mx) mx = c[2];
      float ms = 1.0f + mx / (headroom * headroom); ms /= 1.0f + mx; ms = ms * mx;
      color_t o; o.r = c[0] > 0 ? c[0] * ms / mx : 0; o.g = c[1] > 0 ? c[1] * ms / mx : 0; o.b = c[2] > 0 ? c[2] * ms / mx : 0;
      o = gamut_conv(o, gm); o.r = clamp01(o.r); o.g = clamp01(o.g); o.b = clamp01(o.b);
      color_t og; 
      if (variant == 0) { og.r = srgb_oetf(o.r); og.g = srgb_oetf(o.g); og.b = srgb_oetf(o.b); }
      else { og.r = srgb_oetf_cr(o.r); og.g = srgb_oetf_cr(o.g); og.b = srgb_oetf_cr(o.b); }
      if (variant == 1 || 1) printf("  v%d px(%d,%d) linear %.9g %.9g %.9g -> oetf %.9g %.9g %.9g (bits %08x %08x %08x)\n", variant, i, j, o.r, o.g, o.b, og.r, og.g, og.b, *(unsigned*)&og.r, *(unsigned*)&og.g, *(unsigned*)&og.b);
      color_t yuv = rgb_to_yuv(og, &p3); yuv.g += 0.5f; yuv.b += 0.5f;
      yo[i * 2 + j] = scale_to_8bit(yuv.r); su += yuv.g; sv += yuv.b;
      printf("     Y*255 = %.7f\n", (double)(yuv.r * 255.0f));
    }
    su /= 4.0f; sv /= 4.0f;
    printf(" variant %s: Y %d %d %d %d  U %d (%.7f) V %d (%.7f)\n", variant ? "correctly rounded pow" : "glibc powf", yo[0], yo[1], yo[2], yo[3], scale_to_8bit(su), (double)(su * 255.0f), scale_to_8bit(sv), (double)(sv * 255.0f));
  }
}
int main() {
  init_luts();
  { uint16_t y[4] = {15872, 15872, 15872, 15872}; printf("case A (luma, hip 30 / ref 31), sample 0\n"); run(2, 2, 0, y, 34496, 32000); }
  { uint16_t y[4] = {31680, 29952, 32960, 31104}; printf("case B (Cr, hip 191 / ref 192)\n"); run(2, 2, 0, y, 29952, 37056); }
  { uint16_t y[4] = {46080, 45888, 43712, 46976}; printf("case C (Cr, hip 179 / ref 180)\n"); run(1, 2, 0, y, 34240, 37504); }
  return 0;
}
