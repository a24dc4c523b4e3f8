// Standalone timing harness for the applyGainMap kernels (no Python): builds the kernel TU with
// optional -DUHDR_EXP_* experiment macros and times 8K launches.  tools/kbench.sh builds variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "apply_gainmap.hip"
#include "host_tables.cpp"

using namespace uhdr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
  const uint32_t w = 7680, h = 4320;
  const char* which = argc > 1 ? argv[1] : "A";
  const bool noise = !(argc > 2 && !strcmp(argv[2], "smooth"));
  const bool white = argc > 2 && !strcmp(argv[2], "white");  // white-noise map bytes: the worst case for the factor-table gathers
  const bool gauss = argc > 2 && !strcmp(argv[2], "gauss");  // Gaussian noise like libultrahdr_amd/synth.py: sigma 5 levels on luma, 7.6 on the map
  auto gn = [&](float sigma) { float a = 0; for (int k = 0; k < 12; k++) a += rand() / (float)RAND_MAX; return (a - 6.0f) * sigma; };
  const int mapfmt = which[0] == 'A' ? 0 : (which[0] == 'B' ? 1 : 2);
  const uint32_t scale = mapfmt == 0 ? 4 : 1, mw = w / scale, mh = h / scale, bpp = mapfmt == 0 ? 1 : (mapfmt == 1 ? 3 : 4);
  std::vector<uint8_t> y((size_t)w * h), u((size_t)w * h / 4), v((size_t)w * h / 4), m((size_t)mw * mh * bpp);
  srand(1);
  for (size_t i = 0; i < y.size(); i++) { size_t yy = i / w, xx = i % w; float v = 128 + 100 * sinf(xx / 97.f) * cosf(yy / 61.f) + (gauss ? gn(5.0f) : (noise ? (rand() % 11) - 5 : 0)); y[i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
  for (size_t i = 0; i < u.size(); i++) { u[i] = 128 + (i % 31); v[i] = 128 - (i % 17); }
  for (size_t i = 0; i < m.size(); i++) { float v = 128 + 90 * sinf((i % (mw * bpp)) / 50.f) + (gauss ? gn(7.6f) : (noise ? rand() % 7 : 0)); m[i] = white ? (uint8_t)(rand() & 255) : (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
  const int NSET = getenv("KB_NSET") ? atoi(getenv("KB_NSET")) : 2;
  uint8_t *dy[8], *du[8], *dv[8], *dm[8], *dd[8];
  for (int s = 0; s < NSET; s++) {
    CK(hipMalloc(&dy[s], y.size())); CK(hipMalloc(&du[s], u.size())); CK(hipMalloc(&dv[s], v.size())); CK(hipMalloc(&dm[s], m.size())); CK(hipMalloc(&dd[s], (size_t)w * h * 8));
    CK(hipMemcpy(dy[s], y.data(), y.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(du[s], u.data(), u.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(dv[s], v.data(), v.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dm[s], m.data(), m.size(), hipMemcpyHostToDevice));
  }
  uhdr_gainmap_metadata_t md;
  for (int i = 0; i < 3; i++) { md.max_content_boost[i] = 4.926108f; md.min_content_boost[i] = 1.0f; md.gamma[i] = 1.0f; md.offset_sdr[i] = 1e-7f; md.offset_hdr[i] = 1e-7f; }
  md.hdr_capacity_min = 1.0f; md.hdr_capacity_max = 4.926108f; md.use_base_cg = 0;
  std::vector<float> tab;
  host::build_apply_tables(md, 1.0f, scale, &tab);
  float* dtab; CK(hipMalloc(&dtab, tab.size() * 4)); CK(hipMemcpy(dtab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
  ApplyParams p; memset(&p, 0, sizeof p);
  bool ident; host::gamut_matrix(UHDR_CG_BT_2100, UHDR_CG_BT_709, &p.gamut, &ident);
  p.sdr_gamut_on = getenv("KB_NOGAMUT") ? 0 : 1;  // KB_NOGAMUT: the slope of launch time against VALU work (15 packed ops per pixel pair less)
  p.tables = dtab; p.scale = scale; p.scale_magic = scale > 1 ? (uint32_t)((0x100000000ull + scale - 1) / scale) : 0; p.scale_f = scale;
  p.map_bpp = bpp; p.map_ch = mapfmt == 0 ? 1 : 3; p.out_ct = UHDR_CT_LINEAR;
  for (int i = 0; i < 3; i++) { p.gamma_is_one[i] = 1; p.gamma_inv[i] = 1; p.offset_sdr[i] = 1e-7f; p.offset_hdr[i] = 1e-7f; }
  p.yuv = host::yuv2rgb_coeffs(UHDR_CG_DISPLAY_P3);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&](int s) {
    p.sdr.p[0] = dy[s]; p.sdr.p[1] = du[s]; p.sdr.p[2] = dv[s]; p.sdr.stride[0] = w; p.sdr.stride[1] = p.sdr.stride[2] = w / 2; p.sdr.w = w; p.sdr.h = h; p.sdr.fmt = UHDR_IMG_FMT_12bppYCbCr420;
    p.gm.p[0] = dm[s]; p.gm.stride[0] = mw; p.gm.w = mw; p.gm.h = mh; p.gm.fmt = mapfmt == 0 ? UHDR_IMG_FMT_8bppYCbCr400 : (mapfmt == 1 ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_32bppRGBA8888);
    p.dst.p[0] = dd[s]; p.dst.stride[0] = w; p.dst.w = w; p.dst.h = h; p.dst.fmt = UHDR_IMG_FMT_64bppRGBAHalfFloat;
    CK(launch_apply_gainmap(p, st));
  };
  for (int i = 0; i < 3; i++) launch(i % NSET);
  CK(hipStreamSynchronize(st));
  const int N = getenv("KB_N") ? atoi(getenv("KB_N")) : 10;
  for (int rep = 0; rep < (getenv("KB_REPS") ? atoi(getenv("KB_REPS")) : 1); rep++) {
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < N; i++) launch(i % NSET);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  double us = ms * 1e3 / N, bytes = (1.5 + (double)bpp / (scale * scale) + 8) * w * h;
  printf("%-28s map %s %s: %.1f us  %.0f GB/s (%.1f%% of 8 TB/s)\n", argv[0], which, white ? "white" : (gauss ? "gauss" : (noise ? "noisy" : "smooth")), us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
  }
  return 0;
}
