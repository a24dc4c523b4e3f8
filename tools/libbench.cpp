// Times uhdr_hip_apply_gainmap_dev through the C ABI (libuhdr_hip.so) with plain hipMalloc buffers: the same
// launches bench.py's extras make, without Python or torch's allocator in the picture.  One event pair around N
// back-to-back calls on the context's stream, rotating over NSET buffer sets.
//   hipcc --offload-arch=gfx950 -O2 -I../include -o libbench libbench.cpp -L../libultrahdr_amd/lib -luhdr_hip -Wl,-rpath,'$ORIGIN/../libultrahdr_amd/lib'
//   ./libbench [A|B|C] [w h] [slab] [nset=N] [streams=2]      slab: the planes of a set are carved out of one allocation, 512-byte aligned;
//   streams=2: two contexts (two HIP streams), launches alternate between them -- the tail of one launch overlaps the ramp of the next
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "uhdr_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

int main(int argc, char** argv) {
  const char* which = argc > 1 ? argv[1] : "C";
  const bool dims = argc > 3 && atoi(argv[2]) > 0 && atoi(argv[3]) > 0;
  const uint32_t w = dims ? atoi(argv[2]) : 7680, h = dims ? atoi(argv[3]) : 4320;
  bool slab = false;
  int nset = 3, nstreams = 1, pad = 0;  // pad: extra pixels per row of every plane (row pitch = width + pad)
  for (int i = 1; i < argc; i++) { if (!strcmp(argv[i], "slab")) slab = true; if (!strncmp(argv[i], "nset=", 5)) nset = atoi(argv[i] + 5); if (!strncmp(argv[i], "streams=", 8)) nstreams = atoi(argv[i] + 8); if (!strncmp(argv[i], "pad=", 4)) pad = atoi(argv[i] + 4); }
  const int mapfmt = which[0] == 'A' ? 0 : (which[0] == 'B' ? 1 : 2);
  const uint32_t scale = mapfmt == 0 ? 4 : 1, mw = w / scale, mh = h / scale, bpp = mapfmt == 0 ? 1 : (mapfmt == 1 ? 3 : 4);
  std::vector<uint8_t> y((size_t)w * h), u((size_t)w * h / 4), v((size_t)w * h / 4), m((size_t)mw * mh * bpp);
  srand(1);
  for (size_t i = 0; i < y.size(); i++) { size_t yy = i / w, xx = i % w; y[i] = (uint8_t)(128 + 100 * sinf(xx / 97.f) * cosf(yy / 61.f) + (rand() % 11) - 5); }
  for (size_t i = 0; i < u.size(); i++) { u[i] = 128 + (i % 31); v[i] = 128 - (i % 17); }
  for (size_t i = 0; i < m.size(); i++) m[i] = (uint8_t)(128 + 90 * sinf((i % (mw * bpp)) / 50.f) + rand() % 7);

  uhdr_error_info_t err;
  uhdr_hip_ctx_t* ctx = uhdr_hip_create(0, &err);
  if (!ctx) { fprintf(stderr, "uhdr_hip_create: %s\n", err.detail); return 1; }
  hipStream_t st = (hipStream_t)uhdr_hip_get_stream(ctx);
  uhdr_hip_ctx_t* ctx2 = nstreams > 1 ? uhdr_hip_create(0, &err) : nullptr;
  hipStream_t st2 = ctx2 ? (hipStream_t)uhdr_hip_get_stream(ctx2) : nullptr;

  std::vector<uhdr_raw_image_t> sdr(nset), gm(nset), dst(nset);
  auto up512 = [](size_t n) { return (n + 511) & ~(size_t)511; };
  for (int s = 0; s < nset; s++) {
    uint8_t *dy, *du, *dv, *dm, *dd;
    if (slab) {
      uint8_t* base;
      CK(hipMalloc(&base, up512(y.size()) + up512(u.size()) + up512(v.size()) + up512(m.size()) + (size_t)w * h * 8));
      dy = base; du = dy + up512(y.size()); dv = du + up512(u.size()); dm = dv + up512(v.size()); dd = dm + up512(m.size());
    } else {
      CK(hipMalloc(&dy, (size_t)(w + pad) * h)); CK(hipMalloc(&du, (size_t)(w / 2 + pad / 2) * h / 2)); CK(hipMalloc(&dv, (size_t)(w / 2 + pad / 2) * h / 2)); CK(hipMalloc(&dm, (size_t)(mw + pad) * mh * bpp)); CK(hipMalloc(&dd, (size_t)(w + pad) * h * 8));
    }
    CK(hipMemcpy2D(dy, w + pad, y.data(), w, w, h, hipMemcpyHostToDevice)); CK(hipMemcpy2D(du, w / 2 + pad / 2, u.data(), w / 2, w / 2, h / 2, hipMemcpyHostToDevice));
    CK(hipMemcpy2D(dv, w / 2 + pad / 2, v.data(), w / 2, w / 2, h / 2, hipMemcpyHostToDevice)); CK(hipMemcpy2D(dm, (size_t)(mw + pad) * bpp, m.data(), (size_t)mw * bpp, (size_t)mw * bpp, mh, hipMemcpyHostToDevice));
    memset(&sdr[s], 0, sizeof sdr[s]); memset(&gm[s], 0, sizeof gm[s]); memset(&dst[s], 0, sizeof dst[s]);
    sdr[s].fmt = UHDR_IMG_FMT_12bppYCbCr420; sdr[s].cg = UHDR_CG_BT_709; sdr[s].ct = UHDR_CT_SRGB; sdr[s].range = UHDR_CR_FULL_RANGE;
    sdr[s].w = w; sdr[s].h = h; sdr[s].planes[0] = dy; sdr[s].planes[1] = du; sdr[s].planes[2] = dv;
    sdr[s].stride[0] = w + pad; sdr[s].stride[1] = sdr[s].stride[2] = w / 2 + pad / 2;
    gm[s].fmt = mapfmt == 0 ? UHDR_IMG_FMT_8bppYCbCr400 : (mapfmt == 1 ? UHDR_IMG_FMT_24bppRGB888 : UHDR_IMG_FMT_32bppRGBA8888);
    gm[s].cg = UHDR_CG_BT_2100; gm[s].w = mw; gm[s].h = mh; gm[s].planes[0] = dm; gm[s].stride[0] = mw + pad;
    dst[s].fmt = UHDR_IMG_FMT_64bppRGBAHalfFloat; dst[s].w = w; dst[s].h = h; dst[s].planes[0] = dd; dst[s].stride[0] = w + pad;
  }
  uhdr_gainmap_metadata_t md;
  memset(&md, 0, sizeof md);
  for (int i = 0; i < 3; i++) { md.max_content_boost[i] = 4.926108f; md.min_content_boost[i] = 1.0f; md.gamma[i] = 1.0f; md.offset_sdr[i] = 1e-7f; md.offset_hdr[i] = 1e-7f; }
  md.hdr_capacity_min = 1.0f; md.hdr_capacity_max = 4.926108f; md.use_base_cg = 0;

  int flip = 0;
  auto call = [&](int s) {
    uhdr_hip_ctx_t* cx = (ctx2 && (flip++ & 1)) ? ctx2 : ctx;
    uhdr_error_info_t e = uhdr_hip_apply_gainmap_dev(cx, &sdr[s], &gm[s], &md, UHDR_CT_LINEAR, UHDR_IMG_FMT_64bppRGBAHalfFloat, 3.4e38f, &dst[s], 0, 0);
    if (e.error_code != UHDR_CODEC_OK) { fprintf(stderr, "apply: %s\n", e.detail); exit(1); }
  };
  hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
  for (int i = 0; i < 6; i++) call(i % nset);
  CK(hipStreamSynchronize(st));
  if (st2) CK(hipStreamSynchronize(st2));
  const int N = 30;
  double best = 1e30, sum = 0;
  for (int rep = 0; rep < 3; rep++) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; i++) call(i % nset);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (st2) {  // the region ends when both streams are done
      CK(hipEventRecord(e2, st2));
      CK(hipEventSynchronize(e2));
      float ms2; CK(hipEventElapsedTime(&ms2, e0, e2));
      if (ms2 > ms) ms = ms2;
    }
    const double us = ms * 1e3 / N;
    sum += us; if (us < best) best = us;
  }
  const double bytes = (1.5 + (double)bpp / (scale * scale) + 8) * w * h, us = sum / 3;
  printf("libbench map %s %ux%u %s nset=%d streams=%d pad=%d: mean %.1f us (best %.1f)  %.0f GB/s (%.1f%% of 8 TB/s)\n", which, w, h, slab ? "slab" : "separate", nset, nstreams, pad, us, best,
         bytes / us / 1e3, bytes / us / 1e3 / 80.0);
  uhdr_hip_destroy(ctx);
  return 0;
}
