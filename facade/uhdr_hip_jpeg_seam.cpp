// uhdr_hip_jpeg_seam.cpp -- see uhdr_hip_jpeg_seam.h.  libjpeg errors longjmp to the setjmp the reference's helper
// armed around the call site, exactly as they do for its own libjpeg calls; the scratch buffers here are plain
// malloc blocks owned by one struct so that nothing with a destructor sits between the two.
#include "uhdr_hip_jpeg_seam.h"

#include <cstdlib>
#include <cstring>

#include "uhdr_hip_seam.h"
#include "uhdr_hip.h"

namespace uhdr_hip_seam {

namespace {

struct Scratch {
  void* p[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int n = 0;
  void* get(size_t bytes) {
    void* q = nullptr;
    if (posix_memalign(&q, 64, bytes ? bytes : 64) != 0) return nullptr;
    p[n++] = q;
    return q;
  }
  void drop() {
    for (int i = 0; i < n; i++) free(p[i]);
    n = 0;
  }
};

uhdr_error_info_t mem_error() {
  uhdr_error_info_t s;
  memset(&s, 0, sizeof s);
  s.error_code = UHDR_CODEC_MEM_ERROR;
  s.has_detail = 1;
  snprintf(s.detail, sizeof s.detail, "uhdr_hip_seam: out of host memory for the coefficient arrays");
  return s;
}

unsigned ceil_div(unsigned a, unsigned b) { return (a + b - 1) / b; }

}  // namespace

bool jpeg_compress_on_device(jpeg_compress_struct* cinfo, const unsigned char* planes[3], const unsigned int strides[3],
                             uhdr_img_fmt_t format, const void* icc, size_t icc_size, const char* comment,
                             uhdr_error_info_t* st) {
  if (!enabled()) return false;
  const int nc = cinfo->num_components;
  if (nc != 1 && nc != 3) return false;
  int max_h = 1, max_v = 1;
  for (int c = 0; c < nc; c++) {
    if (cinfo->comp_info[c].h_samp_factor > max_h) max_h = cinfo->comp_info[c].h_samp_factor;
    if (cinfo->comp_info[c].v_samp_factor > max_v) max_v = cinfo->comp_info[c].v_samp_factor;
  }
  unsigned pw[3] = {0, 0, 0}, ph[3] = {0, 0, 0}, bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
  for (int c = 0; c < nc; c++) {
    pw[c] = ceil_div(cinfo->image_width * cinfo->comp_info[c].h_samp_factor, max_h);
    ph[c] = ceil_div(cinfo->image_height * cinfo->comp_info[c].v_samp_factor, max_v);
    // whole blocks only: partial edge blocks take libjpeg's (and the helper's) padding rules
    if (pw[c] % 8 || ph[c] % 8) return false;
    bw[c] = pw[c] / 8;
    bh[c] = ph[c] / 8;
    if (bw[c] % cinfo->comp_info[c].h_samp_factor || bh[c] % cinfo->comp_info[c].v_samp_factor) return false;  // no dummy blocks
  }
  const bool rgb = format == UHDR_IMG_FMT_24bppRGB888;
  if (rgb && nc != 3) return false;

  Scratch sc;
  const unsigned char* src[3] = {planes[0], planes[1], planes[2]};
  unsigned int sstride[3] = {strides[0], strides[1], strides[2]};
  if (rgb) {  // JCS_RGB -> YCbCr 4:4:4 (jccolor.c rgb_ycc_convert) on the device
    uhdr_raw_image_t in, ycc;
    memset(&in, 0, sizeof in);
    memset(&ycc, 0, sizeof ycc);
    in.fmt = UHDR_IMG_FMT_24bppRGB888;
    in.w = cinfo->image_width;
    in.h = cinfo->image_height;
    in.planes[0] = const_cast<unsigned char*>(planes[0]);
    in.stride[0] = strides[0];
    ycc.fmt = UHDR_IMG_FMT_24bppYCbCr444;
    ycc.w = in.w;
    ycc.h = in.h;
    for (int c = 0; c < 3; c++) {
      ycc.planes[c] = sc.get((size_t)pw[c] * ph[c]);
      if (!ycc.planes[c]) { sc.drop(); *st = mem_error(); return true; }
      ycc.stride[c] = pw[c];
      src[c] = static_cast<const unsigned char*>(ycc.planes[c]);
      sstride[c] = pw[c];
    }
    if (!jpeg_rgb_to_ycc(&in, &ycc, st)) { sc.drop(); return false; }
    if (st->error_code != UHDR_CODEC_OK) { sc.drop(); return true; }
  }

  short* coef[3] = {nullptr, nullptr, nullptr};
  const unsigned short* qt[3] = {nullptr, nullptr, nullptr};
  for (int c = 0; c < nc; c++) {
    coef[c] = static_cast<short*>(sc.get((size_t)bw[c] * bh[c] * 64 * sizeof(short)));
    if (!coef[c]) { sc.drop(); *st = mem_error(); return true; }
    JQUANT_TBL* tbl = cinfo->quant_tbl_ptrs[cinfo->comp_info[c].quant_tbl_no];
    if (!tbl) { sc.drop(); return false; }
    qt[c] = tbl->quantval;  // natural order (jpeg_add_quant_table)
  }
  if (!fdct_planes(nc, src, sstride, bw, bh, qt, coef, st)) { sc.drop(); return false; }
  if (st->error_code != UHDR_CODEC_OK) { sc.drop(); return true; }

  // hand the blocks to libjpeg: virtual arrays are requested before jpeg_write_coefficients() realizes them
  jvirt_barray_ptr arrays[3] = {nullptr, nullptr, nullptr};
  for (int c = 0; c < nc; c++)
    arrays[c] = (*cinfo->mem->request_virt_barray)((j_common_ptr)cinfo, JPOOL_IMAGE, FALSE, bw[c], bh[c],
                                                   (JDIMENSION)cinfo->comp_info[c].v_samp_factor);
#if JPEG_LIB_VERSION >= 80
  // what jpeg_copy_critical_parameters() would carry over for a transcode (IJG 8+: the transcoder does not derive them)
  cinfo->jpeg_width = cinfo->image_width;
  cinfo->jpeg_height = cinfo->image_height;
  cinfo->min_DCT_h_scaled_size = DCTSIZE;
  cinfo->min_DCT_v_scaled_size = DCTSIZE;
#endif
  jpeg_write_coefficients(cinfo, arrays);  // SOI + JFIF now; frame / scan headers with the data (jctrans.c)
  if (icc != nullptr && icc_size > 0) jpeg_write_marker(cinfo, JPEG_APP0 + 2, static_cast<const JOCTET*>(icc), (unsigned int)icc_size);
  if (comment) jpeg_write_marker(cinfo, JPEG_COM, reinterpret_cast<const JOCTET*>(comment), (unsigned int)strlen(comment));
  for (int c = 0; c < nc; c++) {
    for (unsigned by = 0; by < bh[c]; by++) {
      JBLOCKARRAY rows = (*cinfo->mem->access_virt_barray)((j_common_ptr)cinfo, arrays[c], by, 1, TRUE);
      memcpy(rows[0], coef[c] + (size_t)by * bw[c] * 64, (size_t)bw[c] * 64 * sizeof(short));
    }
  }
  jpeg_finish_compress(cinfo);  // the Huffman pass
  sc.drop();
  memset(st, 0, sizeof *st);
  return true;
}

bool jpeg_decompress_on_device(jpeg_decompress_struct* cinfo, bool want_rgb, unsigned char* dest,
                               const unsigned int hstride[3], const unsigned int vstride[3], uhdr_img_fmt_t* out_fmt,
                               uhdr_error_info_t* st) {
  if (!enabled()) return false;
  const int nc = cinfo->num_components;
  if (nc != 1 && nc != 3) return false;
  if (cinfo->progressive_mode || cinfo->arith_code) return false;
  if (nc == 3 && cinfo->jpeg_color_space != JCS_YCbCr) return false;
  if (nc == 1 && cinfo->jpeg_color_space != JCS_GRAYSCALE) return false;
  if (want_rgb && nc != 3) return false;
  const int mh = cinfo->max_h_samp_factor, mv = cinfo->max_v_samp_factor;
  uhdr_img_fmt_t fmt = UHDR_IMG_FMT_8bppYCbCr400;
  if (nc == 3) {
    const jpeg_component_info* ci = cinfo->comp_info;
    if (ci[0].h_samp_factor != mh || ci[0].v_samp_factor != mv || ci[1].h_samp_factor != ci[2].h_samp_factor ||
        ci[1].v_samp_factor != ci[2].v_samp_factor || ci[1].h_samp_factor != 1 || ci[1].v_samp_factor != 1)
      return false;
    if (mh == 1 && mv == 1) fmt = UHDR_IMG_FMT_24bppYCbCr444;
    else if (mh == 2 && mv == 1) fmt = UHDR_IMG_FMT_16bppYCbCr422;
    else if (mh == 2 && mv == 2) fmt = UHDR_IMG_FMT_12bppYCbCr420;
    else return false;
    if (want_rgb && fmt != UHDR_IMG_FMT_24bppYCbCr444) return false;  // libjpeg's fancy upsampling stays libjpeg's
  }

  jvirt_barray_ptr* arrays = jpeg_read_coefficients(cinfo);  // the Huffman decode of the whole scan
  Scratch sc;
  unsigned bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
  short* coef[3] = {nullptr, nullptr, nullptr};
  const unsigned short* qt[3] = {nullptr, nullptr, nullptr};
  unsigned char* pl[3] = {nullptr, nullptr, nullptr};
  unsigned int ps[3] = {0, 0, 0};
  for (int c = 0; c < nc; c++) {
    const jpeg_component_info* ci = &cinfo->comp_info[c];
    bw[c] = ci->width_in_blocks;
    bh[c] = ci->height_in_blocks;
    JQUANT_TBL* tbl = ci->quant_table ? ci->quant_table : cinfo->quant_tbl_ptrs[ci->quant_tbl_no];
    if (!tbl) { sc.drop(); memset(st, 0, sizeof *st); st->error_code = UHDR_CODEC_ERROR; st->has_detail = 1;
                snprintf(st->detail, sizeof st->detail, "component %d has no quantization table", c); return true; }
    qt[c] = tbl->quantval;
    coef[c] = static_cast<short*>(sc.get((size_t)bw[c] * bh[c] * 64 * sizeof(short)));
    pl[c] = static_cast<unsigned char*>(sc.get((size_t)bw[c] * 8 * bh[c] * 8));
    if (!coef[c] || !pl[c]) { sc.drop(); *st = mem_error(); return true; }
    ps[c] = bw[c] * 8;
    for (unsigned by = 0; by < bh[c]; by++) {
      JBLOCKARRAY rows = (*cinfo->mem->access_virt_barray)((j_common_ptr)cinfo, arrays[c], by, 1, FALSE);
      memcpy(coef[c] + (size_t)by * bw[c] * 64, rows[0], (size_t)bw[c] * 64 * sizeof(short));
    }
  }
  const short* ccoef[3] = {coef[0], coef[1], coef[2]};
  if (!idct_planes(nc, ccoef, bw, bh, qt, pl, ps, st)) {
    // the coefficients are already out of libjpeg: there is no way back to its own IDCT from here
    sc.drop();
    return true;
  }
  if (st->error_code != UHDR_CODEC_OK) { sc.drop(); return true; }

  if (!want_rgb) {
    size_t off = 0;
    for (int c = 0; c < nc; c++) {
      unsigned char* d = dest + off;
      off += (size_t)hstride[c] * vstride[c];
      const unsigned rows = vstride[c] < bh[c] * 8 ? vstride[c] : bh[c] * 8;
      const unsigned cols = hstride[c] < ps[c] ? hstride[c] : ps[c];
      for (unsigned y = 0; y < rows; y++) memcpy(d + (size_t)y * hstride[c], pl[c] + (size_t)y * ps[c], cols);
    }
    *out_fmt = fmt;
  } else {
    uhdr_raw_image_t ycc, out;
    memset(&ycc, 0, sizeof ycc);
    memset(&out, 0, sizeof out);
    ycc.fmt = UHDR_IMG_FMT_24bppYCbCr444;
    ycc.w = cinfo->image_width;
    ycc.h = cinfo->image_height;
    for (int c = 0; c < 3; c++) { ycc.planes[c] = pl[c]; ycc.stride[c] = ps[c]; }
#ifdef JCS_ALPHA_EXTENSIONS
    out.fmt = UHDR_IMG_FMT_32bppRGBA8888;   // libjpeg-turbo: the helper asks for JCS_EXT_RGBA
    const int variant = 0;
#else
    out.fmt = UHDR_IMG_FMT_24bppRGB888;
    const int variant = JPEG_LIB_VERSION >= 90 ? 1 : 0;  // IJG 9 refined the green-term constants (jdcolor.c)
#endif
    out.w = ycc.w;
    out.h = ycc.h;
    out.planes[0] = dest;
    out.stride[0] = hstride[0];
    jpeg_ycc_to_rgb(&ycc, variant, &out, st);
    *out_fmt = out.fmt;
    if (st->error_code != UHDR_CODEC_OK) { sc.drop(); return true; }
  }
  jpeg_finish_decompress(cinfo);
  sc.drop();
  memset(st, 0, sizeof *st);
  return true;
}

}  // namespace uhdr_hip_seam
