// uhdr_hip_jpeg_seam.cpp -- see uhdr_hip_jpeg_seam.h.  libjpeg errors longjmp to the setjmp the reference's helper
// armed around the call site, exactly as they do for its own libjpeg calls; the scratch buffers here live in the libjpeg
// object's image pool, so nothing with a destructor sits between the two and nothing leaks on that path.
#include "uhdr_hip_jpeg_seam.h"

#include <cstdlib>
#include <cstring>

#include "uhdr_hip_seam.h"
#include "uhdr_hip.h"

namespace uhdr_hip_seam {

namespace {

// Scratch blocks come from the libjpeg object's own JPOOL_IMAGE pool: an error_exit in any libjpeg call made while they are
// live (destination overflow, memory manager failure) longjmps to the helper's setjmp, and jpeg_destroy_* / jpeg_abort frees
// the pool on that path too.  alloc_large reports exhaustion through error_exit, like every libjpeg allocation.
struct Scratch {
  j_common_ptr ci;
  explicit Scratch(j_common_ptr c) : ci(c) {}
  // nullptr: a request beyond libjpeg's MAX_ALLOC_CHUNK (1e9 bytes in libjpeg-turbo; alloc_large would error_exit and fail the
  // whole uhdr_encode / uhdr_decode) -- the caller then leaves the image to the reference's row-by-row CPU path
  void* get(size_t bytes) {
    if (bytes > (size_t)900000000) return nullptr;
    return (*ci->mem->alloc_large)(ci, JPOOL_IMAGE, bytes ? bytes : 64);
  }
  void drop() {}  // the pool is released with the libjpeg object
};

unsigned ceil_div(unsigned a, unsigned b) { return (a + b - 1) / b; }

}  // namespace

namespace {

void emit(jpeg_compress_struct* cinfo, const unsigned char* p, size_t n) {  // through the helper's own destination manager
  jpeg_destination_mgr* d = cinfo->dest;
  while (n > 0) {
    if (d->free_in_buffer == 0) (*d->empty_output_buffer)(cinfo);
    const size_t k = n < d->free_in_buffer ? n : d->free_in_buffer;
    memcpy(d->next_output_byte, p, k);
    d->next_output_byte += k;
    d->free_in_buffer -= k;
    p += k;
    n -= k;
  }
}
void emit_marker(jpeg_compress_struct* cinfo, int code, const void* data, size_t n) {  // jpeg_write_marker's bytes
  const unsigned char h[4] = {0xff, (unsigned char)code, (unsigned char)((n + 2) >> 8), (unsigned char)((n + 2) & 0xff)};
  emit(cinfo, h, 4);
  emit(cinfo, static_cast<const unsigned char*>(data), n);
}

// The whole compressImage on the device, Huffman pass included (the default since round 3).  Without restart markers
// (restart_interval 0) the entropy-coded segment is the one libjpeg writes -- the file equals the reference's byte for byte.
// UHDR_HIP_SEAM_RESTART_INTERVAL=<MCUs | max> (also the older UHDR_HIP_SEAM_DEVICE_ENTROPY=1 = max) asks for restart
// intervals instead: a DRI segment and RSTn markers the reference's files do not have; every decoder reconstructs the same
// coefficients (T.81 B.2.4.4, F.1.3).  UHDR_HIP_SEAM_CPU_ENTROPY=1 leaves the Huffman pass to libjpeg.
// true: the file is written (*st = result); false: not a configuration for this path, nothing written.
bool device_scan_encode(jpeg_compress_struct* cinfo, int nc, const unsigned char* const planes[3], const unsigned int strides[3], bool rgb,
                        const unsigned bw[3], const unsigned bh[3], const void* icc, size_t icc_size, const char* comment,
                        uhdr_error_info_t* st) {
  if (cinfo->arith_code || cinfo->optimize_coding || cinfo->scan_info != nullptr || cinfo->data_precision != 8) return false;
  if (cinfo->restart_interval != 0 || cinfo->restart_in_rows != 0) return false;
  if (!cinfo->write_JFIF_header || cinfo->write_Adobe_marker) return false;  // the header writer below emits what jcmarker.c emits for JFIF files
  if (cinfo->jpeg_color_space != (nc == 3 ? JCS_YCbCr : JCS_GRAYSCALE)) return false;
  if (icc_size > 65533 || (comment && strlen(comment) > 65533) || cinfo->image_width > 65535 || cinfo->image_height > 65535) return false;
  uhdr_hip_jpeg_scan_t scan;
  memset(&scan, 0, sizeof scan);
  uint16_t qt[3][64];
  memset(qt, 0, sizeof qt);
  scan.num_components = nc;
  scan.w = cinfo->image_width;
  scan.h = cinfo->image_height;
  int bpm = 0;
  for (int c = 0; c < nc; c++) {
    const jpeg_component_info* ci = &cinfo->comp_info[c];
    if (ci->h_samp_factor < 1 || ci->h_samp_factor > 2 || ci->v_samp_factor < 1 || ci->v_samp_factor > 2) return false;
    if (ci->quant_tbl_no != (c ? 1 : 0) || ci->dc_tbl_no != (c ? 1 : 0) || ci->ac_tbl_no != (c ? 1 : 0)) return false;  // jpeg_set_defaults' assignment
    const JQUANT_TBL* q = cinfo->quant_tbl_ptrs[ci->quant_tbl_no];
    if (!q) return false;
    for (int i = 0; i < 64; i++) {
      if (q->quantval[i] == 0 || q->quantval[i] > 255) return false;  // baseline tables
      qt[c][i] = q->quantval[i];
    }
    scan.blocks_w[c] = (int)bw[c];
    scan.blocks_h[c] = (int)bh[c];
    scan.h_samp[c] = nc == 1 ? 1 : ci->h_samp_factor;
    scan.v_samp[c] = nc == 1 ? 1 : ci->v_samp_factor;
    bpm += scan.h_samp[c] * scan.v_samp[c];
  }
  scan.restart_interval = 0;  // the reference's stream (jpegencoderhelper.cpp:187-201 never sets restart_interval)
  const char* ri_env = getenv("UHDR_HIP_SEAM_RESTART_INTERVAL");
  if (!ri_env && getenv("UHDR_HIP_SEAM_DEVICE_ENTROPY")) ri_env = "max";
  if (ri_env) {
    const int longest = 64 / bpm;  // what one wavefront (64 blocks) holds
    const int v = strcmp(ri_env, "max") ? atoi(ri_env) : longest;
    scan.restart_interval = v < 0 ? 0 : (v > longest ? longest : v);
  }
  size_t cap = 1u << 16;
  for (int c = 0; c < nc; c++) cap += (size_t)bw[c] * bh[c] * 64;
  Scratch sc((j_common_ptr)cinfo);
  unsigned char* data = static_cast<unsigned char*>(sc.get(cap));
  if (!data) return false;  // larger than one libjpeg allocation: the CPU path handles it row by row
  size_t n = 0;
  uhdr_error_info_t r;
  bool on_device = encode_scan(&scan, qt, planes, strides, rgb ? 3 : 0, data, cap, &n, &r);
  if (on_device && r.error_code == UHDR_CODEC_MEM_ERROR && n > cap) {  // data busier than the raw samples: the call said how much it needs
    sc.drop();
    cap = n + n / 8 + 4096;  // the size reported for a stream that overflowed is an estimate of its stuffed size: leave slack
    data = static_cast<unsigned char*>(sc.get(cap));
    if (!data) return false;
    on_device = encode_scan(&scan, qt, planes, strides, rgb ? 3 : 0, data, cap, &n, &r);
  }
  if (!on_device) { sc.drop(); return false; }
  // a stream that still does not fit (near-saturated coefficients stuff many 0xFF bytes): libjpeg's encoder takes over
  if (r.error_code == UHDR_CODEC_MEM_ERROR) { sc.drop(); return false; }
  if (r.error_code != UHDR_CODEC_OK) { sc.drop(); *st = r; return true; }
  // headers: SOI, JFIF APP0 | the helper's markers | DQT, SOF0, DHT, DRI, SOS | data | EOI  (jcmarker.c's order)
  unsigned char hdr[2048];
  const unsigned char none = 0;
  const size_t hn = uhdr_hip_jpeg_assemble(&scan, qt[0], qt[nc == 3 ? 1 : 0], &none, 0, hdr, sizeof hdr);
  if (hn < 22) { sc.drop(); return false; }
  (*cinfo->dest->init_destination)(cinfo);
  emit(cinfo, hdr, 20);  // SOI + the 18-byte APP0
  if (icc != nullptr && icc_size > 0) emit_marker(cinfo, 0xe2, icc, icc_size);
  if (comment) emit_marker(cinfo, 0xfe, comment, strlen(comment));
  emit(cinfo, hdr + 20, hn - 22);
  emit(cinfo, data, n);
  emit(cinfo, hdr + hn - 2, 2);  // EOI
  (*cinfo->dest->term_destination)(cinfo);
  sc.drop();
  memset(st, 0, sizeof *st);
  return true;
}

}  // namespace

bool jpeg_compress_on_device(jpeg_compress_struct* cinfo, const unsigned char* planes[3], const unsigned int strides[3],
                             uhdr_img_fmt_t format, const void* icc, size_t icc_size, const char* comment,
                             uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
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
      bw[c] = ceil_div(pw[c], 8);  // the component's REAL blocks (jpeg_component_info::width_in_blocks / height_in_blocks)
      bh[c] = ceil_div(ph[c], 8);
    }
    const bool rgb = format == UHDR_IMG_FMT_24bppRGB888;
    if (rgb && nc != 3) return false;
    // Round 4: partial edge blocks (a 1920x1080 base image has 960x540 chroma planes, a 4K frame a 960x540 gain map) and the
    // dummy blocks that complete edge MCUs are made on the device by the helper's / libjpeg's own rules
    // (uhdr_hip_jpeg_encode_image), so every geometry takes the device route.
    if (!getenv("UHDR_HIP_SEAM_CPU_ENTROPY") && device_scan_encode(cinfo, nc, planes, strides, rgb, bw, bh, icc, icc_size, comment, st)) return true;
    // the route below (device FDCT -> jpeg_write_coefficients, libjpeg's Huffman pass) keeps whole-block planes only
    for (int c = 0; c < nc; c++) {
      if (pw[c] % 8 || ph[c] % 8) return false;
      if (bw[c] % cinfo->comp_info[c].h_samp_factor || bh[c] % cinfo->comp_info[c].v_samp_factor) return false;  // no dummy blocks
    }

    Scratch sc((j_common_ptr)cinfo);
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
        if (!ycc.planes[c]) { sc.drop(); return false; }
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
      if (!coef[c]) { sc.drop(); return false; }
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
  }();
  if (!on_device_) uhdr_hip_seam::drop_resident();
  return on_device_;
}

namespace {

// true: the device decoded the whole scan (*st = result).  false: not a file for the device path; nothing consumed.
bool device_scan_decode(jpeg_decompress_struct* cinfo, bool want_rgb, unsigned char* dest, const unsigned int hstride[3],
                        const unsigned int vstride[3], uhdr_img_fmt_t planar_fmt, uhdr_img_fmt_t* out_fmt, uhdr_error_info_t* st) {
  const int nc = cinfo->num_components;
  if (cinfo->progressive_mode || cinfo->arith_code || cinfo->data_precision != 8) return false;
  if (cinfo->comps_in_scan != nc || cinfo->Ss != 0 || cinfo->Se != 63 || cinfo->Ah != 0 || cinfo->Al != 0) return false;
  if (!cinfo->src || !cinfo->src->next_input_byte || cinfo->src->bytes_in_buffer < 2) return false;
  if (cinfo->image_width > 65535 || cinfo->image_height > 65535) return false;
  uhdr_hip_jpeg_header_t hdr;
  memset(&hdr, 0, sizeof hdr);
  hdr.scan.num_components = nc;
  hdr.scan.w = cinfo->image_width;
  hdr.scan.h = cinfo->image_height;
  hdr.scan.restart_interval = (int)cinfo->restart_interval;
  for (int c = 0; c < nc; c++) {
    const jpeg_component_info* ci = &cinfo->comp_info[c];
    if (cinfo->cur_comp_info[c] != ci) return false;  // scan order == frame order
    if (ci->h_samp_factor < 1 || ci->h_samp_factor > 2 || ci->v_samp_factor < 1 || ci->v_samp_factor > 2) return false;
    hdr.scan.blocks_w[c] = (int)ci->width_in_blocks;
    hdr.scan.blocks_h[c] = (int)ci->height_in_blocks;
    hdr.scan.h_samp[c] = nc == 1 ? 1 : ci->h_samp_factor;
    hdr.scan.v_samp[c] = nc == 1 ? 1 : ci->v_samp_factor;
    if (ci->quant_tbl_no < 0 || ci->quant_tbl_no >= NUM_QUANT_TBLS) return false;
    const JQUANT_TBL* q = cinfo->quant_tbl_ptrs[ci->quant_tbl_no];
    if (!q) return false;
    for (int i = 0; i < 64; i++) hdr.qtable[c][i] = q->quantval[i];
  }
  if (nc == 3 && (cinfo->comp_info[1].dc_tbl_no != cinfo->comp_info[2].dc_tbl_no || cinfo->comp_info[1].ac_tbl_no != cinfo->comp_info[2].ac_tbl_no))
    return false;  // the device decoder takes one (DC, AC) pair for component 0 and one for the other two
  const int second = nc == 3 ? 1 : 0;
  const JHUFF_TBL* ht[4] = {cinfo->dc_huff_tbl_ptrs[cinfo->comp_info[0].dc_tbl_no & 3], cinfo->ac_huff_tbl_ptrs[cinfo->comp_info[0].ac_tbl_no & 3],
                            cinfo->dc_huff_tbl_ptrs[cinfo->comp_info[second].dc_tbl_no & 3], cinfo->ac_huff_tbl_ptrs[cinfo->comp_info[second].ac_tbl_no & 3]};
  for (int t = 0; t < 4; t++) {
    if (!ht[t]) return false;
    memcpy(hdr.tables.bits[t], ht[t]->bits, 17);
    memcpy(hdr.tables.vals[t], ht[t]->huffval, 256);
  }
  int channels = 0, variant = 0;
  if (want_rgb) {
#ifdef JCS_ALPHA_EXTENSIONS
    channels = 4;  // libjpeg-turbo: the helper asks for JCS_EXT_RGBA
#else
    channels = 3;
    variant = JPEG_LIB_VERSION >= 90 ? 1 : 0;  // IJG 9 refined the green-term constants (jdcolor.c)
#endif
  }
  unsigned char* planes[3] = {nullptr, nullptr, nullptr};
  size_t off = 0;
  for (int c = 0; c < (want_rgb ? 1 : nc); c++) {
    planes[c] = dest + off;
    off += (size_t)hstride[c] * vstride[c];
  }
  uhdr_error_info_t r;
  if (!decode_scan(&hdr, cinfo->src->next_input_byte, cinfo->src->bytes_in_buffer, channels, variant, planes, hstride, vstride, &r)) return false;
  *st = r;
  if (r.error_code == UHDR_CODEC_OK)
    *out_fmt = want_rgb ? (channels == 4 ? UHDR_IMG_FMT_32bppRGBA8888 : UHDR_IMG_FMT_24bppRGB888) : planar_fmt;
  return true;
}

}  // namespace

bool jpeg_decompress_on_device(jpeg_decompress_struct* cinfo, bool want_rgb, unsigned char* dest,
                               const unsigned int hstride[3], const unsigned int vstride[3], uhdr_img_fmt_t* out_fmt,
                               uhdr_error_info_t* st) {
  // every exit that leaves the work to the reference's CPU code drops the device-resident copies: that code may write the
  // host buffers in place
  const bool on_device_ = [&]() -> bool {
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

    // Whole decode on the device when the file is what libjpeg writes by default (baseline, one interleaved scan, 8 bit):
    // the entropy-coded bytes go up, the samples come down.  Anything else -- progressive, arithmetic coding, several
    // scans, a source manager that does not hold the rest of the file -- takes jpeg_read_coefficients() below.
    if (!getenv("UHDR_HIP_SEAM_CPU_ENTROPY") && device_scan_decode(cinfo, want_rgb, dest, hstride, vstride, fmt, out_fmt, st)) return true;

    // everything that can still hand the image back to the reference's row-by-row CPU path has to be decided BEFORE
    // jpeg_read_coefficients: after it the object is in DSTATE_RDCOEFS and jpeg_start_decompress would raise JERR_BAD_STATE
    for (int c = 0; c < nc; c++) {
      const jpeg_component_info* ci = &cinfo->comp_info[c];
      if ((size_t)ci->width_in_blocks * ci->height_in_blocks * 64 * sizeof(short) > (size_t)900000000) return false;  // beyond one libjpeg allocation (Scratch::get)
    }
    jvirt_barray_ptr* arrays = jpeg_read_coefficients(cinfo);  // the Huffman decode of the whole scan
    Scratch sc((j_common_ptr)cinfo);
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
      if (!coef[c] || !pl[c]) {  // cannot happen after the size check above; if it does, it is an error, not a fallback (the coefficients are consumed)
        sc.drop();
        memset(st, 0, sizeof *st);
        st->error_code = UHDR_CODEC_MEM_ERROR;
        st->has_detail = 1;
        snprintf(st->detail, sizeof st->detail, "scratch for component %d of a %ux%u image exceeds one libjpeg allocation", c, cinfo->image_width, cinfo->image_height);
        return true;
      }
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
  }();
  if (!on_device_) uhdr_hip_seam::drop_resident();
  return on_device_;
}

}  // namespace uhdr_hip_seam
