// Host-side reading of a baseline JPEG's headers (no device work): everything uhdr_hip_huffman_decode_dev,
// uhdr_hip_idct_dequant_dev and uhdr_hip_apply_gainmap_coef_dev need to take a file from bytes to pixels.
//
// In the reference this is libjpeg's jdmarker.c behind jpeg_read_header (JpegDecoderHelper::decompressImage,
// /root/reference/lib/src/jpegdecoderhelper.cpp:212-222).  Restated from ITU-T T.81 Annex B: SOI, then marker
// segments -- DQT (B.2.4.1, zig-zag order in the file), SOF0 (B.2.2), DHT (B.2.4.2), DRI (B.2.4.4), SOS (B.2.3) --
// then entropy-coded data up to the first marker that is neither a stuffed zero nor RSTn.
// Accepted: baseline sequential (SOF0), 8-bit samples, 1 or 3 components, one scan holding all components,
// sampling factors 1 or 2, components 1 and 2 sharing one pair of Huffman tables (every file libjpeg writes with its
// default settings, hence every base image and gain map of an UltraHDR file).  Anything else is reported, not guessed.
#include <cstring>

#include "host_tables.h"
#include "uhdr_hip.h"

#pragma GCC visibility push(default)
extern "C" {

int uhdr_hip_jpeg_parse(const uint8_t* file, size_t size, uhdr_hip_jpeg_header_t* out) {
  if (!file || !out || size < 4) return -1;
  memset(out, 0, sizeof *out);
  if (file[0] != 0xff || file[1] != 0xd8) return -2;  // SOI
  const uint8_t* zz = uhdr::host::jpeg_zigzag_to_natural();
  uint16_t qt[4][64];
  bool have_qt[4] = {false, false, false, false};
  uint8_t hbits[2][4][17], hvals[2][4][256];  // [class][id]
  bool have_ht[2][4] = {{false, false, false, false}, {false, false, false, false}};
  int comp_id[3] = {0, 0, 0}, comp_tq[3] = {0, 0, 0};
  bool have_sof = false;
  size_t i = 2;
  while (i + 4 <= size) {
    if (file[i] != 0xff) return -3;
    const unsigned m = file[i + 1];
    if (m == 0xff) { i++; continue; }  // fill byte
    const size_t ln = ((size_t)file[i + 2] << 8) | file[i + 3];
    if (ln < 2 || i + 2 + ln > size) return -4;
    const uint8_t* seg = file + i + 4;
    const size_t n = ln - 2;
    if (m == 0xdb) {  // DQT
      size_t j = 0;
      while (j < n) {
        const unsigned pq = seg[j] >> 4, tq = seg[j] & 15;
        if (pq != 0 || tq > 3 || j + 65 > n) return -5;  // baseline: 8-bit tables
        for (int k = 0; k < 64; k++) qt[tq][zz[k]] = seg[j + 1 + k];
        have_qt[tq] = true;
        j += 65;
      }
    } else if (m == 0xc4) {  // DHT
      size_t j = 0;
      while (j < n) {
        if (j + 17 > n) return -6;
        const unsigned tc = seg[j] >> 4, th = seg[j] & 15;
        if (tc > 1 || th > 3) return -6;
        int nsym = 0;
        hbits[tc][th][0] = 0;
        for (int l = 1; l <= 16; l++) { hbits[tc][th][l] = seg[j + l]; nsym += seg[j + l]; }
        if (nsym > 256 || j + 17 + (size_t)nsym > n) return -6;
        memset(hvals[tc][th], 0, 256);
        memcpy(hvals[tc][th], seg + j + 17, (size_t)nsym);
        have_ht[tc][th] = true;
        j += 17 + (size_t)nsym;
      }
    } else if (m == 0xc0) {  // SOF0
      if (n < 6 || seg[0] != 8) return -7;
      const unsigned h = ((unsigned)seg[1] << 8) | seg[2], w = ((unsigned)seg[3] << 8) | seg[4], nc = seg[5];
      if ((nc != 1 && nc != 3) || w == 0 || h == 0 || n < 6 + 3 * (size_t)nc) return -7;
      out->scan.num_components = (int)nc;
      out->scan.w = w;
      out->scan.h = h;
      for (unsigned c = 0; c < nc; c++) {
        comp_id[c] = seg[6 + 3 * c];
        out->scan.h_samp[c] = seg[7 + 3 * c] >> 4;
        out->scan.v_samp[c] = seg[7 + 3 * c] & 15;
        comp_tq[c] = seg[8 + 3 * c];
        if (out->scan.h_samp[c] < 1 || out->scan.h_samp[c] > 2 || out->scan.v_samp[c] < 1 || out->scan.v_samp[c] > 2 || comp_tq[c] > 3) return -7;
      }
      have_sof = true;
    } else if ((m >= 0xc1 && m <= 0xcf) && m != 0xc4 && m != 0xc8 && m != 0xcc) {
      return -8;  // extended / progressive / lossless / arithmetic: not this path
    } else if (m == 0xdd) {  // DRI
      if (n != 2) return -9;
      out->scan.restart_interval = ((int)seg[0] << 8) | seg[1];
    } else if (m == 0xda) {  // SOS
      if (!have_sof || n < 1) return -10;
      const int nc = out->scan.num_components;
      if (seg[0] != nc || n != 1 + 2 * (size_t)nc + 3) return -10;  // one scan with all components
      int td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
      for (int c = 0; c < nc; c++) {
        if (seg[1 + 2 * c] != comp_id[c]) return -10;
        td[c] = seg[2 + 2 * c] >> 4;
        ta[c] = seg[2 + 2 * c] & 15;
        if (td[c] > 3 || ta[c] > 3 || !have_ht[0][td[c]] || !have_ht[1][ta[c]] || !have_qt[comp_tq[c]]) return -11;
      }
      if (seg[1 + 2 * nc] != 0 || seg[2 + 2 * nc] != 63 || seg[3 + 2 * nc] != 0) return -10;  // Ss, Se, Ah/Al of a sequential scan
      if (nc == 3 && (td[1] != td[2] || ta[1] != ta[2])) return -12;  // the decoder takes one chroma pair
      // geometry: jpeg_component_info::width_in_blocks / height_in_blocks (jdmaster / jdinput initial_setup)
      int hmax = 1, vmax = 1;
      for (int c = 0; c < nc; c++) {
        if (out->scan.h_samp[c] > hmax) hmax = out->scan.h_samp[c];
        if (out->scan.v_samp[c] > vmax) vmax = out->scan.v_samp[c];
      }
      for (int c = 0; c < nc; c++) {
        const unsigned cw = (out->scan.w * (unsigned)out->scan.h_samp[c] + (unsigned)hmax - 1) / (unsigned)hmax;
        const unsigned ch = (out->scan.h * (unsigned)out->scan.v_samp[c] + (unsigned)vmax - 1) / (unsigned)vmax;
        out->scan.blocks_w[c] = (int)((cw + 7) / 8);
        out->scan.blocks_h[c] = (int)((ch + 7) / 8);
        memcpy(out->qtable[c], qt[comp_tq[c]], sizeof qt[0]);
      }
      if (nc == 1) { out->scan.h_samp[0] = 1; out->scan.v_samp[0] = 1; }  // a single-component scan is not interleaved
      // tables in the decoder's order: DC / AC of component 0, DC / AC of components 1 and 2
      const int pair[4][2] = {{0, td[0]}, {1, ta[0]}, {0, td[nc == 3 ? 1 : 0]}, {1, ta[nc == 3 ? 1 : 0]}};
      for (int t = 0; t < 4; t++) {
        memcpy(out->tables.bits[t], hbits[pair[t][0]][pair[t][1]], 17);
        memcpy(out->tables.vals[t], hvals[pair[t][0]][pair[t][1]], 256);
      }
      // entropy-coded data: up to the first marker that is neither stuffing nor RSTn
      const size_t start = i + 2 + ln;
      size_t e = start;
      while (e + 1 < size) {
        if (file[e] == 0xff && file[e + 1] != 0x00 && (file[e + 1] & 0xf8) != 0xd0 && file[e + 1] != 0xff) break;
        e++;
      }
      if (e + 1 >= size) return -13;  // no EOI (or any marker) after the data
      out->scan_offset = start;
      out->scan_bytes = e - start;
      return 0;
    }
    i += 2 + ln;
  }
  return -14;  // no SOS
}

}  // extern "C"
#pragma GCC visibility pop
