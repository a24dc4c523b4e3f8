// uhdr_hip_jpeg_seam.h -- the facade's seam at the JPEG block stage (see uhdr_hip_seam.h for the whole picture).
//
//   JpegEncoderHelper::encode   lib/src/jpegencoderhelper.cpp:131-244   libjpeg's level shift + JDCT_ISLOW FDCT + quantizer
//                               (and rgb_ycc_convert for a 3-channel map) run on the device (uhdr_hip_fdct_quant,
//                               uhdr_hip_jpeg_rgb_to_ycc); the coefficient blocks go back to libjpeg through
//                               jpeg_write_coefficients(), which keeps the marker writing and the Huffman pass.
//   JpegDecoderHelper::decode   lib/src/jpegdecoderhelper.cpp:205-413   jpeg_read_coefficients() (Huffman decode) stays in
//                               libjpeg; dequantization + JDCT_ISLOW IDCT (+ ycc_rgb_convert for a 3-channel map) run on the
//                               device (uhdr_hip_idct_dequant, uhdr_hip_jpeg_ycc_to_rgb).
//
// Both return true when the device did the work (*st = result; the caller only destroys the libjpeg object) and false
// when libjpeg's own path must run: acceleration not enabled, or a geometry whose edge-padding rules are libjpeg's to
// apply (component planes that are not whole 8x8 blocks, exotic sampling).
#ifndef UHDR_HIP_JPEG_SEAM_H
#define UHDR_HIP_JPEG_SEAM_H

#include <cstddef>
#include <cstdio>

extern "C" {
#include <jpeglib.h>
}

#include "ultrahdr_api.h"

namespace uhdr_hip_seam {

bool jpeg_compress_on_device(jpeg_compress_struct* cinfo, const unsigned char* planes[3], const unsigned int strides[3],
                             uhdr_img_fmt_t format, const void* icc, size_t icc_size, const char* comment,
                             uhdr_error_info_t* st);

bool jpeg_decompress_on_device(jpeg_decompress_struct* cinfo, bool want_rgb, unsigned char* dest,
                               const unsigned int hstride[3], const unsigned int vstride[3], uhdr_img_fmt_t* out_fmt,
                               uhdr_error_info_t* st);

}  // namespace uhdr_hip_seam

#endif  // UHDR_HIP_JPEG_SEAM_H
