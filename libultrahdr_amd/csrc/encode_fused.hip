// API-0 encode front end in ONE pass over the HDR image (MI355X extension, not a reference operator).
//
// JpegR::encodeJPEGR API-0 (/root/reference/lib/src/jpegr.cpp:202-251) runs three full-image loops
// back to back: toneMap (HDR -> 8-bit SDR RGBA8888), generateGainMap (reads both images again) and
// convert_raw_input_to_ycbcr (reads the SDR image a third time to make the base JPEG's planes).
// Unfused that is 4+4, 4+4+3 and 4+3 bytes per pixel of HBM traffic and every HDR pixel is unpacked
// and linearised twice.  Here one thread carries a pixel through all three stages in registers:
// 4 bytes in (RGBA1010102; 8 for RGBA-F16), 3 (base YCbCr 4:4:4) + 3 (map, or 12 of float gains in
// two-pass mode) out, + 4 if the caller also wants the RGBA8888 SDR rendition.
//
// The arithmetic is the same sequence of IEEE operations as the three kernels (encode_core.h), and
// the 8-bit quantisation between the stages is kept: the gain map is computed from the QUANTISED SDR
// bytes exactly as generateGainMap would read them back.  tests/test_gpu_parity.py checks the fused
// outputs against tone_map -> generate_gainmap -> convert_raw_input_to_ycbcr, bit for bit.
// Conditions: RGB HDR input (RGBA1010102 / RGBA-F16), gain map at full resolution (scale 1).
#include "encode_core.h"
#include "lds_copy.h"

namespace uhdr {
int fused_grid(uint32_t tiles, int per_cu);
namespace {

constexpr int kBlock = 512;  // 8 waves share one table set in LDS -> 3 workgroups = 24 waves per CU

template <int HDRF>
struct FusedLds {
  float srgb_of_byte[256];  // byte -> byte / 255.0f -> sRGB inverse-OETF table value (host_tables.cpp)
  // RGBA1010102 input: 10-bit code -> linear value (the host always provides it); RGBA-F16: the inverse-OETF table, if any
  float hdr[HDRF == UHDR_IMG_FMT_32bppRGBA1010102 ? 1024 : kInvOetfN];
  UnormTables unorm;
  uint2 srgb8[kStepTabMax];  // tone map: clamped linear value -> sRGB byte
  uint2 gain8[kStepTabMax];  // one pass: clamped gain -> map byte
};

__device__ __forceinline__ float clipf(float v, float hi) { return (v < 0.0f) ? 0.0f : ((v > hi) ? hi : v); }

template <int HDRF, bool TWO_PASS>
__global__ __launch_bounds__(kBlock) void encode_api0_fused_kernel(const FusedParams p, float* partials) {
  __shared__ FusedLds<HDRF> L;
  const uint32_t tid = threadIdx.x;
  copy_to_lds(L.srgb_of_byte, p.gen.srgb_of_byte, 256u, tid, kBlock);
  constexpr bool code_lin = HDRF == UHDR_IMG_FMT_32bppRGBA1010102;  // launch_encode_api0_fused checks that lin10 is there
  if constexpr (code_lin) {
    copy_to_lds(L.hdr, p.tm.lin10, 1024u, tid, kBlock);
  } else {
    if (p.tm.hdr_inv_lut) copy_to_lds(L.hdr, p.tm.hdr_inv_lut, (uint32_t)p.tm.hdr_inv_n, tid, kBlock);
  }
  stage_step_tab(L.srgb8, p.tm.srgb8, tid, kBlock);
  stage_step_tab(L.gain8, p.gen.gain8, tid, kBlock);
  fill_unorm_tables(L.unorm, tid, kBlock);
  __syncthreads();

  const uint32_t w = p.tm.hdr.w, h = p.tm.hdr.h;
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  const bool hdr_lut = p.tm.hdr_inv_lut != nullptr, hdr_lut_4096 = p.tm.hdr_inv_n == kInvOetfN;
  float mn[3] = {UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT}, mx[3] = {UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT};
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + tid;
    if (x >= w) continue;
    // ---- toneMap (jpegr.cpp:2147-2203) -------------------------------------------------------------------------
    Color3 l;
    if constexpr (code_lin) {  // unpack + inverse OETF (+ OOTF) of a 10-bit code is one table entry
      const uint32_t v = ((const uint32_t*)p.tm.hdr.p[0])[x + (size_t)y * p.tm.hdr.stride[0]];
      l = Color3{L.hdr[v & 0x3ffu], L.hdr[(v >> 10) & 0x3ffu], L.hdr[(v >> 20) & 0x3ffu]};
    } else {
      const Color3 g = fetch_pixel<HDRF>(p.tm.hdr, x, y, &L.unorm);
      l = linearise_hdr(g, L.hdr, hdr_lut, hdr_lut_4096);
    }
    uint32_t r8, g8, b8;  // putRgba8888Pixel's bytes
    tone_curve_bytes(l, p.tm, p.tm.math_tab + kPowDirOff, L.srgb8, r8, g8, b8);  // (the pow table stays in global memory: a call without the byte table is the exception)
    if (p.tm.sdr.p[0]) ((uint32_t*)p.tm.sdr.p[0])[x + (size_t)y * p.tm.sdr.stride[0]] = r8 | (g8 << 8) | (b8 << 16) | (255u << 24);
    // ---- generateGainMap on the quantised SDR pixel (jpegr.cpp:753-818 / 866-931), scale 1 ---------------------------
    const Color3 e = {L.unorm.u8[r8], L.unorm.u8[g8], L.unorm.u8[b8]};  // getRgba8888Pixel: byte / 255.0f
    Color3 sl = {L.srgb_of_byte[r8], L.srgb_of_byte[g8], L.srgb_of_byte[b8]};  // the sRGB inverse OETF of e, per byte value
    if (p.gen.sdr_gamut_on) {  // clipNegatives can only bite behind a matrix: table outputs are never negative
      sl = mat3_apply(sl, p.gen.sdr_gamut);
      sl.r = clip_neg(sl.r); sl.g = clip_neg(sl.g); sl.b = clip_neg(sl.b);
    }
    Color3 hl = l;  // the same inverse OETF (+ OOTF) the tone mapper just applied
    if (p.gen.hdr_gamut_on) {
      hl = mat3_apply(hl, p.gen.hdr_gamut);
      hl.r = clip_neg(hl.r); hl.g = clip_neg(hl.g); hl.b = clip_neg(hl.b);
    }
    gain_of_pixel<TWO_PASS>(sl, hl, p.gen, p.gen.math_tab, x, y, mn, mx, L.gain8);
    // ---- convert_raw_input_to_ycbcr(sdr, 4:4:4) (gainmapmath.cpp:1446-1472) ----------------------------------------------
    const Color3 q = rgb_to_yuv(e, p.base_k);
    ((uint8_t*)p.ycc.p[0])[(size_t)y * p.ycc.stride[0] + x] = (uint8_t)__builtin_amdgcn_fmed3f(q.r * 255.0f + 0.5f, 0.0f, 255.0f);
    ((uint8_t*)p.ycc.p[1])[(size_t)y * p.ycc.stride[1] + x] = (uint8_t)__builtin_amdgcn_fmed3f(q.g * 255.0f + 0.5f + 128.0f, 0.0f, 255.0f);
    ((uint8_t*)p.ycc.p[2])[(size_t)y * p.ycc.stride[2] + x] = (uint8_t)__builtin_amdgcn_fmed3f(q.b * 255.0f + 0.5f + 128.0f, 0.0f, 255.0f);
  }
  if constexpr (TWO_PASS) reduce_block_minmax<kBlock>(mn, mx, partials);
}


// ---- four pixels per lane (round 4) -----------------------------------------------------------------------------------------------
// The kernel above spends a third of its issue slots on per-pixel loop control and wave-uniform branches (61 scalar
// instructions per pixel next to 188 vector ones) and moves single bytes.  Here a lane owns FOUR consecutive pixels of a row:
// one 16-byte load of RGBA1010102, 4-byte stores per base plane, 12 bytes of map (or three 16-byte stores of gain
// ratios), 16 bytes of RGBA8888; gamut mode and channel count are template parameters, so the four pixels are one basic
// block.  Same operations per pixel in the same order -- tests/test_gpu_parity.py holds both kernels to the three operators.
// Requires: RGBA1010102 input, the sRGB byte table (and, one pass, the gain byte table), widths and strides that keep the
// vector accesses aligned (launch_encode_api0_fused checks; anything else runs the kernel above).
struct Fused4Lds {
  float srgb_of_byte[256];
  float hdr[1024];  // 10-bit code -> linear value (unpack + inverse OETF [+ OOTF])
  float u8[256];
  uint2 srgb8[kStepTabMax];
  uint2 gain8[kStepTabMax];
};
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
template <bool TWO_PASS, int GM, int MC, int TG>  // GM: 0 no gamut conversion on the gain side, 1 SDR side, 2 HDR side; TG: tone-map gamut conversion
__global__ __launch_bounds__(kBlock) void encode_api0_fused4_kernel(const FusedParams p, float* partials) {
  __shared__ Fused4Lds L;
  const uint32_t tid = threadIdx.x;
  copy_to_lds(L.srgb_of_byte, p.gen.srgb_of_byte, 256u, tid, kBlock);
  for (uint32_t i = tid; i < 256; i += kBlock) L.u8[i] = (float)i / 255.0f;
  copy_to_lds(L.hdr, p.tm.lin10, 1024u, tid, kBlock);
  stage_step_tab(L.srgb8, p.tm.srgb8, tid, kBlock);
  if constexpr (!TWO_PASS) stage_step_tab(L.gain8, p.gen.gain8, tid, kBlock);
  __syncthreads();
  const uint32_t w4 = p.tm.hdr.w / 4, h = p.tm.hdr.h;
  const uint32_t tiles_x = (w4 + 63) / 64, tiles = tiles_x * h;
  const float inv_tx = 1.0f / (float)tiles_x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (kBlock / 64) + (tid >> 6));
  const uint32_t nwaves = gridDim.x * (kBlock / 64);
  float mn[3] = {UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT}, mx[3] = {UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT};
  constexpr int NCH = MC ? 3 : 1;
  for (uint32_t t = wave; t < tiles; t += nwaves) {
    uint32_t y = (uint32_t)((float)t * inv_tx);  // t / tiles_x for t < 2^24: the float estimate is off by at most one
    if (y * tiles_x > t) y--;
    if ((y + 1) * tiles_x <= t) y++;
    const uint32_t x4 = (t - y * tiles_x) * 64 + lane;
    if (x4 >= w4) continue;
    const uint32_t x = x4 * 4;
    const u4v in = *(const u4v*)((const uint32_t*)p.tm.hdr.p[0] + (size_t)y * p.tm.hdr.stride[0] + x);
    const uint32_t px[4] = {in.x, in.y, in.z, in.w};
    uint32_t rgba[4], oy = 0, ocb = 0, ocr = 0;
    uint32_t mapb[NCH * 4];   // one pass: the map bytes of the four pixels
    float ratio[NCH * 4];     // two pass: their gain ratios
#pragma unroll
    for (int k = 0; k < 4; k++) {
      // ---- toneMap (jpegr.cpp:2147-2203): unpack + inverse OETF (+ OOTF) of a 10-bit code is one table entry -----------------
      const uint32_t v = px[k];
      const Color3 l = Color3{L.hdr[v & 0x3ffu], L.hdr[(v >> 10) & 0x3ffu], L.hdr[(v >> 20) & 0x3ffu]};
      const Color3 o = tone_curve_linear<TG>(l, p.tm);
      const uint32_t r8 = step_code(o.r, L.srgb8, p.tm.srgb8), g8 = step_code(o.g, L.srgb8, p.tm.srgb8), b8 = step_code(o.b, L.srgb8, p.tm.srgb8);
      rgba[k] = r8 | (g8 << 8) | (b8 << 16) | (255u << 24);
      // ---- generateGainMap on the quantised SDR pixel (jpegr.cpp:753-818 / 866-931), scale 1 ---------------------------------
      Color3 sl = {L.srgb_of_byte[r8], L.srgb_of_byte[g8], L.srgb_of_byte[b8]};
      if (GM == 1) {
        sl = mat3_apply(sl, p.gen.sdr_gamut);
        sl.r = clip_neg(sl.r); sl.g = clip_neg(sl.g); sl.b = clip_neg(sl.b);
      }
      Color3 hl = l;
      if (GM == 2) {
        hl = mat3_apply(hl, p.gen.hdr_gamut);
        hl.r = clip_neg(hl.r); hl.g = clip_neg(hl.g); hl.b = clip_neg(hl.b);
      }
      float sn[3], hn[3];
      if (MC) {
        sn[0] = sl.r * 203.0f; sn[1] = sl.g * 203.0f; sn[2] = sl.b * 203.0f;
        hn[0] = hl.r * p.gen.hdr_nits; hn[1] = hl.g * p.gen.hdr_nits; hn[2] = hl.b * p.gen.hdr_nits;
      } else if (p.gen.use_luminance) {  // SDR-gamut luminance coefficients for BOTH images (jpegr.cpp:803-805)
        sn[0] = (p.gen.lum[0] * sl.r + p.gen.lum[1] * sl.g + p.gen.lum[2] * sl.b) * 203.0f;
        hn[0] = (p.gen.lum[0] * hl.r + p.gen.lum[1] * hl.g + p.gen.lum[2] * hl.b) * p.gen.hdr_nits;
      } else {
        sn[0] = fmaxf(sl.r, fmaxf(sl.g, sl.b)) * 203.0f;
        hn[0] = fmaxf(hl.r, fmaxf(hl.g, hl.b)) * p.gen.hdr_nits;
      }
#pragma unroll
      for (int c = 0; c < NCH; c++) {
        if constexpr (!TWO_PASS) {  // encode_gain with its step table (encode_core.h)
          float gain = div_rn(hn[c], __builtin_fmaxf(sn[c], 0x1p-100f));
          gain = sn[c] > 0.0f ? gain : 1.0f;
          mapb[k * NCH + c] = step_code(gain, L.gain8, p.gen.gain8);
        } else {
          const float q = gain_ratio(sn[c], hn[c], p.gen.gain_cap);
          ratio[k * NCH + c] = q;
          mn[c] = __builtin_fminf(mn[c], q);
          mx[c] = __builtin_fmaxf(mx[c], q);
        }
      }
      // ---- convert_raw_input_to_ycbcr(sdr, 4:4:4) (gainmapmath.cpp:1446-1472) ------------------------------------------------
      const Color3 e = {L.u8[r8], L.u8[g8], L.u8[b8]};  // getRgba8888Pixel: byte / 255.0f
      const Color3 q = rgb_to_yuv(e, p.base_k);
      oy |= (uint32_t)__builtin_amdgcn_fmed3f(q.r * 255.0f + 0.5f, 0.0f, 255.0f) << (8 * k);
      ocb |= (uint32_t)__builtin_amdgcn_fmed3f(q.g * 255.0f + 0.5f + 128.0f, 0.0f, 255.0f) << (8 * k);
      ocr |= (uint32_t)__builtin_amdgcn_fmed3f(q.b * 255.0f + 0.5f + 128.0f, 0.0f, 255.0f) << (8 * k);
    }
    if (p.tm.sdr.p[0]) *(u4v*)((uint32_t*)p.tm.sdr.p[0] + (size_t)y * p.tm.sdr.stride[0] + x) = (u4v){rgba[0], rgba[1], rgba[2], rgba[3]};
    *(uint32_t*)((uint8_t*)p.ycc.p[0] + (size_t)y * p.ycc.stride[0] + x) = oy;
    *(uint32_t*)((uint8_t*)p.ycc.p[1] + (size_t)y * p.ycc.stride[1] + x) = ocb;
    *(uint32_t*)((uint8_t*)p.ycc.p[2] + (size_t)y * p.ycc.stride[2] + x) = ocr;
    if constexpr (!TWO_PASS) {
      uint8_t* o = p.gen.out + ((size_t)y * p.gen.out_stride + x) * NCH;
      if (MC) {
        uint32_t wds[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < 12; j++) wds[j / 4] |= mapb[j] << (8 * (j & 3));
        *(uint32_t*)o = wds[0]; *(uint32_t*)(o + 4) = wds[1]; *(uint32_t*)(o + 8) = wds[2];
      } else {
        *(uint32_t*)o = mapb[0] | (mapb[1] << 8) | (mapb[2] << 16) | (mapb[3] << 24);
      }
    } else {
      float* g = p.gen.gain_log2 + ((size_t)y * p.gen.map_w + x) * NCH;
#pragma unroll
      for (int j = 0; j < NCH; j++) *(float4*)(g + 4 * j) = float4{ratio[4 * j], ratio[4 * j + 1], ratio[4 * j + 2], ratio[4 * j + 3]};
    }
  }
  if constexpr (TWO_PASS) reduce_block_minmax<kBlock>(mn, mx, partials);
}

template <bool TWO_PASS, int GM, int MC>
void launch_fused4_t(const FusedParams& p, int grid, float* partials, hipStream_t s) {
  if (p.tm.gamut_on) hipLaunchKernelGGL((encode_api0_fused4_kernel<TWO_PASS, GM, MC, 1>), dim3(grid), dim3(kBlock), 0, s, p, partials);
  else hipLaunchKernelGGL((encode_api0_fused4_kernel<TWO_PASS, GM, MC, 0>), dim3(grid), dim3(kBlock), 0, s, p, partials);
}
template <bool TWO_PASS, int GM>
void launch_fused4_m(const FusedParams& p, int grid, float* partials, hipStream_t s) {
  if (p.gen.multichannel) launch_fused4_t<TWO_PASS, GM, 1>(p, grid, partials, s);
  else launch_fused4_t<TWO_PASS, GM, 0>(p, grid, partials, s);
}
template <bool TWO_PASS>
void launch_fused4(const FusedParams& p, int grid, float* partials, hipStream_t s) {
  if (p.gen.sdr_gamut_on) launch_fused4_m<TWO_PASS, 1>(p, grid, partials, s);
  else if (p.gen.hdr_gamut_on) launch_fused4_m<TWO_PASS, 2>(p, grid, partials, s);
  else launch_fused4_m<TWO_PASS, 0>(p, grid, partials, s);
}
bool fused4_ok(const FusedParams& p, bool two_pass) {
  if (p.tm.hdr.fmt != UHDR_IMG_FMT_32bppRGBA1010102 || !p.tm.lin10 || !p.tm.srgb8.tab || !p.gen.srgb_of_byte) return false;
  if (!two_pass && !p.gen.gain8.tab) return false;
  const uint32_t w = p.tm.hdr.w;
  auto al = [](const void* q, uintptr_t a) { return ((uintptr_t)q & (a - 1)) == 0; };
  if (w % 4 || p.tm.hdr.stride[0] % 4 || !al(p.tm.hdr.p[0], 16)) return false;
  for (int i = 0; i < 3; i++)
    if (p.ycc.stride[i] % 4 || !al(p.ycc.p[i], 4)) return false;
  if (p.tm.sdr.p[0] && (p.tm.sdr.stride[0] % 4 || !al(p.tm.sdr.p[0], 16))) return false;
  const uint32_t nch = p.gen.multichannel ? 3 : 1;
  if (two_pass) return al(p.gen.gain_log2, 16);
  return ((size_t)p.gen.out_stride * nch) % 4 == 0 && al(p.gen.out, 4);
}

}  // namespace

int fused_grid(uint32_t tiles, int per_cu) {
  static const int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  const int resident = cus * per_cu;  // 512-thread workgroups: 48 KB of LDS tables (RGBA1010102) -> 3 per CU, 60 KB (RGBA-F16) -> 2
  uint32_t g = tiles < (uint32_t)resident ? tiles : (uint32_t)resident;
  if (g > 2048) g = 2048;  // the host layer sizes the min/max partials buffer for 2048 workgroups
  return (int)(g < 1 ? 1 : g);
}

// two_pass: p.gen.gain_log2 / p.gen.minmax set as for launch_generate_gainmap; returns the grid size (= the number of
// ratio-extrema partials at p.gen.minmax + 6) through *grid_out for launch_minmax_table.
hipError_t launch_encode_api0_fused(const FusedParams& p, bool two_pass, int* grid_out, hipStream_t s) {
  float* partials4 = two_pass ? p.gen.minmax + 6 : nullptr;
  if (fused4_ok(p, two_pass)) {
    const uint32_t wave_tiles = ((p.tm.hdr.w / 4 + 63) / 64) * p.tm.hdr.h;
    const int grid = fused_grid((wave_tiles + kBlock / 64 - 1) / (kBlock / 64), 4);  // 39 KB of LDS tables: four 512-thread workgroups per CU
    if (grid_out) *grid_out = grid;
    if (two_pass) launch_fused4<true>(p, grid, partials4, s);
    else launch_fused4<false>(p, grid, partials4, s);
    return hipGetLastError();
  }
  const uint32_t tiles = ((p.tm.hdr.w + kBlock - 1) / kBlock) * p.tm.hdr.h;
  const bool f16 = p.tm.hdr.fmt == UHDR_IMG_FMT_64bppRGBAHalfFloat;
  if (!f16 && !p.tm.lin10) return hipErrorInvalidValue;  // the host layer always builds the code -> linear table
  const int grid = fused_grid(tiles, f16 ? 2 : 3);
  if (grid_out) *grid_out = grid;
  float* partials = two_pass ? p.gen.minmax + 6 : nullptr;
  if (f16) {
    if (two_pass) hipLaunchKernelGGL((encode_api0_fused_kernel<UHDR_IMG_FMT_64bppRGBAHalfFloat, true>), dim3(grid), dim3(kBlock), 0, s, p, partials);
    else hipLaunchKernelGGL((encode_api0_fused_kernel<UHDR_IMG_FMT_64bppRGBAHalfFloat, false>), dim3(grid), dim3(kBlock), 0, s, p, partials);
  } else {
    if (two_pass) hipLaunchKernelGGL((encode_api0_fused_kernel<UHDR_IMG_FMT_32bppRGBA1010102, true>), dim3(grid), dim3(kBlock), 0, s, p, partials);
    else hipLaunchKernelGGL((encode_api0_fused_kernel<UHDR_IMG_FMT_32bppRGBA1010102, false>), dim3(grid), dim3(kBlock), 0, s, p, partials);
  }
  return hipGetLastError();
}

}  // namespace uhdr
