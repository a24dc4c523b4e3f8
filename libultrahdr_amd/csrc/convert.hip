// Pixel-format / YUV-encoding conversions of the hot path on gfx950.
//   transformYuv420 / transformYuv444  (/root/reference/lib/src/gainmapmath.cpp:686-748), driven by
//   UltraHdr::convertYuv (lib/src/jpegr.cpp:436-518) -- in place, 8-bit planar.
//   convert_raw_input_to_ycbcr         (lib/src/gainmapmath.cpp:1291-1482) -- RGBA1010102 -> P010 /
//   10-bit 4:4:4, RGBA8888 / RGB888 -> 4:2:0 / 4:4:4 (full range).
// Integer outputs: bit-exact against the reference is required and tested.
#include "pixel_io.h"
#include "uhdr_types.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;

__device__ __forceinline__ uint8_t st8(float v) {  // static_cast<uint8_t>(CLIP3(v, 0, 255))
  // one three-operand median instead of two compares and two selects (the arguments are finite: sums of products of
  // 8- / 10-bit samples; a NaN would come out as 0 either way -- v_med3 falls back to min3, the cast of NaN is 0)
  v = __builtin_amdgcn_fmed3f(v, 0.0f, 255.0f);
  return (uint8_t)v;
}

// One thread per 2x2 quad.  In place is safe: a quad reads only its own 4 lumas + 1 chroma pair.
__global__ __launch_bounds__(kBlock) void transform_yuv420_kernel(const YuvXformParams p) {
  const uint32_t qw = p.img.w / 2, qh = p.img.h / 2;
  const uint32_t tiles_x = (qw + kBlock - 1) / kBlock, tiles = tiles_x * qh;
  uint8_t* yp = (uint8_t*)p.img.p[0];
  uint8_t* up = (uint8_t*)p.img.p[1];
  uint8_t* vp = (uint8_t*)p.img.p[2];
  const size_t sy = p.img.stride[0];
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t qy = t / tiles_x, qx = (t - qy * tiles_x) * kBlock + threadIdx.x;
    if (qx >= qw) continue;
    uint8_t* y0 = yp + (size_t)(qy * 2) * sy + qx * 2;
    uint8_t* y1 = y0 + sy;
    uint8_t* uq = up + (size_t)qy * p.img.stride[1] + qx;
    uint8_t* vq = vp + (size_t)qy * p.img.stride[2] + qx;
    const float u = (float)((int)*uq - 128) * (1 / 255.0f);
    const float v = (float)((int)*vq - 128) * (1 / 255.0f);
    const Color3 a = mat3_apply({(float)y0[0] * (1 / 255.0f), u, v}, p.c);
    const Color3 b = mat3_apply({(float)y0[1] * (1 / 255.0f), u, v}, p.c);
    const Color3 c = mat3_apply({(float)y1[0] * (1 / 255.0f), u, v}, p.c);
    const Color3 d = mat3_apply({(float)y1[1] * (1 / 255.0f), u, v}, p.c);
    const float nu = (((a.g + b.g) + c.g) + d.g) / 4.0f;
    const float nv = (((a.b + b.b) + c.b) + d.b) / 4.0f;
    y0[0] = st8(a.r * 255.0f + 0.5f);
    y0[1] = st8(b.r * 255.0f + 0.5f);
    y1[0] = st8(c.r * 255.0f + 0.5f);
    y1[1] = st8(d.r * 255.0f + 0.5f);
    *uq = st8(nu * 255.0f + 128.0f + 0.5f);
    *vq = st8(nv * 255.0f + 128.0f + 0.5f);
  }
}

__global__ __launch_bounds__(kBlock) void transform_yuv444_kernel(const YuvXformParams p) {
  const uint32_t w = p.img.w, h = p.img.h;
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    uint8_t* yq = (uint8_t*)p.img.p[0] + (size_t)y * p.img.stride[0] + x;
    uint8_t* uq = (uint8_t*)p.img.p[1] + (size_t)y * p.img.stride[1] + x;
    uint8_t* vq = (uint8_t*)p.img.p[2] + (size_t)y * p.img.stride[2] + x;
    const Color3 o = mat3_apply({(float)*yq * (1 / 255.0f), (float)((int)*uq - 128) * (1 / 255.0f),
                                 (float)((int)*vq - 128) * (1 / 255.0f)}, p.c);
    *yq = st8(o.r * 255.0f + 0.5f);
    *uq = st8(o.g * 255.0f + 128.0f + 0.5f);
    *vq = st8(o.b * 255.0f + 128.0f + 0.5f);
  }
}

__device__ __forceinline__ float clipf(float v, float hi) { return (v < 0.0f) ? 0.0f : ((v > hi) ? hi : v); }

// -----------------------------------------------------------------------------------------------------------------------
// Wide variants (round 3): the same arithmetic per pixel, but a lane owns EIGHT pixels of a row pair (four 2x2 quads) for the
// subsampled layouts / eight pixels of one row for 4:4:4, moves them with 8- and 16-byte loads and stores (a wave covers 512
// contiguous bytes of luma per instruction instead of 64), addresses rows as 32-bit offsets from uniform bases, and a
// resident grid strides over the tiles.  The launchers fall back to the one-quad-per-lane kernels above when the geometry
// or the alignment does not allow the vector accesses.
// -----------------------------------------------------------------------------------------------------------------------
typedef uint32_t u2v __attribute__((ext_vector_type(2)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t byte_of(uint32_t w, int k) { return (w >> (8 * k)) & 0xffu; }
__device__ __forceinline__ uint32_t cvt8(float v) { return (uint32_t)st8(v); }

typedef float f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2v splat2(float v) { return (f2v){v, v}; }
// The two pixels of a quad row travel as a 2-vector: every operation below is one IEEE multiply / add per element in
// the order of mat3_apply ((m0 * y + m1 * u) + m2 * v; the chroma products are the same for the quad's four pixels), so the
// bytes are those of the per-pixel kernel; written out because the compiler's own pairing came and went with unrelated
// changes (390 - 421 VALU instructions per 16 pixels, 294 in this form).
__global__ __launch_bounds__(kBlock) void transform_yuv420_wide_kernel(const YuvXformParams p) {
  const uint32_t tw = p.img.w / 8, qh = p.img.h / 2;  // tiles of 8 x 2 pixels
  const uint32_t total = tw * qh;
  uint8_t* const yp = (uint8_t*)p.img.p[0];
  uint8_t* const up = (uint8_t*)p.img.p[1];
  uint8_t* const vp = (uint8_t*)p.img.p[2];
  const uint32_t sy = p.img.stride[0], su = p.img.stride[1], sv = p.img.stride[2];
  const Mat3 c = p.c;
  const float k255 = 1 / 255.0f;
  const f2v m0 = splat2(c.m[0]), m3 = splat2(c.m[3]), m6 = splat2(c.m[6]), s255 = splat2(255.0f), half = splat2(0.5f), n255 = splat2(k255);
  const f2v m12 = {c.m[1], c.m[2]}, m45 = {c.m[4], c.m[5]}, m78 = {c.m[7], c.m[8]};
  for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
    const uint32_t qy = t / tw, tx = t - qy * tw;
    uint8_t* y0p = yp + ((size_t)(2 * qy) * sy + tx * 8);
    uint8_t* y1p = y0p + sy;
    uint8_t* uq = up + ((size_t)qy * su + tx * 4);
    uint8_t* vq = vp + ((size_t)qy * sv + tx * 4);
    const u2v r0 = *(const u2v*)y0p, r1 = *(const u2v*)y1p;
    const uint32_t uu = *(const uint32_t*)uq, vv = *(const uint32_t*)vq;
    uint32_t o0[2] = {0, 0}, o1[2] = {0, 0}, ou = 0, ov = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {  // quad q: pixels 2q, 2q + 1 of both rows, chroma sample q
      const f2v uv = (f2v){(float)((int)byte_of(uu, q) - 128), (float)((int)byte_of(vv, q) - 128)} * n255;  // {u, v}
      const uint32_t w0 = q < 2 ? r0.x : r0.y, w1 = q < 2 ? r1.x : r1.y;
      const int b = (q & 1) * 2;
      const f2v ya = (f2v){(float)byte_of(w0, b), (float)byte_of(w0, b + 1)} * n255;  // row 0: pixels a, bb
      const f2v yc = (f2v){(float)byte_of(w1, b), (float)byte_of(w1, b + 1)} * n255;  // row 1: pixels cc, d
      const f2v pr = m12 * uv, pg = m45 * uv, pb = m78 * uv;  // {m1 * u, m2 * v}, ...
      const f2v ru = splat2(pr.x), rv = splat2(pr.y), gu = splat2(pg.x), gv = splat2(pg.y), bu = splat2(pb.x), bv = splat2(pb.y);
      const f2v Ra = (m0 * ya + ru) + rv, Rc = (m0 * yc + ru) + rv;
      const f2v Ga = (m3 * ya + gu) + gv, Gc = (m3 * yc + gu) + gv;
      const f2v Ba = (m6 * ya + bu) + bv, Bc = (m6 * yc + bu) + bv;
      const float nu = (((Ga.x + Ga.y) + Gc.x) + Gc.y) / 4.0f;
      const float nv = (((Ba.x + Ba.y) + Bc.x) + Bc.y) / 4.0f;
      const f2v La = Ra * s255 + half, Lc = Rc * s255 + half;
      o0[q >> 1] |= (cvt8(La.x) | (cvt8(La.y) << 8)) << (16 * (q & 1));
      o1[q >> 1] |= (cvt8(Lc.x) | (cvt8(Lc.y) << 8)) << (16 * (q & 1));
      ou |= cvt8(nu * 255.0f + 128.0f + 0.5f) << (8 * q);
      ov |= cvt8(nv * 255.0f + 128.0f + 0.5f) << (8 * q);
    }
    *(u2v*)y0p = (u2v){o0[0], o0[1]};
    *(u2v*)y1p = (u2v){o1[0], o1[1]};
    *(uint32_t*)uq = ou;
    *(uint32_t*)vq = ov;
  }
}

__global__ __launch_bounds__(kBlock) void transform_yuv444_wide_kernel(const YuvXformParams p) {
  const uint32_t tw = p.img.w / 8, total = tw * p.img.h;
  const Mat3 c = p.c;
  const float k255 = 1 / 255.0f;
  for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
    const uint32_t y = t / tw, tx = t - y * tw;
    uint8_t* yq = (uint8_t*)p.img.p[0] + ((size_t)y * p.img.stride[0] + tx * 8);
    uint8_t* uq = (uint8_t*)p.img.p[1] + ((size_t)y * p.img.stride[1] + tx * 8);
    uint8_t* vq = (uint8_t*)p.img.p[2] + ((size_t)y * p.img.stride[2] + tx * 8);
    const u2v yy = *(const u2v*)yq, uu = *(const u2v*)uq, vv = *(const u2v*)vq;
    uint32_t oy[2] = {0, 0}, ou[2] = {0, 0}, ov[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint32_t wy = k < 4 ? yy.x : yy.y, wu = k < 4 ? uu.x : uu.y, wv = k < 4 ? vv.x : vv.y;
      const Color3 o = mat3_apply({(float)byte_of(wy, k & 3) * k255, (float)((int)byte_of(wu, k & 3) - 128) * k255,
                                   (float)((int)byte_of(wv, k & 3) - 128) * k255}, c);
      oy[k >> 2] |= cvt8(o.r * 255.0f + 0.5f) << (8 * (k & 3));
      ou[k >> 2] |= cvt8(o.g * 255.0f + 128.0f + 0.5f) << (8 * (k & 3));
      ov[k >> 2] |= cvt8(o.b * 255.0f + 128.0f + 0.5f) << (8 * (k & 3));
    }
    *(u2v*)yq = (u2v){oy[0], oy[1]};
    *(u2v*)uq = (u2v){ou[0], ou[1]};
    *(u2v*)vq = (u2v){ov[0], ov[1]};
  }
}

// packed RGBA8888 / RGBA1010102 -> Y'CbCr, eight pixels of a row (pair) per lane: 32-byte loads, out-of-place outputs as
// nontemporal vector stores (they are written once: the JPEG stage reads them in a later kernel)
template <bool TEN_BIT>
__device__ __forceinline__ Color3 unpack_rgb(uint32_t v, const UnormTables& ut) {
  if constexpr (TEN_BIT) return {ut.u10[v & 0x3ff], ut.u10[(v >> 10) & 0x3ff], ut.u10[(v >> 20) & 0x3ff]};
  else return {ut.u8[v & 0xff], ut.u8[(v >> 8) & 0xff], ut.u8[(v >> 16) & 0xff]};
}

template <bool TEN_BIT>
__global__ __launch_bounds__(kBlock) void rgb_to_ycbcr420_wide_kernel(const RgbToYcbcrParams p) {
  __shared__ UnormTables ut;
  fill_unorm_tables(ut, threadIdx.x, kBlock);
  __syncthreads();
  const uint32_t tw = p.src.w / 8, qh = p.src.h / 2, total = tw * qh;
  const float scale = TEN_BIT ? 1023.0f : 255.0f;
  const uint32_t* const sp = (const uint32_t*)p.src.p[0];
  const uint32_t ss = p.src.stride[0];
  const Rgb2Yuv k = p.k;
  for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
    const uint32_t qy = t / tw, tx = t - qy * tw;
    const uint32_t* s0 = sp + ((size_t)(2 * qy) * ss + tx * 8);
    const uint32_t* s1 = s0 + ss;
    uint32_t px[2][8];
    {
      const u4v a = *(const u4v*)s0, b = *(const u4v*)(s0 + 4), c = *(const u4v*)s1, d = *(const u4v*)(s1 + 4);
      px[0][0] = a.x; px[0][1] = a.y; px[0][2] = a.z; px[0][3] = a.w; px[0][4] = b.x; px[0][5] = b.y; px[0][6] = b.z; px[0][7] = b.w;
      px[1][0] = c.x; px[1][1] = c.y; px[1][2] = c.z; px[1][3] = c.w; px[1][4] = d.x; px[1][5] = d.y; px[1][6] = d.z; px[1][7] = d.w;
    }
    uint32_t yy[2][8], cu[4], cv[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      Color3 e[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {  // source order of the reference's loops: (row 0: x, x + 1), (row 1: x, x + 1)
        e[j] = rgb_to_yuv(unpack_rgb<TEN_BIT>(px[j >> 1][2 * q + (j & 1)], ut), k);
        yy[j >> 1][2 * q + (j & 1)] = (uint32_t)clipf(e[j].r * scale + 0.5f, scale);
      }
      float u = (e[0].g + e[1].g + e[2].g + e[3].g) / 4;
      float v = (e[0].b + e[1].b + e[2].b + e[3].b) / 4;
      if constexpr (TEN_BIT) {
        cu[q] = (uint32_t)clipf((u * 1023.0f) + 512.0f + 0.5f, 1023.0f);
        cv[q] = (uint32_t)clipf((v * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      } else {
        cu[q] = (uint32_t)clipf(u * 255.0f + 0.5f + 128.0f, 255.0f);
        cv[q] = (uint32_t)clipf(v * 255.0f + 0.5f + 128.0f, 255.0f);
      }
    }
    if constexpr (TEN_BIT) {  // P010: value << 6, interleaved UV
      uint16_t* yd = (uint16_t*)p.dst.p[0] + ((size_t)(2 * qy) * p.dst.stride[0] + tx * 8);
      uint16_t* cd = (uint16_t*)p.dst.p[1] + ((size_t)qy * p.dst.stride[1] + tx * 8);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const u4v o = {(yy[r][0] | (yy[r][1] << 16)) << 6, (yy[r][2] | (yy[r][3] << 16)) << 6, (yy[r][4] | (yy[r][5] << 16)) << 6,
                       (yy[r][6] | (yy[r][7] << 16)) << 6};
        __builtin_nontemporal_store(o, (u4v*)(yd + (size_t)r * p.dst.stride[0]));
      }
      const u4v oc = {(cu[0] | (cv[0] << 16)) << 6, (cu[1] | (cv[1] << 16)) << 6, (cu[2] | (cv[2] << 16)) << 6, (cu[3] | (cv[3] << 16)) << 6};
      __builtin_nontemporal_store(oc, (u4v*)cd);
    } else {
      uint8_t* yd = (uint8_t*)p.dst.p[0] + ((size_t)(2 * qy) * p.dst.stride[0] + tx * 8);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        const u2v o = {yy[r][0] | (yy[r][1] << 8) | (yy[r][2] << 16) | (yy[r][3] << 24), yy[r][4] | (yy[r][5] << 8) | (yy[r][6] << 16) | (yy[r][7] << 24)};
        __builtin_nontemporal_store(o, (u2v*)(yd + (size_t)r * p.dst.stride[0]));
      }
      __builtin_nontemporal_store(cu[0] | (cu[1] << 8) | (cu[2] << 16) | (cu[3] << 24), (uint32_t*)((uint8_t*)p.dst.p[1] + ((size_t)qy * p.dst.stride[1] + tx * 4)));
      __builtin_nontemporal_store(cv[0] | (cv[1] << 8) | (cv[2] << 16) | (cv[3] << 24), (uint32_t*)((uint8_t*)p.dst.p[2] + ((size_t)qy * p.dst.stride[2] + tx * 4)));
    }
  }
}

template <bool TEN_BIT>
__global__ __launch_bounds__(kBlock) void rgb_to_ycbcr444_wide_kernel(const RgbToYcbcrParams p) {
  __shared__ UnormTables ut;
  fill_unorm_tables(ut, threadIdx.x, kBlock);
  __syncthreads();
  const uint32_t tw = p.src.w / 8, total = tw * p.src.h;
  const uint32_t* const sp = (const uint32_t*)p.src.p[0];
  const uint32_t ss = p.src.stride[0];
  const Rgb2Yuv k = p.k;
  for (uint32_t t = blockIdx.x * kBlock + threadIdx.x; t < total; t += gridDim.x * kBlock) {
    const uint32_t y = t / tw, tx = t - y * tw;
    const uint32_t* s0 = sp + ((size_t)y * ss + tx * 8);
    const u4v a = *(const u4v*)s0, b = *(const u4v*)(s0 + 4);
    const uint32_t px[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t oy[8], ou[8], ov[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const Color3 q = rgb_to_yuv(unpack_rgb<TEN_BIT>(px[j], ut), k);
      if constexpr (TEN_BIT) {
        oy[j] = (uint32_t)clipf((q.r * 1023.0f) + 0.5f, 1023.0f);
        ou[j] = (uint32_t)clipf((q.g * 1023.0f) + 512.0f + 0.5f, 1023.0f);
        ov[j] = (uint32_t)clipf((q.b * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      } else {
        oy[j] = (uint32_t)clipf(q.r * 255.0f + 0.5f, 255.0f);
        ou[j] = (uint32_t)clipf(q.g * 255.0f + 0.5f + 128.0f, 255.0f);
        ov[j] = (uint32_t)clipf(q.b * 255.0f + 0.5f + 128.0f, 255.0f);
      }
    }
    const uint32_t* const planes[3] = {oy, ou, ov};
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const uint32_t* o = planes[c];
      if constexpr (TEN_BIT) {
        uint16_t* d = (uint16_t*)p.dst.p[c] + ((size_t)y * p.dst.stride[c] + tx * 8);
        __builtin_nontemporal_store((u4v){o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16)}, (u4v*)d);
      } else {
        uint8_t* d = (uint8_t*)p.dst.p[c] + ((size_t)y * p.dst.stride[c] + tx * 8);
        __builtin_nontemporal_store((u2v){o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24), o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24)}, (u2v*)d);
      }
    }
  }
}

// chroma-subsampled variants: one thread per 2x2 quad; 4:4:4 variants: one thread per pixel
template <bool TEN_BIT>
__global__ __launch_bounds__(kBlock) void rgb_to_ycbcr420_kernel(const RgbToYcbcrParams p) {
  __shared__ UnormTables ut;  // sample / 255.0f, sample / 1023.0f (the reference divides)
  fill_unorm_tables(ut, threadIdx.x, kBlock);
  __syncthreads();
  const uint32_t qw = p.src.w / 2, qh = p.src.h / 2;
  const uint32_t tiles_x = (qw + kBlock - 1) / kBlock, tiles = tiles_x * qh;
  const float scale = TEN_BIT ? 1023.0f : 255.0f;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t qy = t / tiles_x, qx = (t - qy * tiles_x) * kBlock + threadIdx.x;
    if (qx >= qw) continue;
    Color3 q[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      q[k] = rgb_to_yuv(fetch_pixel(p.src, qx * 2 + (k & 1), qy * 2 + (k >> 1), &ut), p.k);
      q[k].r = clipf(q[k].r * scale + 0.5f, scale);
    }
    float u = (q[0].g + q[1].g + q[2].g + q[3].g) / 4;
    float v = (q[0].b + q[1].b + q[2].b + q[3].b) / 4;
    if constexpr (TEN_BIT) {  // gainmapmath.cpp:1345-1364 (P010: value << 6, interleaved UV)
      uint16_t* yd = (uint16_t*)p.dst.p[0];
      uint16_t* cd = (uint16_t*)p.dst.p[1];
      const size_t sy = p.dst.stride[0];
      yd[(size_t)(qy * 2) * sy + qx * 2] = (uint16_t)((uint32_t)q[0].r << 6);
      yd[(size_t)(qy * 2) * sy + qx * 2 + 1] = (uint16_t)((uint32_t)q[1].r << 6);
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2] = (uint16_t)((uint32_t)q[2].r << 6);
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2 + 1] = (uint16_t)((uint32_t)q[3].r << 6);
      u = clipf((u * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      v = clipf((v * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      cd[(size_t)qy * p.dst.stride[1] + qx * 2] = (uint16_t)((uint32_t)u << 6);
      cd[(size_t)qy * p.dst.stride[1] + qx * 2 + 1] = (uint16_t)((uint32_t)v << 6);
    } else {  // gainmapmath.cpp:1426-1444
      uint8_t* yd = (uint8_t*)p.dst.p[0];
      const size_t sy = p.dst.stride[0];
      yd[(size_t)(qy * 2) * sy + qx * 2] = (uint8_t)q[0].r;
      yd[(size_t)(qy * 2) * sy + qx * 2 + 1] = (uint8_t)q[1].r;
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2] = (uint8_t)q[2].r;
      yd[(size_t)(qy * 2 + 1) * sy + qx * 2 + 1] = (uint8_t)q[3].r;
      ((uint8_t*)p.dst.p[1])[(size_t)qy * p.dst.stride[1] + qx] = (uint8_t)clipf(u * 255.0f + 0.5f + 128.0f, 255.0f);
      ((uint8_t*)p.dst.p[2])[(size_t)qy * p.dst.stride[2] + qx] = (uint8_t)clipf(v * 255.0f + 0.5f + 128.0f, 255.0f);
    }
  }
}

template <bool TEN_BIT>
__global__ __launch_bounds__(kBlock) void rgb_to_ycbcr444_kernel(const RgbToYcbcrParams p) {
  __shared__ UnormTables ut;
  fill_unorm_tables(ut, threadIdx.x, kBlock);
  __syncthreads();
  const uint32_t w = p.src.w, h = p.src.h;
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    const Color3 q = rgb_to_yuv(fetch_pixel(p.src, x, y, &ut), p.k);
    if constexpr (TEN_BIT) {  // gainmapmath.cpp:1386-1402
      ((uint16_t*)p.dst.p[0])[(size_t)y * p.dst.stride[0] + x] = (uint16_t)clipf((q.r * 1023.0f) + 0.5f, 1023.0f);
      ((uint16_t*)p.dst.p[1])[(size_t)y * p.dst.stride[1] + x] = (uint16_t)clipf((q.g * 1023.0f) + 512.0f + 0.5f, 1023.0f);
      ((uint16_t*)p.dst.p[2])[(size_t)y * p.dst.stride[2] + x] = (uint16_t)clipf((q.b * 1023.0f) + 512.0f + 0.5f, 1023.0f);
    } else {  // gainmapmath.cpp:1458-1472
      ((uint8_t*)p.dst.p[0])[(size_t)y * p.dst.stride[0] + x] = (uint8_t)clipf(q.r * 255.0f + 0.5f, 255.0f);
      ((uint8_t*)p.dst.p[1])[(size_t)y * p.dst.stride[1] + x] = (uint8_t)clipf(q.g * 255.0f + 0.5f + 128.0f, 255.0f);
      ((uint8_t*)p.dst.p[2])[(size_t)y * p.dst.stride[2] + x] = (uint8_t)clipf(q.b * 255.0f + 0.5f + 128.0f, 255.0f);
    }
  }
}

// copy_raw_image's two repacking cases (gainmapmath.cpp:1565-1597): RGB888 -> RGBA8888 (alpha 0xff) and
// RGBA8888 -> Y400 (the R byte).  MODE 0 / 1.  One lane per pixel, tiles of 256 pixels of one row.
template <int MODE>
__global__ __launch_bounds__(kBlock) void repack_kernel(const uint8_t* __restrict__ src, size_t src_pitch,
                                                        uint8_t* __restrict__ dst, size_t dst_pitch, uint32_t w, uint32_t h) {
  const uint32_t tiles_x = (w + kBlock - 1) / kBlock, tiles = tiles_x * h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (x >= w) continue;
    const uint8_t* s = src + (size_t)y * src_pitch;
    uint8_t* d = dst + (size_t)y * dst_pitch;
    if constexpr (MODE == 0) {
      ((uint32_t*)d)[x] = s[3 * x] | ((uint32_t)s[3 * x + 1] << 8) | ((uint32_t)s[3 * x + 2] << 16) | (0xffu << 24);
    } else {
      d[x] = (uint8_t)(((const uint32_t*)s)[x] & 0xff);
    }
  }
}

inline int grid_for(size_t total) {
  size_t g = (total + kBlock - 1) / kBlock;
  if (g > 4096) g = 4096;
  return (int)(g < 1 ? 1 : g);
}

}  // namespace

inline bool al(const void* ptr, size_t a) { return ((uintptr_t)ptr % a) == 0; }
// resident grid for the wide kernels: enough 256-thread workgroups to fill the chip a few times over, at most one tile per lane
inline int wide_grid(size_t tiles) {
  size_t g = (tiles + kBlock - 1) / kBlock;
  if (g > 256 * 8) g = 256 * 8;
  return (int)(g < 1 ? 1 : g);
}

hipError_t launch_transform_yuv(const YuvXformParams& p, hipStream_t s) {
  const ImageViewMut& im = p.img;
  if (im.fmt == UHDR_IMG_FMT_12bppYCbCr420 && im.w % 8 == 0 && im.h % 2 == 0 && im.stride[0] % 8 == 0 && im.stride[1] % 4 == 0 &&
      im.stride[2] % 4 == 0 && al(im.p[0], 8) && al(im.p[1], 4) && al(im.p[2], 4)) {
    hipLaunchKernelGGL(transform_yuv420_wide_kernel, dim3(wide_grid((size_t)(im.w / 8) * (im.h / 2))), dim3(kBlock), 0, s, p);
    return hipGetLastError();
  }
  if (im.fmt == UHDR_IMG_FMT_24bppYCbCr444 && im.w % 8 == 0 && im.stride[0] % 8 == 0 && im.stride[1] % 8 == 0 && im.stride[2] % 8 == 0 &&
      al(im.p[0], 8) && al(im.p[1], 8) && al(im.p[2], 8)) {
    hipLaunchKernelGGL(transform_yuv444_wide_kernel, dim3(wide_grid((size_t)(im.w / 8) * im.h)), dim3(kBlock), 0, s, p);
    return hipGetLastError();
  }
  if (p.img.fmt == UHDR_IMG_FMT_12bppYCbCr420) {
    hipLaunchKernelGGL(transform_yuv420_kernel, dim3(grid_for((size_t)(p.img.w / 2) * (p.img.h / 2))),
                       dim3(kBlock), 0, s, p);
  } else {
    hipLaunchKernelGGL(transform_yuv444_kernel, dim3(grid_for((size_t)p.img.w * p.img.h)), dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

hipError_t launch_repack(int mode, const void* src, size_t src_pitch, void* dst, size_t dst_pitch, uint32_t w, uint32_t h,
                         hipStream_t s) {
  const int g = grid_for((size_t)w * h);
  if (mode == 0) hipLaunchKernelGGL((repack_kernel<0>), dim3(g), dim3(kBlock), 0, s, (const uint8_t*)src, src_pitch, (uint8_t*)dst, dst_pitch, w, h);
  else hipLaunchKernelGGL((repack_kernel<1>), dim3(g), dim3(kBlock), 0, s, (const uint8_t*)src, src_pitch, (uint8_t*)dst, dst_pitch, w, h);
  return hipGetLastError();
}

hipError_t launch_rgb_to_ycbcr(const RgbToYcbcrParams& p, hipStream_t s) {
  const bool ten = p.src.fmt == UHDR_IMG_FMT_32bppRGBA1010102;
  const bool sub = p.dst.fmt == UHDR_IMG_FMT_24bppYCbCrP010 || p.dst.fmt == UHDR_IMG_FMT_12bppYCbCr420;
  // wide path: packed 32-bit source pixels, widths in whole 8-pixel tiles, every vector access aligned
  const bool packed32 = p.src.fmt == UHDR_IMG_FMT_32bppRGBA1010102 || p.src.fmt == UHDR_IMG_FMT_32bppRGBA8888;
  bool wide = packed32 && p.src.w % 8 == 0 && p.src.stride[0] % 4 == 0 && al(p.src.p[0], 16) && (!sub || p.src.h % 2 == 0);
  if (wide) {
    const size_t ya = ten ? 16 : 8;  // bytes of 8 luma samples
    if (sub) {
      wide = al(p.dst.p[0], ya) && (p.dst.stride[0] * (ten ? 2 : 1)) % ya == 0;
      if (ten) wide = wide && al(p.dst.p[1], 16) && (p.dst.stride[1] * 2) % 16 == 0;
      else wide = wide && al(p.dst.p[1], 4) && al(p.dst.p[2], 4) && p.dst.stride[1] % 4 == 0 && p.dst.stride[2] % 4 == 0;
    } else {
      for (int c = 0; c < 3; c++) wide = wide && al(p.dst.p[c], ya) && (p.dst.stride[c] * (ten ? 2 : 1)) % ya == 0;
    }
  }
  if (wide) {
    const int g = wide_grid(sub ? (size_t)(p.src.w / 8) * (p.src.h / 2) : (size_t)(p.src.w / 8) * p.src.h);
    if (sub) {
      if (ten) hipLaunchKernelGGL((rgb_to_ycbcr420_wide_kernel<true>), dim3(g), dim3(kBlock), 0, s, p);
      else hipLaunchKernelGGL((rgb_to_ycbcr420_wide_kernel<false>), dim3(g), dim3(kBlock), 0, s, p);
    } else {
      if (ten) hipLaunchKernelGGL((rgb_to_ycbcr444_wide_kernel<true>), dim3(g), dim3(kBlock), 0, s, p);
      else hipLaunchKernelGGL((rgb_to_ycbcr444_wide_kernel<false>), dim3(g), dim3(kBlock), 0, s, p);
    }
    return hipGetLastError();
  }
  if (sub) {
    const int g = grid_for((size_t)(p.src.w / 2) * (p.src.h / 2));
    if (ten) hipLaunchKernelGGL((rgb_to_ycbcr420_kernel<true>), dim3(g), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((rgb_to_ycbcr420_kernel<false>), dim3(g), dim3(kBlock), 0, s, p);
  } else {
    const int g = grid_for((size_t)p.src.w * p.src.h);
    if (ten) hipLaunchKernelGGL((rgb_to_ycbcr444_kernel<true>), dim3(g), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((rgb_to_ycbcr444_kernel<false>), dim3(g), dim3(kBlock), 0, s, p);
  }
  return hipGetLastError();
}

}  // namespace uhdr
