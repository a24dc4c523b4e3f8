// generateGainMap on gfx950: SDR + HDR renditions -> 8-bit log-ratio gain map.
// Reference loops: /root/reference/lib/src/jpegr.cpp:753-818 (one pass), 866-931 + 992-1013 (two
// pass) with encodeGain / computeGain / affineMapGain from lib/src/gainmapmath.cpp:753-789.
//
// One thread per map pixel (each reads its s x s box of both images once: the minimum traffic); a
// workgroup walks tiles of consecutive map pixels of one row.  The kernel is instantiated per
// (SDR format, HDR format) so the pixel unpack has no per-pixel format dispatch.  Look-up tables in LDS:
//   sRGB inverse OETF (1024 floats), HDR inverse OETF (4096 floats; for HLG the host has already
//   folded hlgOotfApprox's powf(x, 1.2f) into it -- an exact fusion, the composite is evaluated with
//   the host libm at the 4096 table nodes), sample normalisation, and for one-pass maps the gain -> byte step table.
//
// Two-pass mode (round 4): the reference stores (float)log2(ratio) per sample, merges min / max and maps
// (g - min) / (max - min) to a byte.  log2 is monotone, so min / max commute with it: pass 1 here stores the RATIO
// (hdr + eps) / (sdr + eps) -- an exact float division, encode_core.h -- and reduces ratio extrema (wavefront shuffles ->
// LDS -> one partial per workgroup); ONE small kernel (minmax_table_kernel, a workgroup per channel) then takes the
// float64 log2 of the six extrema -- the reference's six floats, bit for bit --, applies jpegr.cpp:969-986 and tabulates
// pass 2's ratio -> byte step function by bisection through the exact evaluation; pass 2 is a table lookup per sample.
// No per-pixel logarithm is left, and nothing was approximated: every table entry is the exact composite's value.
// Across GPUs the six log2 extrema are all-reduced by the host layer (one RCCL min over {min, -max}) between the
// reduce and the finalize halves of that kernel.
#include "encode_core.h"
#include "lds_copy.h"

namespace uhdr {
namespace {

constexpr int kBlock = 256;
constexpr int kGenBlock = 512;  // generate_kernel: 8 waves share one 25 KB table set -> 3 workgroups = 24 waves per CU (VGPR limit)
constexpr int kMaxGrid = 2048;  // the host layer sizes the partials buffer for this many workgroups

struct GenLds {
  float srgb[kSrgbN];
  float hdr[kInvOetfN];
  UnormTables unorm;  // x / 255.0f, x / 1023.0f
};

// General form: one thread per map pixel, any format pair / scale factor.
template <int SDRF, int HDRF, bool TWO_PASS>
__global__ __launch_bounds__(kGenBlock) void generate_kernel(const GenParams p, float* partials) {
  __shared__ GenLds L;
  __shared__ uint2 s_gain8[TWO_PASS ? 1 : kStepTabMax];  // one pass: clamped gain -> map byte (host_tables.cpp)
  const uint32_t tid = threadIdx.x;
  if constexpr (!TWO_PASS) stage_step_tab(s_gain8, p.gain8, tid, kGenBlock);
  for (uint32_t i = tid; i < kSrgbN; i += kGenBlock) L.srgb[i] = p.srgb_lut[i];
  if (p.hdr_inv_lut)
    for (uint32_t i = tid; i < (uint32_t)p.hdr_inv_n; i += kGenBlock) L.hdr[i] = p.hdr_inv_lut[i];
  fill_unorm_tables(L.unorm, tid, kGenBlock);
  __syncthreads();

  const bool hdr_lut = p.hdr_inv_lut != nullptr, hdr_lut_4096 = p.hdr_inv_n == kInvOetfN;
  float mn[3] = {UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT}, mx[3] = {UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT};
  const uint32_t mw = p.map_w, mh = p.map_h;
  const uint32_t tiles_x = (mw + kGenBlock - 1) / kGenBlock, tiles = tiles_x * mh;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, x = (t - y * tiles_x) * kGenBlock + tid;
    if (x >= mw) continue;
    // one map pixel from the two samples as sample_box delivers them.  clipNegatives (jpegr.cpp:783-784) can only bite behind
    // a gamut conversion: table outputs and sanitised half floats are never negative.
    Color3 s = sample_box<SDRF>(p.sdr, p.scale, x, y, &L.unorm), h = sample_box<HDRF>(p.hdr, p.scale, x, y, &L.unorm);
    if (!p.sdr_is_rgb) s = yuv_to_rgb(s.r, s.g, s.b, p.sdr_yuv);
    Color3 sl = linearise_hdr(s, L.srgb, true, false);  // the 1024-entry sRGB table; box means of RGB samples stay in [0, 1]
    if (p.sdr_gamut_on) {
      sl = mat3_apply(sl, p.sdr_gamut);
      sl.r = clip_neg(sl.r); sl.g = clip_neg(sl.g); sl.b = clip_neg(sl.b);
    }
    if (!p.hdr_is_rgb) h = yuv_to_rgb(h.r, h.g, h.b, p.hdr_yuv);
    Color3 hl = linearise_hdr(h, L.hdr, hdr_lut, hdr_lut_4096);
    if (p.hdr_gamut_on) {
      hl = mat3_apply(hl, p.hdr_gamut);
      hl.r = clip_neg(hl.r); hl.g = clip_neg(hl.g); hl.b = clip_neg(hl.b);
    }
    gain_of_pixel<TWO_PASS>(sl, hl, p, p.math_tab, x, y, mn, mx, TWO_PASS ? nullptr : s_gain8);
  }
  if constexpr (TWO_PASS) reduce_block_minmax<kGenBlock>(mn, mx, partials);
}

// QUAD form: 4:2:0 SDR + P010 HDR at scale 1 with even geometry (the API-1 default) -- one thread per 2x2 quad, both images
// read with the coalesced quad fetches of pixel_io.h (luma as one vector load per row, chroma once per quad); a WAVE walks
// tiles of 64 consecutive quads of one quad row.  Gamut mode (0 none, 1 SDR side, 2 HDR side), channel count and the
// presence of an HDR linearisation table are template parameters, the table size a float factor: the four pixels of a
// quad are one basic block and their 24 table reads go out together.
template <bool TWO_PASS, int GM, int MC, bool LUT>
__global__ __launch_bounds__(kGenBlock) void generate_quad_kernel(const GenParams p, float* partials) {
  __shared__ GenLds L;
  __shared__ uint2 s_gain8[TWO_PASS ? 1 : kStepTabMax];
  const uint32_t tid = threadIdx.x;
  if constexpr (!TWO_PASS) stage_step_tab(s_gain8, p.gain8, tid, kGenBlock);
  for (uint32_t i = tid; i < kSrgbN; i += kGenBlock) L.srgb[i] = p.srgb_lut[i];
  if (LUT)
    for (uint32_t i = tid; i < (uint32_t)p.hdr_inv_n; i += kGenBlock) L.hdr[i] = p.hdr_inv_lut[i];
  fill_unorm_tables(L.unorm, tid, kGenBlock);
  __syncthreads();
  float mn[3] = {UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT, UHDR_RATIO_MIN_INIT}, mx[3] = {UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT, UHDR_RATIO_MAX_INIT};
  const uint32_t qw = p.map_w / 2, qh = p.map_h / 2;
  const uint32_t tiles_x = (qw + 63) / 64, tiles = tiles_x * qh;
  const float inv_tx = 1.0f / (float)tiles_x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (kGenBlock / 64) + (tid >> 6));
  const uint32_t nwaves = gridDim.x * (kGenBlock / 64);
  const float lut_scale = (float)(p.hdr_inv_n - 1);
  for (uint32_t t = wave; t < tiles; t += nwaves) {
    uint32_t qy = (uint32_t)((float)t * inv_tx);  // t / tiles_x for t < 2^24: the float estimate is off by at most one
    if (qy * tiles_x > t) qy--;
    if ((qy + 1) * tiles_x <= t) qy++;
    const uint32_t qx = (t - qy * tiles_x) * 64 + lane;
    if (qx >= qw) continue;
    const QuadYuv sq = fetch_quad_420(p.sdr, qx, qy);
    const QuadYuv hq = fetch_quad_p010(p.hdr, qx, qy, &L.unorm);
    Color3 sl[4], hl[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const Color3 s = yuv_to_rgb(sq.px[k].r, sq.px[k].g, sq.px[k].b, p.sdr_yuv);  // in [0, 1]
      sl[k] = lut3_unit<kSrgbN>(s, L.srgb);
      const Color3 h = yuv_to_rgb(hq.px[k].r, hq.px[k].g, hq.px[k].b, p.hdr_yuv);
      hl[k] = h;
      if (LUT) hl[k] = Color3{L.hdr[rpi(h.r * lut_scale)], L.hdr[rpi(h.g * lut_scale)], L.hdr[rpi(h.b * lut_scale)]};
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (GM == 1) {  // clipNegatives (jpegr.cpp:783-784) can only bite behind a matrix: table outputs are never negative
        sl[k] = mat3_apply(sl[k], p.sdr_gamut);
        sl[k].r = clip_neg(sl[k].r); sl[k].g = clip_neg(sl[k].g); sl[k].b = clip_neg(sl[k].b);
      }
      if (GM == 2) {
        hl[k] = mat3_apply(hl[k], p.hdr_gamut);
        hl[k].r = clip_neg(hl[k].r); hl[k].g = clip_neg(hl[k].g); hl[k].b = clip_neg(hl[k].b);
      }
      gain_of_pixel<TWO_PASS, MC>(sl[k], hl[k], p, p.math_tab, 2 * qx + (k & 1), 2 * qy + (k >> 1), mn, mx, TWO_PASS ? nullptr : s_gain8);
    }
  }
  if constexpr (TWO_PASS) reduce_block_minmax<kGenBlock>(mn, mx, partials);
}

// ---- between the passes ------------------------------------------------------------------------------------------------------
// pass 2's composite for one channel, exactly as the reference evaluates it per sample (jpegr.cpp:900-928 stored the float
// log2 gain, affineMapGain gainmapmath.cpp:784-789 maps it): ratio -> byte
template <bool GAMMA = true>
__device__ __forceinline__ uint32_t affine_code(float q, float mn, double rr, float gamma, const double* T) {
  const float g = gain_log2_of_ratio(q, T);
  float m = div_by_rcp64(g - mn, rr);  // (g - min) / (max - min), exact (device_math.h)
  if (GAMMA && gamma != 1.0f) m = (float)pow((double)m, (double)gamma);
  m *= 255.0f;
  float t2 = m + 0.5f;
  t2 = (t2 < 0.0f) ? 0.0f : ((t2 > 255.0f) ? 255.0f : t2);
  return (uint32_t)t2;
}

constexpr int kTabBlock = 1024;
__global__ __launch_bounds__(kTabBlock) void minmax_table_kernel(const MinmaxTableParams p) {
  __shared__ float s_red[kTabBlock / 64][2];
  __shared__ float s_mm[2];
  __shared__ uint32_t s_geo[8];
  __shared__ double s_T[kMathTabDoubles];  // the log2 tables: every bisection step is a dependent table read
  const int c = blockIdx.x;  // channel
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // this thread's partials first, THEN the table staging: both round trips to memory are in flight together (the kernel is a
  // chain of latencies -- 6 of its 9 us were loads waiting for one another)
  float a = UHDR_RATIO_MIN_INIT, b = UHDR_RATIO_MAX_INIT;
  if (p.do_reduce && !p.empty)
    for (int i = tid; i < p.n_partials; i += kTabBlock) {
      a = fminf(a, p.partials[(size_t)i * 6 + c]);
      b = fmaxf(b, p.partials[(size_t)i * 6 + 3 + c]);
    }
  float merged_min = 0.0f, merged_max = 0.0f;
  if (p.do_finalize && p.merged_in) { merged_min = p.merged_in[c]; merged_max = -p.merged_in[3 + c]; }
  for (int i = tid; i < kMathTabDoubles; i += kTabBlock) s_T[i] = p.math_tab[i];
  __syncthreads();
  const double* T = s_T;
  float gmin, gmax;
  if (p.do_reduce) {  // ratio extrema of this channel over the partials of pass 1 -> the reference's log2 extrema
    a = wave_min(a);
    b = wave_max(b);
    if (lane == 0) { s_red[wv][0] = a; s_red[wv][1] = b; }
    __syncthreads();
    if (tid == 0) {
      for (int k = 1; k < kTabBlock / 64; k++) { a = fminf(a, s_red[k][0]); b = fmaxf(b, s_red[k][1]); }
      const float lmin = log2_extremum(a, true, T), lmax = log2_extremum(b, false, T);
      p.mm6[c] = lmin;
      p.mm6[3 + c] = lmax;
      if (p.merged6) { p.merged6[c] = lmin; p.merged6[3 + c] = -lmax; }
      s_mm[0] = lmin; s_mm[1] = lmax;
    }
    __syncthreads();
    gmin = s_mm[0]; gmax = s_mm[1];
  }
  if (!p.do_finalize && !p.do_table) return;
  if (p.do_finalize) {  // jpegr.cpp:969-986: clamp to [-14.3, 15.6], the user's min / max content-boost hints, the epsilon guard
    if (p.merged_in) { gmin = merged_min; gmax = merged_max; }
    if (c < p.nch) {
      gmin = gmin < -14.3f ? -14.3f : (gmin > 15.6f ? 15.6f : gmin);
      gmax = gmax < -14.3f ? -14.3f : (gmax > 15.6f ? 15.6f : gmax);
      if (p.has_max_hint) gmax = gmax < p.log2_max_hint ? gmax : p.log2_max_hint;
      if (p.has_min_hint) gmin = gmin < p.log2_min_hint ? p.log2_min_hint : gmin;
      if (fabsf(gmax - gmin) < 1.1920928955078125e-07f) gmax += 0.1f;  // FLT_EPSILON
    }
    if (tid == 0 && p.out_mm) { p.out_mm[c] = gmin; p.out_mm[3 + c] = gmax; }
  } else {
    gmin = p.final_mm[c]; gmax = p.final_mm[3 + c];
  }
  // the affine map's per-channel constants exactly as the host computed them (float subtraction, float64 reciprocal)
  const double rr = 1.0 / (double)(gmax - gmin);
  AffineTabDev& td = p.dev->tab[c];
  if (tid == 0) { p.dev->mn[c] = gmin; p.dev->mx[c] = gmax; p.dev->range_rcp[c] = rr; }
  if (!p.do_table || c >= p.nch || p.gamma != 1.0f) {  // (a user gamma: pass 2 evaluates per sample, launch_affine_map)
    if (tid == 0) {
      td.ok = 0;
      if (p.out_mm) p.out_mm[6 + c] = 0.0f;
    }
    return;
  }
  // ---- the ratio -> byte step table of this channel --------------------------------------------------------------------------
  // geometry: buckets per binade B = 2^(23 - shift) such that a bucket is narrower (in log2) than the spacing of the byte
  // thresholds, (max - min) / 255 for gamma 1: 1.4427 / B < range / 255 with 5 % to spare
  if (tid == 0) {
    const float range = gmax - gmin;
    uint32_t ok = (range > 0.0f) && (range < 64.0f);
    uint32_t shift = 0, lo_bits = 0, hi_bits = 0, n = 0;
    if (ok) {
      const float need = 386.3f / range;  // 1.4427 * 255 * 1.05
      int e = 0;
      while ((float)(1u << e) < need && e < 21) e++;
      shift = 23u - (uint32_t)e;
      if (e > 20) ok = 0;  // step_code addresses buckets with shift >= 3
      // domain: two buckets beyond [2^min, 2^max] on either side (exp2f is accurate to a few ulp, a bucket is at least 8;
      // the saturation checks below are what counts)
      const float lo_q = exp2f(gmin), hi_q = exp2f(gmax);
      ok = ok && (lo_q > 0x1p-100f) && (hi_q < 0x1p100f);
      lo_bits = ((__float_as_uint(lo_q) >> shift) - 2u) << shift;             // start of the second bucket below
      hi_bits = (((__float_as_uint(hi_q) >> shift) + 3u) << shift) - 1u;      // end of the second bucket above
      n = (hi_bits >> shift) - (lo_bits >> shift) + 1u;
      if (n > (uint32_t)kAffTabMax) ok = 0;
      if (ok) {  // the step function must be saturated at both ends of the domain and beyond
        const uint32_t c_lo = affine_code<false>(__uint_as_float(lo_bits), gmin, rr, p.gamma, T), c_hi = affine_code<false>(__uint_as_float(hi_bits), gmin, rr, p.gamma, T);
        if (c_lo != affine_code<false>(0x1p-120f, gmin, rr, p.gamma, T) || c_hi != affine_code<false>(0x1p120f, gmin, rr, p.gamma, T) || c_hi < c_lo) ok = 0;
      }
    }
    s_geo[0] = ok; s_geo[1] = shift; s_geo[2] = lo_bits; s_geo[3] = hi_bits; s_geo[4] = n;
    s_geo[5] = 1;  // every bucket holds at most one threshold
  }
  __syncthreads();
  const uint32_t shift = s_geo[1], lo_bits = s_geo[2], n = s_geo[4];
  uint2* tab = (uint2*)((char*)p.dev + kAffineTablesOff) + (size_t)c * kAffTabMax;
  if (s_geo[0]) {
    for (uint32_t k = tid; k < n; k += kTabBlock) {
      const uint32_t start = lo_bits + (k << shift), end = start + (1u << shift) - 1u;
      const uint32_t f_lo = affine_code<false>(__uint_as_float(start), gmin, rr, p.gamma, T), f_hi = affine_code<false>(__uint_as_float(end), gmin, rr, p.gamma, T);
      uint32_t thr = 0xFFFFFFFFu;
      if (f_hi != f_lo) {
        uint32_t a = start, b = end;  // invariant: code(a) == f_lo < code(b)
        // A bracket first: the byte changes from f_hi - 1 to f_hi where (g - min) / range * 255 + 0.5 crosses f_hi, i.e. at
        // g = min + (f_hi - 0.5) * range / 255; exp2f of that is within a few ulp of the threshold ratio.  If the exact
        // evaluation confirms the bracket the bisection needs 7 steps instead of `shift` (17 at a typical range: each step is
        // a dependent float64 chain, and the kernel was little else).  If not (a step of two codes, a range so narrow
        // that the log2's float rounding decides), the whole bucket is searched as before.
        {
          const float g_thr = gmin + ((float)f_hi - 0.5f) * ((gmax - gmin) * (1.0f / 255.0f));
          const uint32_t est = __float_as_uint(exp2f(g_thr));
          const uint32_t a2 = max(start, est - 64u), b2 = min(end, est + 64u);
          if (est > 64u && a2 < b2 && a2 >= start && b2 <= end && affine_code<false>(__uint_as_float(a2), gmin, rr, p.gamma, T) == f_lo &&
              affine_code<false>(__uint_as_float(b2), gmin, rr, p.gamma, T) > f_lo) {
            a = a2;
            b = b2;
          }
        }
        while (b - a > 1u) {
          const uint32_t mid = a + (b - a) / 2u;
          if (affine_code<false>(__uint_as_float(mid), gmin, rr, p.gamma, T) > f_lo) b = mid; else a = mid;
        }
        thr = b;
        if (affine_code<false>(__uint_as_float(b), gmin, rr, p.gamma, T) != f_hi || f_hi < f_lo) atomicAnd(&s_geo[5], 0u);  // a second threshold / not monotone
      }
      tab[k] = uint2{thr, f_lo | (f_hi << 16)};
    }
  }
  __syncthreads();
  if (tid == 0) {
    td.n = n;
    td.base8 = (lo_bits >> shift) * 8u;
    td.shm3 = shift - 3u;
    td.lo_bits = lo_bits;
    td.hi_bits = s_geo[3];
    td.ok = s_geo[0] & s_geo[5];
    if (p.out_mm) p.out_mm[6 + c] = td.ok ? 1.0f : 0.0f;  // for uhdr_hip_get_stats
  }
}

// ---- pass 2: affineMapGain (gainmapmath.cpp:784-789) over the float plane (jpegr.cpp:992-1013) ------------------------------
struct AffineLds {
  uint2 tab[3][kAffTabMax];
};
template <int NCH>
__device__ __forceinline__ bool stage_affine_tabs(const AffineParams& p, AffineLds& L, StepTab st[3], uint32_t tid, uint32_t nthreads) {
  bool ok = true;
#pragma unroll
  for (int c = 0; c < NCH; c++) {
    const AffineTabDev& td = p.dev->tab[c];  // wave-uniform: scalar loads
    st[c].tab = nullptr;
    st[c].n = td.n; st[c].base8 = td.base8; st[c].shm3 = td.shm3; st[c].lo_bits = td.lo_bits; st[c].hi_bits = td.hi_bits;
    ok = ok && td.ok != 0;
  }
  if (ok) {
    const uint2* src = (const uint2*)((const char*)p.dev + kAffineTablesOff);
#pragma unroll
    for (int c = 0; c < NCH; c++)
      copy_to_lds(L.tab[c], src + (size_t)c * kAffTabMax, st[c].n * 2u, tid, nthreads);
  }
  __syncthreads();
  return ok;
}

// one thread per four consecutive samples of a row (16-byte load, 4-byte store) when the geometry allows, else per sample
template <bool VEC4>
__global__ __launch_bounds__(kBlock) void affine_kernel(const AffineParams p) {
  __shared__ AffineLds L;
  StepTab st[3];
  const bool tabs = p.nch == 3 ? stage_affine_tabs<3>(p, L, st, threadIdx.x, kBlock) : stage_affine_tabs<1>(p, L, st, threadIdx.x, kBlock);
  // (no table: a range too dense for one threshold per bucket; gamma is 1 here, see launch_affine_map)
  const uint32_t row_elems = p.map_w * p.nch;
  const uint32_t per_row = VEC4 ? row_elems / 4 : row_elems;
  const uint32_t tiles_x = (per_row + kBlock - 1) / kBlock, tiles = tiles_x * p.map_h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t y = t / tiles_x, j = (t - y * tiles_x) * kBlock + threadIdx.x;
    if (j >= per_row) continue;
    const float* src = p.gain_log2 + (size_t)y * row_elems;
    uint8_t* dst = p.out + (size_t)y * p.out_stride * p.nch;
    auto map1 = [&](float q, uint32_t e) -> uint32_t {
      const uint32_t c = p.nch == 3 ? e % 3 : 0;
      if (!tabs) return affine_code<false>(q, p.dev->mn[c], p.dev->range_rcp[c], 1.0f, p.math_tab);
      return c == 0 ? step_code(q, L.tab[0], st[0]) : (c == 1 ? step_code(q, L.tab[1], st[1]) : step_code(q, L.tab[2], st[2]));
    };
    if constexpr (VEC4) {
      const float4 g = *(const float4*)(src + j * 4);
      const uint32_t e = j * 4;
      *(uint32_t*)(dst + e) = map1(g.x, e) | (map1(g.y, e + 1) << 8) | (map1(g.z, e + 2) << 16) | (map1(g.w, e + 3) << 24);
    } else {
      dst[j] = (uint8_t)map1(src[j], j);
    }
  }
}

// Wide variant: one thread maps 16 consecutive map pixels of a row (48 samples for a 3-channel map: twelve
// 16-byte loads in flight per lane, three 16-byte nontemporal stores; the channel of every sample is a
// compile-time constant).
template <int NCH>
__global__ __launch_bounds__(kBlock) void affine_wide_kernel(const AffineParams p) {
  __shared__ AffineLds L;
  StepTab st[3];
  const bool tabs = stage_affine_tabs<NCH>(p, L, st, threadIdx.x, kBlock);
  constexpr int NS = 16 * NCH;  // samples per thread
  const uint32_t row_elems = p.map_w * NCH, per_row = row_elems / NS;
  const uint32_t total = per_row * p.map_h, tiles = (total + kBlock - 1) / kBlock;  // flat: narrow maps still fill the lanes
  typedef uint32_t u4v __attribute__((ext_vector_type(4)));
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t idx = t * kBlock + threadIdx.x;
    if (idx >= total) continue;
    const uint32_t y = idx / per_row, j = idx - y * per_row;
    const float4* src = (const float4*)(p.gain_log2 + (size_t)y * row_elems + (size_t)j * NS);
    uint8_t* dst = p.out + (size_t)y * p.out_stride * NCH + (size_t)j * NS;
    if (!tabs) {  // a range too dense for one threshold per bucket (gamma is 1 here: launch_affine_map): per sample through the
                  // exact evaluation, one load and one store at a time -- register-light, this path must not cost the table
                  // path its occupancy
      const float* sp = (const float*)src;
#pragma unroll 1
      for (int e = 0; e < NS; e++) {
        const int c = e % NCH;
        dst[e] = (uint8_t)affine_code<false>(sp[e], p.dev->mn[c], p.dev->range_rcp[c], 1.0f, p.math_tab);
      }
      continue;
    }
    float4 g[NS / 4];
#pragma unroll
    for (int k = 0; k < NS / 4; k++) g[k] = src[k];
    uint32_t o[NS / 4];
#pragma unroll
    for (int k = 0; k < NS / 4; k++) {
      const float v[4] = {g[k].x, g[k].y, g[k].z, g[k].w};
      uint32_t w = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int c = (4 * k + i) % NCH;
        w |= step_code(v[i], L.tab[c], st[c]) << (8 * i);
      }
      o[k] = w;
    }
#pragma unroll
    for (int k = 0; k < NS / 16; k++)
      __builtin_nontemporal_store((u4v){o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]}, (u4v*)(dst + 16 * k));
  }
}

// The per-sample evaluation for maps with a user gamma != 1 (the host knows: no tables are built, launch_affine_map comes
// here directly).  Kept out of the table kernels because its float64 pow would cost them half their occupancy.
__global__ __launch_bounds__(kBlock) void affine_exact_kernel(const AffineParams p) {
  const uint32_t row_elems = p.map_w * p.nch;
  const size_t total = (size_t)row_elems * p.map_h;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const uint32_t y = (uint32_t)(i / row_elems), e = (uint32_t)(i - (size_t)y * row_elems);
    const uint32_t c = p.nch == 3 ? e % 3 : 0;
    p.out[(size_t)y * p.out_stride * p.nch + e] = (uint8_t)affine_code<true>(p.gain_log2[i], p.dev->mn[c], p.dev->range_rcp[c], p.gamma, p.math_tab);
  }
}

int gen_grid(uint32_t tiles) {
  // residency of the CURRENT device (the entry points make their context's device current first), cached per device id: a
  // process that drives contexts on different devices must not inherit the first one's CU count
  static int resident_of[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (resident_of[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    resident_of[dev] = cus * 3;  // 512-thread workgroups, 25 KB of LDS tables each: three are resident per CU (24 waves, the VGPR limit)
  }
  const int resident = resident_of[dev];
  uint32_t g = tiles < (uint32_t)resident ? tiles : (uint32_t)resident;
  if (g > kMaxGrid) g = kMaxGrid;
  if (g < 1) g = 1;
  return (int)g;
}
bool gen_quad_path(const GenParams& p) {
  return p.scale == 1 && p.sdr.fmt == UHDR_IMG_FMT_12bppYCbCr420 && p.hdr.fmt == UHDR_IMG_FMT_24bppYCbCrP010 && quad_layout_ok(p.sdr) &&
         quad_layout_ok(p.hdr);
}
int gen_grid_of(const GenParams& p) {
  if (gen_quad_path(p)) {
    const uint32_t wave_tiles = ((p.map_w / 2 + 63) / 64) * (p.map_h / 2);
    return gen_grid((wave_tiles + kGenBlock / 64 - 1) / (kGenBlock / 64));
  }
  return gen_grid(((p.map_w + kGenBlock - 1) / kGenBlock) * p.map_h);
}

template <int SDRF, int HDRF>
void launch_gen(const GenParams& p, bool two_pass, int grid, float* partials, hipStream_t s) {
  if (two_pass) hipLaunchKernelGGL((generate_kernel<SDRF, HDRF, true>), dim3(grid), dim3(kGenBlock), 0, s, p, partials);
  else hipLaunchKernelGGL((generate_kernel<SDRF, HDRF, false>), dim3(grid), dim3(kGenBlock), 0, s, p, partials);
}
template <bool TWO_PASS, int GM, int MC>
void launch_quad_l(const GenParams& p, int grid, float* partials, hipStream_t s) {
  if (p.hdr_inv_lut) hipLaunchKernelGGL((generate_quad_kernel<TWO_PASS, GM, MC, true>), dim3(grid), dim3(kGenBlock), 0, s, p, partials);
  else hipLaunchKernelGGL((generate_quad_kernel<TWO_PASS, GM, MC, false>), dim3(grid), dim3(kGenBlock), 0, s, p, partials);
}
template <bool TWO_PASS, int GM>
void launch_quad_m(const GenParams& p, int grid, float* partials, hipStream_t s) {
  if (p.multichannel) launch_quad_l<TWO_PASS, GM, 1>(p, grid, partials, s);
  else launch_quad_l<TWO_PASS, GM, 0>(p, grid, partials, s);
}
template <bool TWO_PASS>
void launch_quad(const GenParams& p, int grid, float* partials, hipStream_t s) {
  if (p.sdr_gamut_on) launch_quad_m<TWO_PASS, 1>(p, grid, partials, s);
  else if (p.hdr_gamut_on) launch_quad_m<TWO_PASS, 2>(p, grid, partials, s);
  else launch_quad_m<TWO_PASS, 0>(p, grid, partials, s);
}
template <int SDRF>
void launch_gen_h(const GenParams& p, bool two_pass, int grid, float* partials, hipStream_t s) {
  switch (p.hdr.fmt) {
    case UHDR_IMG_FMT_24bppYCbCrP010: return launch_gen<SDRF, UHDR_IMG_FMT_24bppYCbCrP010>(p, two_pass, grid, partials, s);
    case UHDR_IMG_FMT_32bppRGBA1010102: return launch_gen<SDRF, UHDR_IMG_FMT_32bppRGBA1010102>(p, two_pass, grid, partials, s);
    default: return launch_gen<SDRF, -1>(p, two_pass, grid, partials, s);
  }
}

}  // namespace

int gen_partials_count(const GenParams& p) { return gen_grid_of(p); }

// Two-pass: writes the ratio plane p.gain_log2 and gen_partials_count(p) x 6 ratio extrema at p.minmax + 6 (the host
// layer allocates 6 + 2048 * 6 floats); launch_minmax_table turns them into the reference's six log2 extrema.
hipError_t launch_generate_gainmap(const GenParams& p, bool two_pass, hipStream_t s) {
  float* partials = two_pass ? p.minmax + 6 : nullptr;
  const int grid = gen_grid_of(p);
  if (gen_quad_path(p)) {  // the API-1 default: quad lanes, coalesced plane loads
    if (two_pass) launch_quad<true>(p, grid, partials, s);
    else launch_quad<false>(p, grid, partials, s);
    return hipGetLastError();
  }
  switch (p.sdr.fmt) {
    case UHDR_IMG_FMT_12bppYCbCr420: launch_gen_h<UHDR_IMG_FMT_12bppYCbCr420>(p, two_pass, grid, partials, s); break;
    case UHDR_IMG_FMT_32bppRGBA8888: launch_gen_h<UHDR_IMG_FMT_32bppRGBA8888>(p, two_pass, grid, partials, s); break;
    default: launch_gen_h<-1>(p, two_pass, grid, partials, s); break;
  }
  return hipGetLastError();
}

hipError_t launch_minmax_table(const MinmaxTableParams& p, hipStream_t s) {
  hipLaunchKernelGGL(minmax_table_kernel, dim3(3), dim3(kTabBlock), 0, s, p);
  return hipGetLastError();
}

hipError_t launch_affine_map(const AffineParams& p, hipStream_t s) {
  const uint32_t row_elems = p.map_w * p.nch;
  if (p.gamma != 1.0f) {
    hipLaunchKernelGGL(affine_exact_kernel, dim3(2048), dim3(kBlock), 0, s, p);
    return hipGetLastError();
  }
  if ((p.nch == 1 || p.nch == 3) && p.map_w % 16 == 0 && ((size_t)p.out_stride * p.nch) % 16 == 0 && (((uintptr_t)p.out & 15) == 0) &&
      (((uintptr_t)p.gain_log2 & 15) == 0)) {
    const uint32_t total = (p.map_w / 16) * p.map_h;
    const uint32_t tiles = (total + kBlock - 1) / kBlock;
    const int grid = (int)(tiles < 2048u ? (tiles ? tiles : 1u) : 2048u);  // resident-sized: every workgroup stages the tables once
    if (p.nch == 3) hipLaunchKernelGGL((affine_wide_kernel<3>), dim3(grid), dim3(kBlock), 0, s, p);
    else hipLaunchKernelGGL((affine_wide_kernel<1>), dim3(grid), dim3(kBlock), 0, s, p);
    return hipGetLastError();
  }
  const bool vec4 = (row_elems % 4 == 0) && ((p.out_stride * p.nch) % 4 == 0) && (((uintptr_t)p.out & 3) == 0) &&
                    (((uintptr_t)p.gain_log2 & 15) == 0);
  const uint32_t per_row = vec4 ? row_elems / 4 : row_elems;
  const uint32_t tiles = ((per_row + kBlock - 1) / kBlock) * p.map_h;
  const int grid = (int)(tiles < 2048u ? (tiles ? tiles : 1u) : 2048u);
  if (vec4) hipLaunchKernelGGL((affine_kernel<true>), dim3(grid), dim3(kBlock), 0, s, p);
  else hipLaunchKernelGGL((affine_kernel<false>), dim3(grid), dim3(kBlock), 0, s, p);
  return hipGetLastError();
}

}  // namespace uhdr
