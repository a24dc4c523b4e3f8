// Image effects on gfx950: rotate (90 / 180 / 270 clockwise), mirror, crop, resize.
// Reference: /root/reference/lib/src/editorhelper.cpp:20-87 (rotate_buffer_clockwise, mirror_buffer, crop_buffer,
// resize_buffer -- the nearest-sample resize of the effects chain) applied plane by plane by apply_rotate / apply_mirror /
// apply_crop / apply_resize (:210-520).  All four are pure element remaps dst[i][j] = src[f(i, j)] over planes of 1-, 2-,
// 4- or 8-byte elements (P010's interleaved chroma travels as one 4-byte element per sample pair), so ONE kernel covers
// them: a lane produces a run of consecutive destination elements of a row (coalesced stores; the gathers of a rotation
// walk a source column, which the L2 absorbs).  SURVEY.md 8f-3.
#include "uhdr_types.h"

namespace uhdr {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void effect_remap_kernel(const EffectPlane p) {
  const T* __restrict__ src = (const T*)p.src;
  T* __restrict__ dst = (T*)p.dst;
  const uint32_t tiles_x = (p.dst_w + 255u) / 256u, tiles = tiles_x * p.dst_h;
  for (uint32_t t = blockIdx.x; t < tiles; t += gridDim.x) {
    const uint32_t i = t / tiles_x, j = (t - i * tiles_x) * 256u + threadIdx.x;
    if (j >= p.dst_w) continue;
    size_t s;
    switch (p.mode) {
      case 0: s = (size_t)(p.src_h - j - 1) * p.src_stride + i; break;                    // rotate 90
      case 1: s = (size_t)(p.src_h - i - 1) * p.src_stride + (p.src_w - j - 1); break;    // rotate 180
      case 2: s = (size_t)j * p.src_stride + (p.src_w - i - 1); break;                    // rotate 270
      case 3: s = (size_t)(p.src_h - i - 1) * p.src_stride + j; break;                    // mirror vertical (flip over the x axis)
      case 4: s = (size_t)i * p.src_stride + (p.src_w - j - 1); break;                    // mirror horizontal
      case 5: s = (size_t)(p.a1 + i) * p.src_stride + (p.a0 + j); break;                  // crop: a0 = left, a1 = top
      default: s = (size_t)i * p.a1 * p.src_stride + (size_t)j * p.a0; break;             // resize: a0 = src_w / dst_w, a1 = src_h / dst_h
    }
    dst[(size_t)i * p.dst_stride + j] = src[s];
  }
}

}  // namespace

hipError_t launch_effect_plane(const EffectPlane& p, hipStream_t s) {
  const uint32_t tiles = ((p.dst_w + 255u) / 256u) * p.dst_h;
  if (tiles == 0) return hipSuccess;
  const int grid = (int)(tiles < 16384u ? tiles : 16384u);
  switch (p.elem) {
    case 1: hipLaunchKernelGGL((effect_remap_kernel<uint8_t>), dim3(grid), dim3(256), 0, s, p); break;
    case 2: hipLaunchKernelGGL((effect_remap_kernel<uint16_t>), dim3(grid), dim3(256), 0, s, p); break;
    case 4: hipLaunchKernelGGL((effect_remap_kernel<uint32_t>), dim3(grid), dim3(256), 0, s, p); break;
    case 8: hipLaunchKernelGGL((effect_remap_kernel<uint64_t>), dim3(grid), dim3(256), 0, s, p); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace uhdr
