// Load-instruction throughput by width (data L1/L2 resident): cycles per wave-load per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;

template <typename T>
__global__ __launch_bounds__(256) void k_load(const uint8_t* __restrict__ src, uint32_t* out, uint32_t span, uint32_t lane_stride) {
  // each lane reads sizeof(T) bytes at lane*lane_stride (+ moving window), 8 independent loads per iteration
  const uint32_t lane = threadIdx.x;
  uint32_t acc = 0;
  uint32_t base = (blockIdx.x * 4096u) % span;
  for (int i = 0; i < ITERS; i++) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = *(const T*)(src + ((base + k * 1024u + lane * lane_stride) % span));
#pragma unroll
    for (int k = 0; k < 8; k++) acc += (uint32_t)(*(const uint8_t*)&v[k]);
    base = (base + 8192u) % span;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <typename F>
float time_ms(F f, int reps = 3) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; i++) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const double clk = pr.clockRate * 1e3;
  uint8_t* src; uint32_t* out;
  const uint32_t span = 1u << 20;  // 1 MiB window: L2 resident
  CK(hipMalloc(&src, span + 65536)); CK(hipMemset(src, 1, span + 65536)); CK(hipMalloc(&out, 2048 * 256 * 4));
#define RUNL(T, name, stride) { float ms = time_ms([&] { hipLaunchKernelGGL((k_load<T>), dim3(2048), dim3(256), 0, 0, src, out, span, (uint32_t)stride); }); \
    double winst = 2048.0 * 4 * ITERS * 8; \
    printf("%-10s lane stride %2d B: %7.3f ms  %6.1f cycles per wave-load per CU  (%.0f GB/s useful)\n", name, (int)stride, ms, ms * 1e-3 * clk / (winst / pr.multiProcessorCount), winst * 64 * sizeof(T) / (ms * 1e-3) / 1e9); }
  RUNL(uint8_t, "ubyte", 1) RUNL(uint16_t, "ushort", 2) RUNL(uint32_t, "dword", 4) RUNL(uint2, "dwordx2", 8) RUNL(uint4, "dwordx4", 16)
  RUNL(uint8_t, "ubyte", 4) RUNL(uint16_t, "ushort", 4) RUNL(uint8_t, "ubyte", 0)
  return 0;
}
