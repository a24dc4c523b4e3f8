#!/usr/bin/env python
"""Generates tools/_exp/: a copy of the decode kernel with a per-wave end-time record (s_memrealtime + XCC id at wave exit),
an optional s_setprio progress balancer (-DEXP_PRIO) and an optional dynamic row queue (KB_DYN=1 at run time), plus the
kbench driver that prints the end-time statistics.  The product sources are not touched; tools/_exp/ is git-ignored.

    python tools/make_wavetimes_exp.py
    cd tools && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DEXP_PRIO] -I../libultrahdr_amd/csrc -I../include -I. \
        -Wno-unused-variable -o kb_wt _exp/kbench_wt.cpp
    KB_N=300 ./kb_wt C            # results: profiles/r02_wave_end_times.txt
"""
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.makedirs(os.path.join(HERE, "_exp"), exist_ok=True)

src = open(os.path.join(ROOT, "libultrahdr_amd/csrc/apply_gainmap.hip")).read()
src = src.replace("namespace uhdr {\n", "namespace uhdr {\n__device__ unsigned long long* g_wave_times = nullptr;\n"
                  "__device__ unsigned int* g_row_ctr = nullptr;  // [0] finished waves, [1 + frame * strips + strip] next quad row\n", 1)
old = '''    Raw a = fetch(qy0, H0{});  // in flight while the tables are staged
    stage_tables();
    if (!live) return;
    for (uint32_t i = 0; i < n_iter; i++) {
      const Raw b = fetch(qy0 + i * groups, H1{});
      process(a, H0{});
      a = fetch(qy0 + (i + 1) * groups, H0{});
      process(b, H1{});
    }
  } else {'''
new = '''    if (g_row_ctr) {  // dynamic row queue: one atomic per quad row and column strip (8x slower: same-address atomics)
      unsigned int* ctr = g_row_ctr + 1 + frame * strips_x + sx;
      auto claim = [&]() -> uint32_t {
        uint32_t v = 0;
        if (lane == 0) v = atomicAdd(ctr, 1u);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
      };
      uint32_t r = live ? claim() : qh;
      Raw a = fetch(r, H0{});
      stage_tables();
      if (!live) return;
      while (r < qh) {
        const uint32_t rn = claim();
        const Raw b = fetch(r, H1{});
        process(a, H0{});
        a = fetch(rn, H0{});
        process(b, H1{});
        r = rn;
      }
      if (lane == 0) {
        __threadfence();
        const uint32_t total = per_frame * p.n_frames;
        if (atomicAdd(g_row_ctr, 1u) == total - 1) {  // the last wave out resets the counters for the next launch
          for (uint32_t i = 0; i < strips_x * p.n_frames; i++) g_row_ctr[1 + i] = 0;
          g_row_ctr[0] = 0;
          __threadfence();
        }
      }
    } else {
      Raw a = fetch(qy0, H0{});  // in flight while the tables are staged
      stage_tables();
      if (!live) return;
      for (uint32_t i = 0; i < n_iter; i++) {
#ifdef EXP_PRIO
        {  // a wave that is behind outranks one that is ahead: progress of a SIMD's waves evens out
          const uint32_t q4 = (i * 4u) / n_iter;  // quartile of the wave's own progress
          if (q4 == 0) __builtin_amdgcn_s_setprio(3);
          else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
          else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
#endif
        const Raw b = fetch(qy0 + i * groups, H1{});
        process(a, H0{});
        a = fetch(qy0 + (i + 1) * groups, H0{});
        process(b, H1{});
      }
    }
    if (g_wave_times && lane == 0) {
      unsigned long long t = __builtin_amdgcn_s_memrealtime();
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      g_wave_times[wave * 2] = t;
      g_wave_times[wave * 2 + 1] = ((unsigned long long)xcc << 32);
    }
  } else {'''
assert old in src, "the kernel's main loop changed: update this script"
open(os.path.join(HERE, "_exp/apply_gainmap.hip"), "w").write(src.replace(old, new))

kb = open(os.path.join(HERE, "kbench.cpp")).read()
kb = kb.replace('#include "apply_gainmap.hip"', '#include "_exp/apply_gainmap.hip"')
kb = kb.replace("  hipStream_t st; CK(hipStreamCreate(&st));",
                '  if (getenv("KB_DYN")) { unsigned int* dc; CK(hipMalloc(&dc, 4096 * 4)); CK(hipMemset(dc, 0, 4096 * 4)); '
                'CK(hipMemcpyToSymbol(HIP_SYMBOL(uhdr::g_row_ctr), &dc, sizeof dc)); }\n  hipStream_t st; CK(hipStreamCreate(&st));')
tail = r'''  {  // one launch with per-wave end times
    const size_t nw = 16384;
    unsigned long long* d; CK(hipMalloc(&d, nw * 16)); CK(hipMemset(d, 0, nw * 16));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(uhdr::g_wave_times), &d, sizeof d));
    launch(0); CK(hipStreamSynchronize(st));
    CK(hipMemset(d, 0, nw * 16));
    launch(1); CK(hipStreamSynchronize(st));
    std::vector<unsigned long long> h(nw * 2); CK(hipMemcpy(h.data(), d, nw * 16, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0; size_t n = 0;
    for (size_t i = 0; i < nw; i++) if (h[2 * i]) { n++; if (h[2 * i] < t0) t0 = h[2 * i]; if (h[2 * i] > t1) t1 = h[2 * i]; }
    printf("waves %zu: end times span %.2f us (first finisher -> last)\n", n, (t1 - t0) / 100.0);
    double sum[8] = {0}; unsigned long long mx[8] = {0}, mn[8]; int cnt[8] = {0}; for (int x = 0; x < 8; x++) mn[x] = ~0ull;
    for (size_t i = 0; i < nw; i++) if (h[2 * i]) { int x = (int)((h[2 * i + 1] >> 32) & 7); cnt[x]++; sum[x] += (double)(h[2 * i] - t0); if (h[2 * i] > mx[x]) mx[x] = h[2 * i]; if (h[2 * i] < mn[x]) mn[x] = h[2 * i]; }
    for (int x = 0; x < 8; x++) if (cnt[x]) printf("  xcc %d: %5d waves, end mean %.2f us, first %.2f, last %.2f (relative to the first finisher)\n", x, cnt[x], sum[x] / cnt[x] / 100.0, (mn[x] - t0) / 100.0, (mx[x] - t0) / 100.0);
    int hist[64] = {0}; for (size_t i = 0; i < nw; i++) if (h[2 * i]) { int b = (int)((h[2 * i] - t0) / 100); if (b > 63) b = 63; hist[b]++; }
    printf("  end-time histogram (1 us bins):"); for (int b = 0; b < 64; b++) if (hist[b]) printf(" %d:%d", b, hist[b]); printf("\n");
  }
  return 0;
}'''
assert kb.rstrip().endswith("return 0;\n}")
kb = kb.rstrip()[: -len("return 0;\n}")].rstrip() + "\n" + tail + "\n"
open(os.path.join(HERE, "_exp/kbench_wt.cpp"), "w").write(kb)
print("wrote tools/_exp/apply_gainmap.hip and tools/_exp/kbench_wt.cpp")
