// tests/cpp/zero_pages_check.cpp -- facade/uhdr_zero_pages.h against the std::vector<uint8_t> it stands in for:
// the same random sequence of clear / resize / writes applied to both, contents compared after every step
// (resize's new elements must be zero whether the block is fresh, recycled after clear(), grown in place or reallocated).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "uhdr_zero_pages.h"

int main() {
  std::mt19937 rng(20250926);
  uhdr_zero_pages::bytes b;
  std::vector<uint8_t> v;
  size_t steps = 0;
  for (int it = 0; it < 4000; it++) {
    const unsigned op = rng() % 10;
    if (op == 0) {
      b.clear();
      v.clear();
    } else if (op < 6) {
      // sizes on both sides of glibc's mmap threshold, now and then a big one
      const size_t n = (rng() % 7 == 0) ? (size_t)(rng() % (6u << 20)) : (size_t)(rng() % 70000u);
      b.resize(n);
      v.resize(n);
    } else if (!v.empty()) {
      for (int k = 0; k < 64; k++) {
        const size_t i = rng() % v.size();
        const uint8_t x = (uint8_t)(rng() | 1u);
        b.data()[i] = x;
        v[i] = x;
      }
      // dirty the whole block now and then, so that a recycled or shrunk-then-grown block would show stale bytes
      if (rng() % 5 == 0) {
        memset(b.data(), 0xa5, b.size());
        memset(v.data(), 0xa5, v.size());
      }
    }
    if (b.size() != v.size() || (v.size() && memcmp(b.data(), v.data(), v.size()) != 0)) {
      printf("MISMATCH at step %d (op %u): sizes %zu / %zu\n", it, op, b.size(), v.size());
      return 1;
    }
    steps++;
  }
  // the deleter / allocation pair of uhdr_memory_block
  for (size_t n : {(size_t)0, (size_t)1, (size_t)4096, (size_t)(50u << 20)}) {
    uint8_t* p = uhdr_zero_pages::zeroed(n);
    for (size_t i = 0; i < n; i += 4093)
      if (p[i] != 0) {
        printf("zeroed(%zu) is not zero at %zu\n", n, i);
        return 1;
      }
    if (n) p[n - 1] = 7;
    uhdr_zero_pages::block_free()(p);
  }
  printf("zero_pages ok: %zu steps\n", steps);
  return 0;
}
