// facade/uhdr_zero_pages.h -- zero-filled host blocks that do not touch pages which are zero already.
//
// The reference value-initialises every buffer it allocates: uhdr_memory_block (ultrahdr_api.cpp:45-48, make_unique<uint8_t[]>(n))
// behind all raw and compressed images, and JpegDecoderHelper::mResultBuffer (jpegdecoderhelper.cpp:370-392, vector::resize).
// On fresh mappings that writes zeros over pages the kernel hands out zeroed anyway, and it faults every page in: a 4K
// uhdr_encode spent 7.5 of its 10.5 ms filling a 49.8 MB output buffer of which 4 MB are used; a 4K uhdr_decode 6 ms on decoded
// planes that stay on the device (profiles/r05_api_trace.txt).  calloc gives the same contents -- glibc leaves fresh mappings
// alone and clears recycled chunks -- so the patch (facade/make_patch.py) swaps the allocation, not the semantics.
#ifndef UHDR_ZERO_PAGES_H
#define UHDR_ZERO_PAGES_H

#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>

namespace uhdr_zero_pages {

inline uint8_t* zeroed(size_t n) {
  void* p = calloc(n ? n : 1, 1);
  if (!p) throw std::bad_alloc();
  return static_cast<uint8_t*>(p);
}

struct block_free {  // deleter of uhdr_memory_block::m_buffer
  void operator()(uint8_t* p) const { free(p); }
};

// What JpegDecoderHelper uses of std::vector<JOCTET> for mResultBuffer -- clear / resize / data / size -- with resize's new
// elements zero as value-initialisation leaves them.
class bytes {
 public:
  bytes() = default;
  bytes(const bytes&) = delete;
  bytes& operator=(const bytes&) = delete;
  ~bytes() { free(p_); }
  void clear() { n_ = 0; }
  void resize(size_t n) {
    if (n > n_) {
      if (n_ == 0) {  // nothing to keep: a fresh block instead of clearing the old one
        free(p_);
        p_ = nullptr;
        cap_ = 0;
        p_ = zeroed(n);
        cap_ = n;
      } else if (n > cap_) {
        uint8_t* q = zeroed(n);
        memcpy(q, p_, n_);
        free(p_);
        p_ = q;
        cap_ = n;
      } else {
        memset(p_ + n_, 0, n - n_);
      }
    }
    n_ = n;
  }
  uint8_t* data() { return p_; }
  const uint8_t* data() const { return p_; }
  size_t size() const { return n_; }

 private:
  uint8_t* p_ = nullptr;
  size_t n_ = 0, cap_ = 0;
};

}  // namespace uhdr_zero_pages
#endif
