// common.h — shared definitions for the HIP Linear-WordPiece path (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace wp {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

struct HipError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

#define WP_HIP(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      throw ::wp::HipError(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ \
                           ":" + std::to_string(__LINE__) + ")");                            \
    }                                                                                         \
  } while (0)

#define WP_LAUNCH_CHECK() WP_HIP(hipGetLastError())

inline unsigned cdiv(size_t a, size_t b) { return static_cast<unsigned>((a + b - 1) / b); }

// Debug build (-DWP_DEBUG_BOUNDS, libwordpiece_amd_dbg.so): the stores / gathers whose address comes out of
// a computed table (radix offsets, rank destinations, token ids, list slots) check it first; a violation is
// counted per site and skipped instead of faulting the GPU, and the encode fails with the counts.
// Release build: the checks compile to nothing.
enum BoundSite { kSiteRadixScatter = 0, kSiteRankStore = 1, kSiteTokenId = 2, kSiteListSlot = 3, kBoundSites = 4 };
#ifdef WP_DEBUG_BOUNDS
__device__ unsigned int g_wp_oob[kBoundSites];
__device__ __forceinline__ bool wp_in_bounds(bool ok, int site) {
  if (!ok) atomicAdd(&g_wp_oob[site], 1u);
  return ok;
}
#else
__device__ __forceinline__ bool wp_in_bounds(bool, int) { return true; }
#endif

// ---- character classes: utf8.cpp:10-29 of the reference --------------------------------
constexpr uint32_t kSpaceToken = 9601;  // utf8.hpp:14 (U+2581)
constexpr uint32_t kInvalidUnicode = 0x110000;

__host__ __device__ inline bool is_space(uint32_t c) {  // "C"-locale isspace, or U+2581
  return (c < 256 && ((c >= 0x09 && c <= 0x0D) || c == 0x20)) || c == kSpaceToken;
}
__host__ __device__ inline bool is_punctuation(uint32_t c) {  // "C"-locale ispunct + extras
  if (c < 256) {
    if ((c >= 0x21 && c <= 0x2F) || (c >= 0x3A && c <= 0x40) || (c >= 0x5B && c <= 0x60)
        || (c >= 0x7B && c <= 0x7E)) {
      return true;
    }
  }
  return c == 183 || c == 171 || c == 187 || c == 8249 || c == 8250 || (8208 <= c && c <= 8248);
}
__host__ __device__ inline bool is_chinese(uint32_t c) {
  return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF)
         || (c >= 0x20000 && c <= 0x2A6DF) || (c >= 0x2A700 && c <= 0x2B73F)
         || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B820 && c <= 0x2CEAF)
         || (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}
__host__ __device__ inline bool is_spacing_char(uint32_t c) {
  return is_space(c) || is_punctuation(c) || is_chinese(c);
}

// per-text-position class byte written by the decode kernel
constexpr uint8_t kClsSpace = 1;    // is_space
constexpr uint8_t kClsSpacing = 2;  // is_spacing_char
constexpr uint8_t kClsSoft = 4;     // spacing char that occurs inside an eligible multi-char token
constexpr uint8_t kClsPunct = 8;    // is_punctuation (the fast path's word rule, fast.cpp:56)
// derived bits, merged into the class bytes by the Linear path's anchor kernels (walk.h) so that the walk reads ONE byte
// per position it lands on: a word-prefix position (linear.cpp:215-219); an anchor of the walk (class or coverage rule)
constexpr uint8_t kClsWordPrefix = 16;
constexpr uint8_t kClsAnchor = 32;

// ---- UTF-8: utf8.cpp:31-90 ----------------------------------------------------------------
// Decodes the sequence starting at p[0] with `size` bytes available.  Returns the code point or
// kInvalidUnicode.  (Continuation bytes never start a sequence: they are either consumed by a
// valid lead or dropped, so the decode kernel evaluates this at every non-continuation byte.)
__host__ __device__ inline uint32_t decode_one(const uint8_t *p, int64_t size) {
  const uint32_t b0 = p[0];
  if ((b0 & 0x80u) == 0) return b0;
  auto cont = [](uint32_t x) { return (x & 0xc0u) == 0x80u; };
  auto okcp = [](uint32_t x) { return (x < 0xd800) || (0xdfff < x && x < 0x110000); };
  if ((b0 & 0xe0u) == 0xc0u) {
    if (size >= 2 && cont(p[1])) {
      uint32_t cp = ((b0 & 0x1fu) << 6) | (p[1] & 0x3fu);
      if (cp >= 0x80 && okcp(cp)) return cp;
    }
  } else if ((b0 & 0xf0u) == 0xe0u) {
    if (size >= 3 && cont(p[1]) && cont(p[2])) {
      uint32_t cp = ((b0 & 0x0fu) << 12) | ((p[1] & 0x3fu) << 6) | (p[2] & 0x3fu);
      if (cp >= 0x800 && okcp(cp)) return cp;
    }
  } else if ((b0 & 0xf8u) == 0xf0u) {
    if (size >= 4 && cont(p[1]) && cont(p[2]) && cont(p[3])) {
      uint32_t cp = ((b0 & 0x07u) << 18) | ((p[1] & 0x3fu) << 12) | ((p[2] & 0x3fu) << 6) | (p[3] & 0x3fu);
      if (cp >= 0x10000 && okcp(cp)) return cp;
    }
  }
  return kInvalidUnicode;
}

}  // namespace wp
