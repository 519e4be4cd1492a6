// utf8_swar.h — UTF-8 structure tests on 32-bit words (SWAR), shared by the two decode passes (decode.h).
//
// Replaces the per-byte tests of vkcom::decode_utf8 (utf8.cpp:54-90) for four byte positions at once; verdicts
// are identical to decode_one (common.h), which tests/test_utf8_swar.py checks on the host over every
// (lead, second byte) pair and millions of random windows.  Host-compilable: the one GPU builtin used
// (v_alignbyte_b32) has a plain C++ statement for the CPU test.
#pragma once
#include <cstdint>

#if defined(__HIP_DEVICE_COMPILE__)
#define WP_HD __host__ __device__ __forceinline__
#define WP_ALIGNBYTE(hi, lo, sh) __builtin_amdgcn_alignbyte((hi), (lo), (sh))
#else
#if defined(__HIPCC__)
#define WP_HD __host__ __device__ inline
#else
#define WP_HD inline
#endif
// ({hi, lo} >> 8 * (sh & 3)), low 32 bits — what v_alignbyte_b32 computes
#define WP_ALIGNBYTE(hi, lo, sh) \
  static_cast<uint32_t>(((static_cast<uint64_t>(hi) << 32) | static_cast<uint64_t>(lo)) >> (8 * ((sh) & 3u)))
#endif

namespace wp {

constexpr uint32_t kHi = 0x80808080u;

// per byte position of a word (bit 7 of byte j): does a valid 1 / 2 / 3 / 4-byte sequence start here?
// Same verdicts as decode_one (utf8.cpp:54-90): w0 = the word, nx = the four bytes behind it.
struct Utf8Starts {
  uint32_t v1, v2, v3, v4;
};
WP_HD Utf8Starts utf8_starts(uint32_t w0, uint32_t nx) {
  const uint32_t w1 = WP_ALIGNBYTE(nx, w0, 1u);  // byte j of w1 = the byte behind byte j of w0
  const uint32_t w2 = WP_ALIGNBYTE(nx, w0, 2u);
  const uint32_t w3 = WP_ALIGNBYTE(nx, w0, 3u);
  auto cont = [](uint32_t x) { return x & ~(x << 1) & kHi; };  // 10xxxxxx
  const uint32_t c1 = cont(w1), c2 = cont(w2), c3 = cont(w3);
  const uint32_t s1 = w0 << 1, s2 = w0 << 2, s3 = w0 << 3, s4 = w0 << 4;  // bit 7 of a byte <- its bit 6 / 5 / 4 / 3
  const uint32_t l2 = w0 & s1 & ~s2 & kHi;                                  // 110xxxxx
  const uint32_t l3 = w0 & s1 & s2 & ~s3 & kHi;                             // 1110xxxx
  const uint32_t l4 = w0 & s1 & s2 & s3 & ~s4 & kHi;                        // 11110xxx
  auto byte_zero = [](uint32_t x) { return ~(x + 0x7f7f7f7fu) & kHi; };     // (bytes of x <= 0x7f: no carries)
  const uint32_t over2 = byte_zero(w0 & 0x1e1e1e1eu);                       // C0, C1: code point < 0x80
  const uint32_t lo4 = w0 & 0x0f0f0f0fu;
  const uint32_t n5 = w1 << 2;                                              // bit 5 of the next byte
  // E0 80..9F: code point < 0x800; ED A0..BF: surrogates
  const uint32_t bad3 = (byte_zero(lo4) & ~n5) | (byte_zero(lo4 ^ 0x0d0d0d0du) & n5);
  const uint32_t lo3 = w0 & 0x07070707u;
  const uint32_t n54 = (w1 << 2) | (w1 << 3);                               // next byte >= 0x90 (a continuation byte)
  // F0 80..8F: code point < 0x10000; F4 90..: > U+10FFFF; F5..F7
  const uint32_t bad4 = (byte_zero(lo3) & ~n54) | (byte_zero(lo3 ^ 0x04040404u) & n54) | ((w0 << 5) & ((w0 << 6) | (w0 << 7)));
  Utf8Starts r;
  r.v1 = ~w0 & kHi;
  r.v2 = l2 & c1 & ~over2;
  r.v3 = l3 & c1 & c2 & ~bad3 & kHi;
  r.v4 = l4 & c1 & c2 & c3 & ~bad4 & kHi;
  return r;
}
// the code point of a VALID sequence: x = the four bytes from its lead on
WP_HD uint32_t utf8_value(uint32_t x) {
  const uint32_t b0 = x & 0xffu, b1 = (x >> 8) & 0x3fu, b2 = (x >> 16) & 0x3fu, b3 = (x >> 24) & 0x3fu;
  if (b0 < 0x80u) return b0;
  if (b0 < 0xe0u) return ((b0 & 0x1fu) << 6) | b1;
  if (b0 < 0xf0u) return ((b0 & 0x0fu) << 12) | (b1 << 6) | b2;
  return ((b0 & 0x07u) << 18) | (b1 << 12) | (b2 << 6) | b3;
}
// bit 7 of byte j -> bit j
WP_HD uint32_t byte_mask4(uint32_t m) {
  const uint32_t x = m >> 7;
  return (x | (x >> 7) | (x >> 14) | (x >> 21)) & 0xfu;
}

}  // namespace wp
