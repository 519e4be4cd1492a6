// test_utf8_swar.cpp — host check of csrc/utf8_swar.h (the word-at-a-time UTF-8 structure tests of the two decode
// kernels) against the oracle's sequential decoder (oracle/wp_oracle.c: wpo_decode_utf8, itself pinned against the
// reference's utf8.cpp in tests/test_oracle.py).  The property the kernels rely on: the positions where a valid
// sequence starts, found independently per byte position, are exactly the decode points of the sequential decoder
// (a valid sequence only ever swallows continuation bytes, and those never start one), with the same code points.
// Built and run by tests/test_utf8_swar.py (g++, no GPU).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../wordpiece_amd/csrc/utf8_swar.h"

extern "C" size_t wpo_decode_utf8(const uint8_t *s, size_t nbytes, uint32_t *out, int *had_invalid);

// the kernels' view of a buffer: words, the four bytes behind each word, bytes behind the end read as zero
static std::vector<uint32_t> swar_decode(const std::vector<uint8_t> &buf, size_t &consumed) {
  std::vector<uint8_t> b(buf);
  b.resize((buf.size() + 3) / 4 * 4 + 8, 0);
  std::vector<uint32_t> out;
  consumed = 0;
  for (size_t p = 0; p < buf.size(); p += 4) {
    uint32_t w0, nx;
    std::memcpy(&w0, &b[p], 4);
    std::memcpy(&nx, &b[p + 4], 4);
    const wp::Utf8Starts u = wp::utf8_starts(w0, nx);
    const uint32_t all = wp::byte_mask4(u.v1 | u.v2 | u.v3 | u.v4);
    for (unsigned j = 0; j < 4 && p + j < buf.size(); j++) {
      if (!((all >> j) & 1u)) continue;
      out.push_back(wp::utf8_value(WP_ALIGNBYTE(nx, w0, j)));
      consumed += ((u.v1 >> (8 * j + 7)) & 1u) * 1 + ((u.v2 >> (8 * j + 7)) & 1u) * 2 + ((u.v3 >> (8 * j + 7)) & 1u) * 3 +
                  ((u.v4 >> (8 * j + 7)) & 1u) * 4;
    }
  }
  return out;
}

static long checks = 0;
static bool same(const std::vector<uint8_t> &buf, const char *what) {
  std::vector<uint32_t> exp(buf.size() + 1);
  int invalid = 0;
  exp.resize(wpo_decode_utf8(buf.data(), buf.size(), exp.data(), &invalid));
  size_t consumed = 0;
  const std::vector<uint32_t> got = swar_decode(buf, consumed);
  checks++;
  if (got != exp || (consumed != buf.size()) != (invalid != 0)) {
    std::printf("MISMATCH (%s): %zu bytes, %zu vs %zu code points, consumed %zu, invalid %d\n", what, buf.size(), got.size(),
                exp.size(), consumed, invalid);
    return false;
  }
  return true;
}

int main() {
  bool ok = true;
  // every (lead, second byte) pair with representative third / fourth bytes, at every byte position of a word
  const int reps[] = {0x00, 0x41, 0x7f, 0x80, 0x8f, 0x90, 0x9f, 0xa0, 0xbf, 0xc0, 0xc2, 0xe0, 0xed, 0xf0, 0xf4, 0xf5, 0xff};
  for (int shift = 0; shift < 4 && ok; shift++) {
    std::vector<uint8_t> buf;
    for (int b0 = 0x80; b0 < 256; b0++) {
      for (int b1 = 0; b1 < 256; b1++) {
        for (int b2 : reps) {
          for (int b3 : reps) {
            for (int k = 0; k < shift; k++) buf.push_back('x');
            buf.push_back(static_cast<uint8_t>(b0));
            buf.push_back(static_cast<uint8_t>(b1));
            buf.push_back(static_cast<uint8_t>(b2));
            buf.push_back(static_cast<uint8_t>(b3));
            buf.push_back(' ');
          }
        }
      }
      ok = ok && same(buf, "pairs");
      buf.clear();
    }
  }
  // random buffers: ASCII, continuation bytes and arbitrary bytes mixed; every tail length (truncated sequences)
  std::mt19937 rnd(20261005);
  for (int it = 0; it < 4000 && ok; it++) {
    std::vector<uint8_t> buf(1 + rnd() % 5000);
    const unsigned mode = rnd() % 4;
    for (auto &c : buf) {
      const uint32_t r = rnd();
      c = mode == 0 ? (r & 0xff) : (r & 0x300) == 0 ? (r & 0x7f) : (r & 0x400) ? (0x80 | (r & 0x3f)) : (0xc0 | (r & 0x3f));
    }
    ok = ok && same(buf, "random");
  }
  // well-formed text of every length class, cut at every offset
  {
    const char *s = "a\xc3\xa9\xd0\xb6\xe4\xb8\xad\xe2\x96\x81\xf0\x9f\x98\x80z\xed\x9f\xbf\xee\x80\x80\xf4\x8f\xbf\xbf";
    const size_t n = std::strlen(s);
    for (size_t a = 0; a < n && ok; a++) {
      for (size_t b = a; b <= n && ok; b++) ok = ok && same(std::vector<uint8_t>(s + a, s + b), "cuts");
    }
  }
  std::printf("%ld buffers checked: %s\n", checks, ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}
