// ref_shim.cpp — C entry points into the reference's OWN utils/utf8 code, so that
// tests can diff the oracle's restatement against it.  Compiled only when
// /root/reference is present (oracle/Makefile target _ref/librefutils.so); the
// reference sources are compiled where they lie and are never copied.
// TEST INFRASTRUCTURE ONLY.
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "third_party/utf8.hpp"
#include "utils.hpp"

extern "C" {

int ref_is_space(uint32_t c) { return vkcom::is_space(c); }
int ref_is_punctuation(uint32_t c) { return vkcom::is_punctuation(c); }
int ref_is_spacing_char(uint32_t c) { return vkcom::is_spacing_char(c); }

// utf8.cpp:130-147 — out must hold nbytes entries
size_t ref_decode_utf8(const char *s, size_t nbytes, uint32_t *out) {
  std::vector<uint32_t> v = vkcom::decode_utf8(s, s + nbytes);
  std::memcpy(out, v.data(), v.size() * sizeof(uint32_t));
  return v.size();
}

// utils.cpp:37-79 with a real pool (exercises the chunked path for >= 1e7 bytes)
size_t ref_parse_text(const char *s, size_t nbytes, uint32_t *out) {
  std::vector<uint32_t> v = utils::parseText(s, nbytes, utils::globalThreadPool());
  std::memcpy(out, v.data(), v.size() * sizeof(uint32_t));
  return v.size();
}

// utils.cpp:81-121 — one token: returns flags (bit0 prefix, bit1 special, bit2
// malformed), word into out (capacity nbytes), length into *len; -1 if it throws.
int ref_token(const char *s, size_t nbytes, uint32_t *out, int64_t *len) {
  try {
    utils::WordPieceToken t{std::string(s, nbytes)};
    std::memcpy(out, t.word.data(), t.word.size() * sizeof(uint32_t));
    *len = static_cast<int64_t>(t.word.size());
    return (t.is_prefix ? 1 : 0) | (t.is_special ? 2 : 0) | (t.is_malformed ? 4 : 0);
  } catch (const std::runtime_error &) {
    return -1;
  }
}

// utils.cpp:108-121 parseVocab → unk_token_id (or -2 if it throws)
int ref_unk_id(const char *buf, const int64_t *off, int64_t V) {
  std::vector<std::string> vocab;
  for (int64_t i = 0; i < V; i++) vocab.emplace_back(buf + off[i], static_cast<size_t>(off[i + 1] - off[i]));
  try {
    return utils::parseVocab(vocab).unk_token_id;
  } catch (const std::runtime_error &) {
    return -2;
  }
}
}
