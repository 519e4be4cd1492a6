// word_piece.cpp — word_piece::linear::* (include/word_piece.hpp) as a thin wrapper over the C ABI.
// Mirrors linear.cpp:330-374 of the reference: same signatures, same error messages.
#include "../../include/word_piece.hpp"

#include <iostream>
#include <stdexcept>

#include "../../include/wordpiece_amd.h"

namespace {
struct VocabHandle {
  wp_vocab *v = nullptr;
  ~VocabHandle() { wp_vocab_destroy(v); }
};
[[noreturn]] void fail() { throw std::runtime_error(wp_last_error()); }
}  // namespace

namespace word_piece::linear {

std::vector<int> encode(const std::string &text, const std::vector<std::string> &vocab) {
  std::vector<const char *> ptrs;
  std::vector<size_t> lens;
  ptrs.reserve(vocab.size());
  lens.reserve(vocab.size());
  for (const std::string &w : vocab) {
    ptrs.push_back(w.data());
    lens.push_back(w.size());
  }
  VocabHandle h;
  if (wp_vocab_create(ptrs.data(), lens.data(), vocab.size(), &h.v) != WP_OK) fail();
  int32_t *ids = nullptr;
  size_t n = 0;
  if (wp_linear_encode(h.v, text.data(), text.size(), &ids, &n) != WP_OK) fail();
  std::vector<int> out(ids, ids + n);
  wp_free(ids);
  return out;
}

std::vector<int> encode(const std::string &text_file, const std::string &vocab_file) {
  int32_t *ids = nullptr;
  size_t n = 0;
  if (wp_linear_encode_file(text_file.c_str(), vocab_file.c_str(), &ids, &n) != WP_OK) fail();
  std::vector<int> out(ids, ids + n);
  wp_free(ids);
  return out;
}

void encodeExternal(const std::string &text_file, const std::string &vocab_file, const std::string &out_file,
                    size_t memory_limit) {
  if (wp_linear_encode_external(text_file.c_str(), vocab_file.c_str(), out_file.c_str(), memory_limit) != WP_OK) {
    fail();
  }
}

}  // namespace word_piece::linear

// word_piece::fast (fast.cpp:159-220 of the reference): same shapes, the GPU trie walk underneath.
namespace word_piece::fast {

std::vector<int> encode(const std::string &text, const std::vector<std::string> &vocab) {
  std::vector<const char *> ptrs;
  std::vector<size_t> lens;
  ptrs.reserve(vocab.size());
  lens.reserve(vocab.size());
  for (const std::string &w : vocab) {
    ptrs.push_back(w.data());
    lens.push_back(w.size());
  }
  VocabHandle h;
  if (wp_vocab_create(ptrs.data(), lens.data(), vocab.size(), &h.v) != WP_OK) fail();
  int32_t *ids = nullptr;
  size_t n = 0;
  if (wp_fast_encode(h.v, text.data(), text.size(), &ids, &n) != WP_OK) fail();
  std::vector<int> out(ids, ids + n);
  wp_free(ids);
  return out;
}

std::vector<int> encode(const std::string &text_file, const std::string &vocab_file) {
  int32_t *ids = nullptr;
  size_t n = 0;
  if (wp_fast_encode_file(text_file.c_str(), vocab_file.c_str(), &ids, &n) != WP_OK) fail();
  std::vector<int> out(ids, ids + n);
  wp_free(ids);
  return out;
}

// fast.cpp:172-187: the token strings of the ids; "##" in front of continuation tokens, a notice on
// stderr and no string for an id outside the vocabulary or a malformed token.
std::vector<std::string> decode(const std::string vocab_file, const std::vector<int> &ids) {
  VocabHandle h;
  if (wp_vocab_from_file(vocab_file.c_str(), &h.v) != WP_OK) fail();
  const int64_t size = wp_vocab_size(h.v);
  std::vector<std::string> result;
  result.reserve(ids.size());
  for (int id : ids) {
    if (id < 0 || id >= size) {  // (the reference tests `>` and lets vector::at throw for id == size)
      std::cerr << "no token " << id << std::endl;
      continue;
    }
    const int32_t flags = wp_vocab_token_flags(h.v, id);
    if (flags & 4) {
      std::cerr << "trying to access malformed token" << std::endl;
      continue;
    }
    std::string word(static_cast<size_t>(wp_vocab_token_utf8(h.v, id, nullptr, 0)), '\0');
    wp_vocab_token_utf8(h.v, id, word.data(), word.size());
    result.push_back((flags & 1) ? word : "##" + word);
  }
  return result;
}

void encodeExternal(const std::string &text_file, const std::string &vocab_file, const std::string &out_file,
                    size_t memory_limit) {
  if (wp_fast_encode_external(text_file.c_str(), vocab_file.c_str(), out_file.c_str(), memory_limit) != WP_OK) fail();
}

}  // namespace word_piece::fast
