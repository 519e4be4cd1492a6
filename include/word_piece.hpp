// word_piece.hpp — the reference's public C++ API, unchanged in shape (gleb-kov/wordpiece
// src/word_piece.hpp:10-36), implemented on the MI355X HIP path through the C ABI of wordpiece_amd.h.
// Errors surface as std::runtime_error with the reference's messages ("Vocab word is empty",
// "64bit not implemented"); HIP failures replace "SACA return code: N".
// linear:: is the hot path (suffix array + LCP + scanlines, src/linear.cpp); fast:: is the sibling
// per-word longest-match algorithm (src/fast.cpp), here a trie walk on the GPU.
#pragma once

#include <string>
#include <vector>

namespace word_piece {

namespace linear {

std::vector<int> encode(const std::string &text, const std::vector<std::string> &vocab);

std::vector<int> encode(const std::string &text_file, const std::string &vocab_file);

void encodeExternal(const std::string &text_file,
                    const std::string &vocab_file,
                    const std::string &out_file,
                    size_t memory_limit);

} // namespace linear

namespace fast {

std::vector<int> encode(const std::string &text, const std::vector<std::string> &vocab);

std::vector<int> encode(const std::string &text_file, const std::string &vocab_file);

std::vector<std::string> decode(const std::string vocab_file, const std::vector<int> &ids);

void encodeExternal(const std::string &text_file,
                    const std::string &vocab_file,
                    const std::string &out_file,
                    size_t memory_limit);

} // namespace fast

} // namespace word_piece
