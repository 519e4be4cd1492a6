// word_piece.hpp — the reference's public C++ API for the Linear path, unchanged in shape
// (gleb-kov/wordpiece src/word_piece.hpp:10-21), implemented on the MI355X HIP path through the
// C ABI of wordpiece_amd.h.  Errors surface as std::runtime_error with the reference's messages
// ("Vocab word is empty", "64bit not implemented"); HIP failures replace "SACA return code: N".
// The sibling namespace word_piece::fast of the reference is out of scope.
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

} // namespace word_piece
