// runner.cpp — CLI with the reference runner's positional arguments (tests/runner.cpp:13-65):
//   runner <mode> <text_file> <vocab_file> [n_threads] [out_file] [memory_limit_mb]
// modes: fast, linear, fast-external, linear-external.  n_threads is accepted and ignored (the work runs on the GPU).
#include <fstream>
#include <iostream>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/word_piece.hpp"

int main(int argc, char *argv[]) {
  if (argc < 4 || argc > 7) {
    throw std::runtime_error("Usage: ./runner <mode> <text_file> <vocab_file> [n_threads] "
                             "[out_file] [memory_limit_mb]. Modes: fast, linear, fast-external, linear-external.");
  }
  const std::string mode = argv[1], text_file = argv[2], vocab_file = argv[3];
  const std::optional<std::string> out_file = argc >= 6 ? std::optional<std::string>(argv[5]) : std::nullopt;
  std::optional<size_t> memory_limit = argc >= 7 ? std::optional<size_t>(std::stoull(argv[6])) : std::nullopt;
  if (memory_limit.has_value()) {
    if (*memory_limit < 50) throw std::runtime_error("memory_limit cannot be less than 50Mb");
    *memory_limit *= 1'000'000;
  }
  if (mode == "linear" || mode == "fast") {
    std::vector<int> ids = mode == "linear" ? word_piece::linear::encode(text_file, vocab_file)
                                            : word_piece::fast::encode(text_file, vocab_file);
    std::cout << "Total ids " << ids.size() << std::endl;
    if (out_file) {  // utils.cpp:30-35 writeToFile
      std::ofstream fout(*out_file);
      for (int id : ids) fout << id << ' ';
    }
  } else if (mode == "linear-external" || mode == "fast-external") {
    if (!memory_limit.has_value()) throw std::runtime_error("For external mode provide out_file and memory_limit");
    if (mode == "linear-external") {
      word_piece::linear::encodeExternal(text_file, vocab_file, out_file.value(), memory_limit.value());
    } else {
      word_piece::fast::encodeExternal(text_file, vocab_file, out_file.value(), memory_limit.value());
    }
  } else {
    throw std::runtime_error("Unknown mode");
  }
}
