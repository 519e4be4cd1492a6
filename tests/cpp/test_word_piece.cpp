// test_word_piece.cpp — C++ known-answer test of word_piece::linear::* (include/word_piece.hpp) on
// the GPU path, shaped like the reference's tests/tests.cpp: check(text, vocab, expected) for the
// vectors of tests.cpp:137-217 (unknown id = -1, none of these vocabularies holds "[UNK]"), plus
// the error behaviour of the API.  The differential half of the reference's test (Linear == Fast)
// has no counterpart here (fast:: is out of scope); the Python parity suite diffs against the
// CPU oracle instead.  Built by wordpiece_amd/build.py, run by tests/test_gpu_api.py.
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/word_piece.hpp"

static constexpr int kUnkTokenId = -1;
static int total_checks = 0;

static void check(const std::string &s, const std::vector<std::string> &vocab, const std::vector<int> &expected) {
  ++total_checks;
  const std::vector<int> got = word_piece::linear::encode(s, vocab);
  if (got != expected) {
    std::cout << "Comparison failed for \"" << s << "\": got";
    for (int x : got) std::cout << ' ' << x;
    std::cout << ", expected";
    for (int x : expected) std::cout << ' ' << x;
    std::cout << std::endl;
    throw std::runtime_error("Comparison failed");
  }
}

static void testSimple() {
  check("abcdef", {"bcde", "ac", "def", "bc", "bcdef", "a"}, {kUnkTokenId});
  check("abcdef", {"bcde", "ac", "def", "bc", "##bcdef", "a"}, {5, 4});
  check("   aaaa  ", {"aa", "##aa"}, {0, 1});
  check("   aaaa  ", {"aa"}, {kUnkTokenId});
  check("aaaa", {"aaaa"}, {0});
  check("aaaa", {"##aaaa"}, {kUnkTokenId});
  check("aaaa", {"aaaa", "##aaaa", "##aaa", "##aa", "##a"}, {0});
  check("aaaa", {"##aaa", "aaaa", "##aa", "##a"}, {1});
  check("aaaa", {"aaa", "##aa", "##a", "##aaa"}, {0, 2});
  check("aaaa", {"aa", "a", "##aa"}, {0, 2});
  check("aaaa", {"aa", "a", "##aaa"}, {kUnkTokenId});
  check("aaaa", {"aa", "##a"}, {0, 1, 1});
  check("abcdef", {"##def", "abc"}, {1, 0});
  check("abcdef", {"##bcde", "##ac", "##def", "##bc", "##bcdef", "a", "##a"}, {5, 4});
  check("abcdef", {"##bcdd", "##ac", "##def", "##bc", "##bcdff", "a"}, {5, 3, 2});
  check("djzhoyuhmcij", {"d", "##j", "##z", "##h", "##o", "##y", "##u", "##m", "##c", "##i", "##d"},
        {0, 1, 2, 3, 4, 5, 6, 3, 7, 8, 9, 1});
}

static void testPunctuation() {
  check("self-made", {"self", "made", "-", "##-", "##made"}, {0, 2, 1});
  check("self, made", {"self", "made", ",", "##,", "##made"}, {0, 2, 1});
  check("self  , made", {"self", "made", ",", "##,", "##made"}, {0, 2, 1});
}

static void testNonSplitted() {
  check("abc", {"a", "abd"}, {kUnkTokenId});
  check("abc a abc abd", {"a", "abd"}, {kUnkTokenId, 0, kUnkTokenId, 1});
  check("abcdef", {"bcde", "ac", "def", "bc", "bcdef", "##a", "##b", "##c", "##d"}, {kUnkTokenId});
}

static void testMaxMatch() {
  check("abcdef", {"a", "##bcdef", "ab", "##c", "##d", "##e", "##f"}, {2, 3, 4, 5, 6});
  check("abcdef abc abcd", {"abcd", "def", "abc"}, {kUnkTokenId, 2, 0});
}

static void testUtf8() {
  check("привет мир", {"привет", "мир"}, {0, 1});
  check("привет мир", {"при", "##вет", "мир"}, {0, 1, 2});
  check("токенизация это круто", {"ток", "крут", "это", "##за", "##ция", "ция"}, {kUnkTokenId, 2, kUnkTokenId});
  check("токенизация это круто", {"ток", "крут", "это", "##за", "##ени", "##о", "##ция", "ция"},
        {0, 4, 3, 6, 2, 1, 5});
}

static void testErrors() {
  ++total_checks;
  bool thrown = false;
  try {
    word_piece::linear::encode("a", std::vector<std::string>{"a", "##"});
  } catch (const std::runtime_error &e) {
    thrown = std::string(e.what()) == "Vocab word is empty";  // utils.cpp:99-101
  }
  if (!thrown) throw std::runtime_error("expected \"Vocab word is empty\"");
  check("", {"a"}, {});  // linear.cpp:323-325
}

int main() {
  std::cout << "running small unit tests." << std::endl;
  testSimple();
  testNonSplitted();
  testPunctuation();
  testMaxMatch();
  testUtf8();
  testErrors();
  std::cout << "Tests are finished. Passed " << total_checks << " checks." << std::endl;
}
