// test_word_piece.cpp — C++ known-answer test of word_piece::linear::* and word_piece::fast::*
// (include/word_piece.hpp) on the GPU path, shaped like the reference's tests/tests.cpp:
// check(text, vocab, expected) asserts BOTH algorithms against the vectors of tests.cpp:137-217 (unknown id
// = -1, none of these vocabularies holds "[UNK]") as tests.cpp:80-88 does, check(text, vocab) asserts
// linear == fast (tests.cpp:90-97) on a small random-split grid (tests.cpp:219-246, own generator), plus the
// error behaviour of the API.  Built by wordpiece_amd/build.py, run by tests/test_gpu_api.py.
#include <iostream>
#include <random>
#include <set>
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
  if (word_piece::fast::encode(s, vocab) != expected) throw std::runtime_error("Comparison failed (fast) for \"" + s + "\"");
}

static void check(const std::string &s, const std::vector<std::string> &vocab) {  // tests.cpp:90-97
  ++total_checks;
  if (word_piece::linear::encode(s, vocab) != word_piece::fast::encode(s, vocab)) {
    throw std::runtime_error("linear != fast for \"" + s + "\"");
  }
}

static void testRandomSplit() {  // tests.cpp:219-246 in miniature
  std::mt19937 rnd(17);
  for (size_t text_len : {10u, 40u, 150u, 300u}) {
    for (size_t parts : {2u, 7u, 30u}) {
      for (int positive = 0; positive < 2; positive++) {
        std::string s;
        for (size_t i = 0; i < text_len; i++) s.push_back(static_cast<char>('a' + rnd() % 26));
        std::set<size_t> borders{text_len};
        while (borders.size() < std::min(parts, text_len)) borders.insert(1 + rnd() % (text_len - 1));
        std::set<std::string> res;
        size_t start = 0;
        for (size_t b : borders) {
          if (start == 0) res.insert(s.substr(0, b));
          res.insert("##" + s.substr(start, b - start));
          start = b;
        }
        std::vector<std::string> vocab(res.begin(), res.end());
        if (!positive) vocab.erase(vocab.begin());
        if (!vocab.empty()) check(s, vocab);
      }
    }
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
  std::cout << "running stress tests (split)." << std::endl;
  testRandomSplit();
  std::cout << "Tests are finished. Passed " << total_checks << " checks." << std::endl;
}
