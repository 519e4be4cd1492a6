// test_word_piece.cpp — C++ known-answer test of word_piece::linear::* and word_piece::fast::*
// (include/word_piece.hpp) on the GPU path, shaped like the reference's tests/tests.cpp:
// check(text, vocab, expected) asserts BOTH algorithms against the vectors of tests.cpp:137-217 (unknown id
// = -1, none of these vocabularies holds "[UNK]") as tests.cpp:80-88 does, check_split asserts
// linear == fast (tests.cpp:90-97) on the reference's full random-split grid (tests.cpp:219-265, own generator:
// ~30,000 checks, each also against a host-side greedy match), plus the error behaviour of the API.  Built by wordpiece_amd/build.py, run by tests/test_gpu_api.py.
#include <algorithm>
#include <iostream>
#include <random>
#include <set>
#include <unordered_map>
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

// ---- the reference's acceptance grid (tests.cpp:219-272) at its real size ------------------------------------
// A lowercase word of text_len letters is cut at `parts` random borders; the vocabulary is the first piece as a
// word-initial token plus every piece as a "##" continuation (sorted, duplicates merged); the negative form
// drops the first vocabulary line.  tests.cpp asserts linear == fast on every sample; here both device paths
// are also held against a host-side greedy longest-match of the single word (own code, a third opinion).
// Grid: text_len 10..300 step 5 x parts 2..min(text_len, 100) x 3 samples x {positive, negative} (tests.cpp:257-258)
// and text_len 1e5, 5e5, 9e5 with 30,000 parts x 3 samples (tests.cpp:259-265); the 1e7 case runs in pytest.
static int checks_with_unknown = 0;

static std::vector<int> greedy_word(const std::string &w, const std::vector<std::string> &vocab) {
  std::unordered_map<std::string, int> first, cont;  // a later duplicate line would win (fast.cpp:22-36); none here
  size_t longest = 0;
  for (size_t i = 0; i < vocab.size(); i++) {
    const std::string &t = vocab[i];
    const bool is_cont = t.size() > 2 && t[0] == '#' && t[1] == '#';
    const std::string body = is_cont ? t.substr(2) : t;
    (is_cont ? cont : first)[body] = static_cast<int>(i);
    longest = std::max(longest, body.size());
  }
  std::vector<int> ids;
  size_t p = 0;
  while (p < w.size()) {
    const auto &table = p == 0 ? first : cont;
    size_t len = std::min(longest, w.size() - p);
    int id = -1;
    for (; len > 0; len--) {
      auto it = table.find(w.substr(p, len));
      if (it != table.end()) {
        id = it->second;
        break;
      }
    }
    if (id < 0) return {kUnkTokenId};  // the whole word is unknown (linear.cpp:266-272)
    ids.push_back(id);
    p += len;
  }
  return ids;
}

static void check_split(std::mt19937 &rnd, size_t text_len, size_t parts, bool positive) {
  std::string word(text_len, 'a');
  for (char &c : word) c = static_cast<char>('a' + rnd() % 26);
  std::vector<size_t> cut{text_len};
  {
    std::set<size_t> seen{text_len};
    while (seen.size() < parts) {
      const size_t b = 1 + rnd() % (text_len - 1);
      if (seen.insert(b).second) cut.push_back(b);
    }
    std::sort(cut.begin(), cut.end());
  }
  std::set<std::string> lines{word.substr(0, cut[0])};
  for (size_t i = 0, from = 0; i < cut.size(); from = cut[i++]) lines.insert("##" + word.substr(from, cut[i] - from));
  std::vector<std::string> vocab(lines.begin(), lines.end());
  if (!positive) vocab.erase(vocab.begin());
  const std::vector<int> lin = word_piece::linear::encode(word, vocab);
  const std::vector<int> fst = word_piece::fast::encode(word, vocab);
  ++total_checks;
  if (std::find(lin.begin(), lin.end(), kUnkTokenId) != lin.end()) ++checks_with_unknown;
  if (lin != fst) throw std::runtime_error("linear != fast: text_len " + std::to_string(text_len) + ", parts " + std::to_string(parts));
  if (lin != greedy_word(word, vocab)) {
    throw std::runtime_error("linear != host greedy match: text_len " + std::to_string(text_len) + ", parts " + std::to_string(parts));
  }
}

static void testRandomSplit(size_t len_from, size_t len_to, size_t len_step, size_t parts_from, size_t parts_to, bool positive) {
  std::mt19937 rnd(17);
  for (size_t text_len = len_from; text_len <= len_to; text_len += len_step) {
    for (size_t parts = std::min(text_len, parts_from); parts <= std::min(text_len, parts_to); parts++) {
      for (int sample = 0; sample < 3; sample++) check_split(rnd, text_len, parts, positive);
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
  testRandomSplit(10, 300, 5, 2, 100, true);
  testRandomSplit(10, 300, 5, 2, 100, false);
  testRandomSplit(100000, 1000000, 400000, 30000, 30000, true);
  std::cout << "Tests are finished. Passed " << total_checks << " checks, including " << checks_with_unknown
            << " with an unknown word." << std::endl;
}
