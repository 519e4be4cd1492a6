// vocab.h — host-side vocabulary model (utils.cpp:81-146 of the reference) and the derived
// tables the device path needs.
#pragma once
#include <algorithm>
#include <cstdint>
#include <iostream>
#include <string>
#include <utility>
#include <vector>

#include "common.h"

namespace wp {

struct HostToken {
  bool is_prefix = true, is_special = false, is_malformed = false;
  std::vector<uint32_t> word;
};

inline std::vector<uint32_t> host_decode_utf8(const char *s, size_t nbytes, bool warn = true) {
  // utf8.cpp:130-147: invalid bytes are dropped one at a time
  std::vector<uint32_t> out;
  out.reserve(nbytes / 4 + 4);
  bool invalid = false;
  size_t pos = 0;
  const uint8_t *b = reinterpret_cast<const uint8_t *>(s);
  while (pos < nbytes) {
    if ((b[pos] & 0xc0u) == 0x80u) {  // stray continuation byte: utf_length == 0
      invalid = true;
      pos++;
      continue;
    }
    uint32_t cp = decode_one(b + pos, static_cast<int64_t>(nbytes - pos));
    if (cp == kInvalidUnicode) {
      invalid = true;
      pos++;
    } else {
      out.push_back(cp);
      pos += cp < 0x80 ? 1 : cp < 0x800 ? 2 : cp < 0x10000 ? 3 : 4;
    }
  }
  if (invalid && warn) std::cerr << "WARNING Input contains invalid unicode characters." << std::endl;
  return out;
}

struct HostVocab {
  std::vector<HostToken> tokens;
  int32_t unk_id = -1;  // utils.hpp:30-33
  int64_t longest = 1;  // longest_word_vocab, linear.cpp:78-82

  // derived
  std::vector<uint32_t> stream;      // tok0 · 1 · tok1 · 1 · ...  (code points, linear.cpp:93-100)
  std::vector<uint32_t> elig_start;  // offset in `stream` of every eligible token
  std::vector<int32_t> elig_id;      // its vocab line index
  std::vector<uint32_t> elig_info;   // len | class << 30  (class 1 = ## suffix token)
  std::vector<int32_t> tok_len;      // per vocab line
  std::vector<uint32_t> soft;        // sorted spacing chars occurring inside eligible multi-char tokens
  // the vocabulary's share of the alphabet (linear.cpp:83-101: every token's code points and the separator 1 are
  // symbols of S) as words of the used-code-point bitmap: index (code point >> 5) and bits, one entry per word
  std::vector<uint32_t> used_word_idx, used_word_bits;
  // class byte (common.h: space | spacing | soft | punct) of every code point of the BMP, soft flag included:
  // the decode kernel's class lookup is one load instead of ~25 range compares and a search in `soft`
  std::vector<uint8_t> cls_bmp;
  int64_t n_dup_eligible = 0;        // eligible tokens that repeat an earlier (class, word)

  // ---- token trie of the Linear path (trie.h) --------------------------------------------------------------
  // Every distinct prefix of an eligible token's word is a node (both classes in one trie: a token's reach does not
  // depend on its class, linear.cpp:161-189 pushes both kinds on the same kind of stack).  Node ids are PREORDER
  // numbers with children in code point order — the order `elig` is in — so the nodes below a token's node are
  // the id range [elig_node, elig_node + elig_subtree): a suffix of the text has token t as a prefix exactly when
  // the deepest node its symbols reach lies in t's range.  A node with exactly one child starts a unary chain (its
  // only child is the next id); chains are walked 8 symbols per load against the token that runs through them.
  std::vector<uint32_t> lt_chain_len;    // per node: nodes of the unary chain below it (0: leaf or branching node)
  std::vector<uint32_t> lt_chain_off;    // per node: offset in `stream` of the chain's first label
  std::vector<uint32_t> lt_child_begin;  // per node (+1): its children in lt_child_cp / lt_child_node
  std::vector<uint32_t> lt_child_cp, lt_child_node;  // code point (ascending per node) and id of every child
  std::vector<uint32_t> elig_node, elig_subtree;     // per eligible token (in `elig` order): its node and the nodes below it + 1

  // word_piece::fast (fast.cpp:22-36): the two word -> id maps as one trie stored in a hash table.
  // Node 0 / 1 = root of the prefix-class / ##-class tokens; every distinct prefix of an eligible
  // token is a node; trie_key[h] = parent << 32 | code point, trie_child[h] = node (open addressing,
  // kTrieEmpty = free); trie_id[node] = vocab line ending there (the last of equal words, as
  // operator[] assignment gives in fast.cpp:34) or -1.
  static constexpr uint64_t kTrieEmpty = ~0ull;
  std::vector<uint64_t> trie_key;
  std::vector<uint32_t> trie_child;
  std::vector<int32_t> trie_id;
  int64_t fast_max_len = 0;  // longest eligible token (fast.cpp:30)

  static uint32_t trie_hash(uint64_t key) {
    key ^= key >> 33;
    key *= 0xff51afd7ed558ccdull;
    key ^= key >> 29;
    return static_cast<uint32_t>(key);
  }

  void build_trie() {
    size_t total = 2;
    for (const HostToken &t : tokens) {
      if (!t.is_special && !t.is_malformed) total += t.word.size();
    }
    size_t cap = 64;
    while (cap < 2 * total) cap *= 2;
    trie_key.assign(cap, kTrieEmpty);
    trie_child.assign(cap, 0);
    trie_id.assign(2, -1);
    fast_max_len = 0;
    const uint32_t mask = static_cast<uint32_t>(cap - 1);
    for (size_t i = 0; i < tokens.size(); i++) {
      const HostToken &t = tokens[i];
      if (t.is_special || t.is_malformed) continue;  // fast.cpp:27-29
      fast_max_len = std::max<int64_t>(fast_max_len, static_cast<int64_t>(t.word.size()));
      uint32_t node = t.is_prefix ? 0u : 1u;
      for (uint32_t c : t.word) {
        const uint64_t key = (static_cast<uint64_t>(node) << 32) | c;
        uint32_t h = trie_hash(key) & mask;
        while (trie_key[h] != kTrieEmpty && trie_key[h] != key) h = (h + 1) & mask;
        if (trie_key[h] == kTrieEmpty) {
          trie_key[h] = key;
          trie_child[h] = static_cast<uint32_t>(trie_id.size());
          trie_id.push_back(-1);
        }
        node = trie_child[h];
      }
      trie_id[node] = static_cast<int32_t>(i);
    }
  }

  // returns "" or the error message (utils.cpp:99-101)
  std::string build(const std::vector<std::pair<const char *, size_t>> &lines) {
    tokens.clear();
    tokens.reserve(lines.size());
    int32_t id = 0;
    for (auto &ln : lines) {
      if (ln.second == 5 && std::string(ln.first, 5) == "[UNK]") unk_id = id;  // utils.cpp:113-115
      HostToken t;
      t.word = host_decode_utf8(ln.first, ln.second);
      auto &w = t.word;
      if (w.size() >= 2 && w[0] == '#' && w[1] == '#') {  // utils.cpp:139-141
        t.is_prefix = false;
        w.erase(w.begin(), w.begin() + 2);
      } else if (w.size() > 2 && w[0] == '[' && w.back() == ']') {  // utils.cpp:143-146
        t.is_special = true;
      }
      bool all_punct = true;
      for (uint32_t c : w) {
        if (c == kInvalidUnicode) t.is_malformed = true;
        if (!is_punctuation(c) && !is_space(c)) all_punct = false;
      }
      if (w.empty()) return "Vocab word is empty";
      if (t.is_malformed || (all_punct && w.size() > 1)) {
        t.is_malformed = true;
        std::cerr << "Vocab word is malformed: " << std::string(ln.first, ln.second) << std::endl;
      }
      tokens.push_back(std::move(t));
      ++id;
    }
    derive();
    return "";
  }

  bool space_in_token = false;  // an eligible multi-char token holds a space: a match can reach across whitespace,
                                // so the text must not be cut into independent shards (SURVEY 8e caveat, Q13)
  bool low_cp = false;  // some token holds code point 0 or 1 (1 is the separator of S, linear.cpp:92,99):
                        // such vocabularies always go through the reference's S = text . 1 . vocab layout

  void derive() {
    stream.clear();
    elig_start.clear();
    elig_id.clear();
    elig_info.clear();
    tok_len.clear();
    soft.clear();
    longest = 1;
    n_dup_eligible = 0;
    low_cp = false;
    space_in_token = false;
    std::vector<uint32_t> starts(tokens.size());
    std::vector<size_t> elig;
    for (size_t i = 0; i < tokens.size(); i++) {
      const HostToken &t = tokens[i];
      longest = std::max<int64_t>(longest, static_cast<int64_t>(t.word.size()));
      tok_len.push_back(static_cast<int32_t>(t.word.size()));
      starts[i] = static_cast<uint32_t>(stream.size());
      for (uint32_t c : t.word) {
        if (c <= 1) low_cp = true;
      }
      if (!t.is_special && !t.is_malformed) {  // linear.cpp:179
        elig.push_back(i);
        if (t.word.size() > 1) {
          for (uint32_t c : t.word) {
            if (is_spacing_char(c)) soft.push_back(c);
            if (is_space(c)) space_in_token = true;
          }
        }
      }
      stream.insert(stream.end(), t.word.begin(), t.word.end());
      stream.push_back(1);
    }
    // eligible tokens in lexicographic order of their words (a proper prefix first): the order of their
    // suffixes in S, so that the marks of the text-only layout come out sorted by slot
    // (the first three code points packed into one integer decide most comparisons without touching the vectors)
    std::vector<std::pair<uint64_t, size_t>> order(elig.size());
    for (size_t k = 0; k < elig.size(); k++) {
      const std::vector<uint32_t> &w = tokens[elig[k]].word;
      uint64_t key = 0;
      for (size_t j = 0; j < 3; j++) key = (key << 21) | (j < w.size() ? static_cast<uint64_t>(w[j]) + 1 : 0);
      order[k] = {key, elig[k]};
    }
    std::stable_sort(order.begin(), order.end(), [&](const std::pair<uint64_t, size_t> &a, const std::pair<uint64_t, size_t> &b) {
      if (a.first != b.first) return a.first < b.first;
      return tokens[a.second].word < tokens[b.second].word;
    });
    for (size_t k = 0; k < elig.size(); k++) elig[k] = order[k].second;
    // same-class duplicates (they force the true suffix array): equal words are adjacent now
    for (size_t a = 0; a < elig.size();) {
      size_t b = a, np = 0, ns = 0;
      while (b < elig.size() && tokens[elig[b]].word == tokens[elig[a]].word) {
        (tokens[elig[b]].is_prefix ? np : ns)++;
        b++;
      }
      n_dup_eligible += static_cast<int64_t>((np > 1 ? np - 1 : 0) + (ns > 1 ? ns - 1 : 0));
      a = b;
    }
    for (size_t i : elig) {
      const HostToken &t = tokens[i];
      elig_start.push_back(starts[i]);
      elig_id.push_back(static_cast<int32_t>(i));
      elig_info.push_back(static_cast<uint32_t>(t.word.size()) | (t.is_prefix ? 0u : 1u) << 30);
    }
    std::sort(soft.begin(), soft.end());
    soft.erase(std::unique(soft.begin(), soft.end()), soft.end());
    {
      std::vector<uint32_t> cps(stream);
      cps.push_back(1u);  // the separator
      std::sort(cps.begin(), cps.end());
      used_word_idx.clear();
      used_word_bits.clear();
      for (uint32_t c : cps) {
        if (c >= kInvalidUnicode) continue;  // (malformed tokens keep the marker: never part of the alphabet)
        if (used_word_idx.empty() || used_word_idx.back() != (c >> 5)) {
          used_word_idx.push_back(c >> 5);
          used_word_bits.push_back(0u);
        }
        used_word_bits.back() |= 1u << (c & 31u);
      }
    }
    cls_bmp.assign(0x10000, 0);
    for (uint32_t c = 0; c < 0x10000; c++) {
      uint8_t f = 0;
      if (is_space(c)) f |= kClsSpace;
      if (is_punctuation(c)) f |= kClsPunct;
      if (is_spacing_char(c)) f |= kClsSpacing;
      cls_bmp[c] = f;
    }
    for (uint32_t c : soft) {
      if (c < 0x10000) cls_bmp[c] |= kClsSoft;
    }
    build_trie();
    build_token_trie();
  }

  // the eligible tokens come in lexicographic order, so the trie is built along the current path (a stack): the
  // nodes are created in preorder, a node's subtree is closed when the path leaves it
  void build_token_trie() {
    const size_t E = elig_id.size();
    std::vector<uint32_t> depth{0}, nchild{0}, creator{0}, size{0};
    std::vector<uint32_t> edge_parent, edge_cp, edge_node;
    std::vector<uint32_t> path{0};
    elig_node.assign(E, 0);
    elig_subtree.assign(E, 0);
    const std::vector<uint32_t> *prev = nullptr;
    for (size_t k = 0; k < E; k++) {
      const std::vector<uint32_t> &w = tokens[static_cast<size_t>(elig_id[k])].word;
      size_t l = 0;
      if (prev) {
        while (l < prev->size() && l < w.size() && (*prev)[l] == w[l]) l++;
      }
      while (path.size() - 1 > l) {
        const uint32_t x = path.back();
        path.pop_back();
        size[x] = static_cast<uint32_t>(depth.size()) - x;
      }
      for (size_t j = l; j < w.size(); j++) {
        const uint32_t id = static_cast<uint32_t>(depth.size()), parent = path.back();
        depth.push_back(static_cast<uint32_t>(j + 1));
        nchild.push_back(0);
        creator.push_back(static_cast<uint32_t>(k));
        size.push_back(0);
        nchild[parent]++;
        edge_parent.push_back(parent);
        edge_cp.push_back(w[j]);
        edge_node.push_back(id);
        path.push_back(id);
      }
      elig_node[k] = path.back();
      prev = &w;
    }
    const uint32_t N = static_cast<uint32_t>(depth.size());
    while (!path.empty()) {
      size[path.back()] = N - path.back();
      path.pop_back();
    }
    for (size_t k = 0; k < E; k++) elig_subtree[k] = size[elig_node[k]];
    lt_child_begin.assign(static_cast<size_t>(N) + 1, 0);
    for (uint32_t x = 0; x < N; x++) lt_child_begin[x + 1] = lt_child_begin[x] + nchild[x];
    lt_child_cp.assign(edge_node.size(), 0);
    lt_child_node.assign(edge_node.size(), 0);
    std::vector<uint32_t> fill(lt_child_begin.begin(), lt_child_begin.end() - 1);
    for (size_t e = 0; e < edge_node.size(); e++) {  // (creation order: ascending code points under one parent)
      const uint32_t at = fill[edge_parent[e]]++;
      lt_child_cp[at] = edge_cp[e];
      lt_child_node[at] = edge_node[e];
    }
    lt_chain_len.assign(N, 0);
    lt_chain_off.assign(N, 0);
    for (uint32_t x = N; x-- > 0;) {
      if (nchild[x] == 1) lt_chain_len[x] = 1 + lt_chain_len[x + 1];  // (the only child of x is x + 1)
    }
    for (uint32_t x = 0; x < N; x++) {
      if (!lt_chain_len[x]) continue;
      const uint32_t k = creator[x + lt_chain_len[x]];  // the token that runs through the whole chain
      lt_chain_off[x] = elig_start[k] + depth[x];
    }
  }
};

}  // namespace wp
