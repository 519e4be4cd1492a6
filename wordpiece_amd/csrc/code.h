// code.h — order-preserving variable-length symbol code for the round-0 sort keys.
//
// (Alphabets of more than 255 symbols — mixed scripts — use a *split* code: the Garsia–Wachs code
// is built over the high part of the dense symbol id (symbol >> lo_bits, at most 256 values) and the
// low lo_bits follow verbatim.  Dense ids are order preserving, frequent scripts occupy few high
// values, so Latin / Cyrillic symbols cost 7-8 bits and CJK 12-13 instead of 13 for everything; the
// code stays prefix free and order preserving.)
//
// Round 0 of the suffix sort packs the beginning of every suffix into one 63-bit key.  With a
// fixed b bits per symbol that is floor(63/b) symbols (9 for English text).  An *alphabetic*
// prefix code (Garsia–Wachs: optimal among codes whose codeword order equals the symbol order)
// spends ~entropy+0.3 bits per symbol instead, so the same 63 bits hold ~14 symbols of English
// text; after the same number of radix passes far fewer suffixes are still tied (22 % instead of
// 62 % on a 25 MB sample).  Comparing two keys as integers still compares the suffixes
// correctly: the codes are prefix free and order preserving, so the first differing bit lies in
// the first differing symbol.  Keys are cut after 63 bits wherever that falls; a tied group
// shares the complete codewords inside those bits (its depth, >= 1 symbol).
#pragma once
#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <vector>

namespace wp {

// 32 bits = 4 radix passes over 8-byte (key u32, index u32) records.  With the round-0 pruning (prune.h)
// ties cost next to nothing — only the groups that carry a long token's key go on — so fewer key bits are
// a plain saving of passes until the needed groups grow.  Measured on the 100 MB bench shard with 8-byte
// keys in 12-byte records, ms per step / entries in round 1: 63 bits 11.3 / 12 k, 56: 10.6 / 60 k,
// 48: 10.1 / 0.25 M, 40: 9.5 / 0.8 M, 32: 9.6 / 3.0 M, 24: 9.9 / 5.6 M; keys of up to 32 bits are stored
// as uint32 (Key0), which makes a pass move 16 instead of 24 bytes per element.
#ifndef WP_KEY_BITS
#define WP_KEY_BITS 32
#endif
constexpr int kKeyBits = WP_KEY_BITS;  // bits of the codeword stream kept in a round-0 key (<= 63)
using Key0 = std::conditional_t<(kKeyBits <= 32), uint32_t, uint64_t>;  // storage type of the round-0 keys
constexpr int kMaxCodeLen = 12;  // decode table: 2^12 entries (first codeword length of a 12-bit window)

struct SymbolCode {
  int uniform_bits = 0;            // > 0: every symbol takes this many bits (fixed-width mode)
  int lo_bits = 0;                 // split code (alphabets > 255): the tables code symbol >> lo_bits,
                                   // the low lo_bits of the symbol follow the codeword verbatim
  std::vector<uint16_t> cw;        // codeword of dense symbol s (0 = past-the-end padding), right aligned
  std::vector<uint8_t> len;        // its length in bits
  std::vector<uint8_t> first_len;  // [4096]: length of the first codeword of a 12-bit window
  std::vector<uint16_t> bmask;     // [4096]: bit j set = a codeword of the window ends after j+1 bits
  double avg_bits = 0;
};

// optimal alphabetic code lengths for weights w (Garsia–Wachs), O(n^2) worst case, n <= 256
inline std::vector<int> garsia_wachs(const std::vector<double> &w) {
  const int n = static_cast<int>(w.size());
  if (n == 1) return {1};
  struct Node {
    double w;
    int left, right, leaf;
  };
  std::vector<Node> nodes;
  std::vector<int> seq;  // indices into nodes; sentinels are -1
  for (int i = 0; i < n; i++) nodes.push_back({w[i], -1, -1, i});
  const double INF = 1e300;
  auto weight = [&](int id) { return id < 0 ? INF : nodes[id].w; };
  seq.push_back(-1);
  for (int i = 0; i < n; i++) seq.push_back(i);
  seq.push_back(-1);
  while (seq.size() > 3) {
    size_t i = 1;
    while (!(weight(seq[i - 1]) <= weight(seq[i + 1]))) i++;
    nodes.push_back({weight(seq[i - 1]) + weight(seq[i]), seq[i - 1], seq[i], -1});
    const int id = static_cast<int>(nodes.size()) - 1;
    seq.erase(seq.begin() + static_cast<long>(i) - 1, seq.begin() + static_cast<long>(i) + 1);
    long j = static_cast<long>(i) - 2;
    while (weight(seq[static_cast<size_t>(j)]) < nodes[id].w) j--;
    seq.insert(seq.begin() + j + 1, id);
  }
  std::vector<int> depth(n, 0);
  std::vector<std::pair<int, int>> st{{seq[1], 0}};
  while (!st.empty()) {
    auto [id, d] = st.back();
    st.pop_back();
    if (nodes[id].leaf >= 0) {
      depth[nodes[id].leaf] = d;
    } else {
      st.push_back({nodes[id].left, d + 1});
      st.push_back({nodes[id].right, d + 1});
    }
  }
  return depth;
}

// freq[s] for s = 0..sigma (s = 0 is the padding symbol).  Falls back to fixed width when the
// alphabet does not fit the 8-bit symbol path or the length limit cannot be met.
inline SymbolCode build_symbol_code(const std::vector<uint64_t> &freq, int fixed_bits, bool allow_variable) {
  SymbolCode c;
  const int n = static_cast<int>(freq.size());
  auto fixed = [&] {
    c.uniform_bits = fixed_bits;
    c.cw.clear();
    c.len.clear();
    c.first_len.clear();
    c.bmask.clear();
    c.avg_bits = fixed_bits;
    return c;
  };
  if (!allow_variable || n < 2 || n > 256) return fixed();
  double total = 0;
  for (uint64_t f : freq) total += static_cast<double>(f);
  if (total <= 0) return fixed();
  std::vector<int> L;
  for (double floor_div = 512; floor_div >= 32; floor_div /= 2) {
    std::vector<double> w(n);
    for (int i = 0; i < n; i++) w[i] = std::max<double>(std::max<double>(static_cast<double>(freq[i]), 1.0), total / floor_div);
    L = garsia_wachs(w);
    if (*std::max_element(L.begin(), L.end()) <= kMaxCodeLen) break;
    L.clear();
  }
  if (L.empty()) return fixed();
  // codewords in symbol order from the depth sequence of the alphabetic tree
  c.cw.resize(n);
  c.len.resize(n);
  uint32_t code = 0;
  int prev = L[0];
  c.cw[0] = 0;
  c.len[0] = static_cast<uint8_t>(L[0]);
  for (int i = 1; i < n; i++) {
    code += 1;
    if (L[i] > prev) code <<= (L[i] - prev); else code >>= (prev - L[i]);
    prev = L[i];
    c.cw[i] = static_cast<uint16_t>(code);
    c.len[i] = static_cast<uint8_t>(L[i]);
  }
  // sanity: strictly increasing when left aligned (order preserving and prefix free)
  for (int i = 1; i < n; i++) {
    const uint32_t a = static_cast<uint32_t>(c.cw[i - 1]) << (16 - c.len[i - 1]);
    const uint32_t b = static_cast<uint32_t>(c.cw[i]) << (16 - c.len[i]);
    if (!(a < b)) return fixed();
  }
  c.first_len.assign(1 << kMaxCodeLen, 0);
  for (int i = 0; i < n; i++) {
    const uint32_t lo = static_cast<uint32_t>(c.cw[i]) << (kMaxCodeLen - c.len[i]);
    const uint32_t cnt = 1u << (kMaxCodeLen - c.len[i]);
    for (uint32_t k = 0; k < cnt; k++) c.first_len[lo + k] = c.len[i];
  }
  for (uint8_t v : c.first_len) {
    if (v == 0) return fixed();  // not a complete code (cannot happen for a full binary tree)
  }
  c.bmask.assign(1 << kMaxCodeLen, 0);
  for (uint32_t w = 0; w < (1u << kMaxCodeLen); w++) {
    int pos = 0;
    uint32_t bm = 0;
    while (pos < kMaxCodeLen) {
      const int l = c.first_len[(w << pos) & ((1u << kMaxCodeLen) - 1)];
      if (pos + l > kMaxCodeLen) break;
      pos += l;
      bm |= 1u << (pos - 1);
    }
    c.bmask[w] = static_cast<uint16_t>(bm);
  }
  double bits = 0;
  for (int i = 0; i < n; i++) bits += static_cast<double>(freq[i]) * L[i];
  c.avg_bits = bits / total;
  return c;
}

}  // namespace wp
