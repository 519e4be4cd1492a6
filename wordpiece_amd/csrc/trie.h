// trie.h — which slots of a needed group a long token reaches, found along the token trie instead of by
// prefix-doubling rounds (default, text-only layout).
//
// After round 0 a needed group G (prune.h) holds the suffixes that share the round-0 key of some token whose code
// stream is longer than the key.  The scanlines (linear.cpp:161-213) only ask, per suffix, WHICH tokens are its
// prefixes, and the tokens that are prefixes of a string are the ancestors of the deepest node the string reaches in
// the trie of all eligible tokens (vocab.h: nodes numbered in preorder).  So every member of a needed group walks
// the trie once — unary chains 8 symbols per load against the token that runs through them — and its end node is
// its sort key inside the group: one segmented sort (the machinery of a doubling round, local_sort.h /
// suffix_array.h, keyed by the node instead of rank[i + depth]) puts the members of G into an order in which every
// token's reach is the contiguous run of the nodes below its own, [node, node + subtree), found by two integer
// binary searches.  The depth of the vocabulary no longer shows in the number of rounds: a 512-symbol token took
// ~80 doubling rounds over its group (the second keys pointed into groups that had retired in round 0, each adding
// ~7 symbols), and takes one walk of 64 loads here.  The doubling rounds remain for the true suffix array (full
// depth, duplicate lines) and the reference's S = text . 1 . vocab layout.
#pragma once
#include "decode.h"
#include "primitives.h"

namespace wp {

struct TokenTrie {
  const uint32_t *chain_len;    // per node: nodes of the unary chain below it
  const uint32_t *chain_off;    // per node: offset of the chain's labels in the vocabulary stream
  const uint32_t *child_begin;  // per node (+1): its children
  const uint32_t *child_node;   // per child: node id
  const uint32_t *child_sym;    // per child: dense symbol of its label (this encode's alphabet), ascending per node
};

// dense symbols of the vocabulary stream and of the child labels under this encode's alphabet
template <typename SymT>
__global__ __launch_bounds__(kBlock) void trie_map_symbols_kernel(const uint32_t *__restrict__ stream_cps, size_t n_stream,
                                                                  const uint32_t *__restrict__ child_cps, size_t n_child,
                                                                  const uint32_t *__restrict__ lut_excl,
                                                                  SymT *__restrict__ vsym, uint32_t *__restrict__ child_sym) {
  const size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n_stream) vsym[i] = static_cast<SymT>(lut_excl[stream_cps[i]] + 1u);
  if (i < n_child) child_sym[i] = lut_excl[child_cps[i]] + 1u;
}

// symbols text[pos ..] and label[0 ..] agree on (at most len)
template <typename SymT>
__device__ __forceinline__ uint32_t trie_chain_match(const SymT *__restrict__ sym, size_t n, size_t pos,
                                                     const SymT *__restrict__ label, uint32_t len) {
  if (pos >= n) return 0u;
  const uint32_t lim = static_cast<uint32_t>(min(static_cast<size_t>(len), n - pos));
  uint32_t t = 0;
  if (sizeof(SymT) == 1) {  // 8 symbols per (unaligned) 64-bit load; both arrays are padded by 16 bytes
    const uint8_t *pa = reinterpret_cast<const uint8_t *>(sym) + pos, *pb = reinterpret_cast<const uint8_t *>(label);
    while (t < lim) {
      uint64_t wa, wb;
      __builtin_memcpy(&wa, pa + t, 8);
      __builtin_memcpy(&wb, pb + t, 8);
      const uint64_t x = wa ^ wb;
      if (x) {
        t += static_cast<uint32_t>(__ffsll(static_cast<long long>(x)) - 1) >> 3;
        break;
      }
      t += 8;
    }
    return min(t, lim);
  }
  while (t < lim && sym[pos + t] == label[t]) t++;
  return t;
}

// from `node` (whose path the suffix is known to follow up to text position pos) to the deepest node the suffix
// reaches; with limit != 0 the walk stops after `limit` more symbols (left_out = how many of them were not matched)
template <typename SymT>
__device__ __forceinline__ uint32_t trie_descend(const TokenTrie &t, const SymT *__restrict__ sym, size_t n,
                                                 const SymT *__restrict__ vsym, uint32_t node, size_t pos, uint32_t limit,
                                                 uint32_t &left_out) {
  uint32_t left = limit ? limit : 0xffffffffu;
  for (;;) {
    const uint32_t len = t.chain_len[node];
    if (len) {
      const uint32_t want = min(len, left);
      const uint32_t l = trie_chain_match(sym, n, pos, vsym + t.chain_off[node], want);
      node += l;  // (the nodes of a chain are consecutive in preorder)
      pos += l;
      left -= l;
      if (l < len) break;  // (mismatch, end of the text, or the limit)
    }
    if (left == 0) break;
    uint32_t lo = t.child_begin[node], hi = t.child_begin[node + 1];
    if (lo == hi || pos >= n) break;
    const uint32_t s = static_cast<uint32_t>(sym[pos]), end = hi;
    while (lo < hi) {
      const uint32_t md = (lo + hi) >> 1;
      if (t.child_sym[md] < s) lo = md + 1; else hi = md;
    }
    if (lo == end || t.child_sym[lo] != s) break;
    node = t.child_node[lo];
    pos++;
    left--;
  }
  left_out = limit ? left : 0u;
  return node;
}

// one thread per needed group: the node its members share — they agree on the first gdepth[g] symbols (the whole
// codewords of the round-0 key), so that stretch of the path is walked once per group instead of once per entry
template <typename SymT>
__global__ __launch_bounds__(kBlock) void trie_group_start_kernel(const uint32_t *__restrict__ sorted_vals,
                                                                  const uint32_t *__restrict__ gfirst,
                                                                  const uint32_t *__restrict__ gdepth,
                                                                  const uint32_t *__restrict__ sizes_dev,
                                                                  const SymT *__restrict__ sym, size_t n,
                                                                  const SymT *__restrict__ vsym, TokenTrie t,
                                                                  uint32_t *__restrict__ gnode, uint32_t *__restrict__ gdone) {
  const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  if (g >= sizes_dev[1]) return;
  const uint32_t d = gdepth[g];
  uint32_t left = 0, node = 0;
  if (d) node = trie_descend(t, sym, n, vsym, 0u, static_cast<size_t>(sorted_vals[gfirst[g]]), d, left);
  gnode[g] = node;
  gdone[g] = d - left;  // symbols of the shared stretch that are behind `node` (all of them, when a token carries the key)
}

// one lane per entry of the needed list: the deepest trie node the suffix at vals[p] reaches
// (sizes_dev[0] = list length, read on the device: the launch does not wait for the host to know it)
template <typename SymT>
__global__ __launch_bounds__(kBlock) void trie_walk_kernel(const uint32_t *__restrict__ vals,
                                                           const uint32_t *__restrict__ gid,
                                                           const uint32_t *__restrict__ gnode,
                                                           const uint32_t *__restrict__ gdone,
                                                           const uint32_t *__restrict__ sizes_dev,
                                                           const SymT *__restrict__ sym, size_t n,
                                                           const SymT *__restrict__ vsym, TokenTrie t,
                                                           uint32_t *__restrict__ node_out) {
  const size_t m = sizes_dev[0];
  for (size_t p = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; p < m; p += static_cast<size_t>(gridDim.x) * kBlock) {
    const uint32_t g = gid[p];
    uint32_t unused;
    node_out[p] = trie_descend(t, sym, n, vsym, gnode[g], static_cast<size_t>(vals[p]) + gdone[g], 0u, unused);
  }
}


// After the segmented sort: node_of_slot[] holds, for the slots of the needed groups, the end node of the suffix in
// that slot, ascending inside every group.  The reach of long token m inside its group [rng_lo, rng_hi) is the run of
// nodes [tok_node, tok_node + tok_subtree).
__global__ __launch_bounds__(kBlock) void trie_token_range_kernel(const uint32_t *__restrict__ node_of_slot,
                                                                  const uint32_t *__restrict__ tok_node,
                                                                  const uint32_t *__restrict__ tok_subtree, int M,
                                                                  uint32_t *__restrict__ rng_lo, uint32_t *__restrict__ rng_hi,
                                                                  const uint8_t *__restrict__ rng_long) {
  const int m = blockIdx.x * kBlock + threadIdx.x;
  if (m >= M || !rng_long[m]) return;
  const uint32_t glo = rng_lo[m], ghi = rng_hi[m];
  const uint32_t a = tok_node[m], b = a + tok_subtree[m];
  uint32_t lo = glo, hi = ghi;
  while (lo < hi) {
    const uint32_t md = lo + ((hi - lo) >> 1);
    if (node_of_slot[md] < a) lo = md + 1; else hi = md;
  }
  const uint32_t first = lo;
  hi = ghi;
  while (lo < hi) {
    const uint32_t md = lo + ((hi - lo) >> 1);
    if (node_of_slot[md] < b) lo = md + 1; else hi = md;
  }
  rng_lo[m] = first;
  rng_hi[m] = lo;
}

}  // namespace wp
