// fast.h — word_piece::fast::encode on the GPU (fast.cpp:19-150 of the reference): per word, the longest
// vocab token that is a prefix of the rest of the word, found by walking a trie of the eligible tokens
// (vocab.h: the reference's two unordered_maps of whole words, fast.cpp:22-36, as one hash-stored trie,
// so that a lookup stops at the first symbol no token continues with instead of hashing every prefix
// of the segment from the longest down, fast.cpp:67-78).
//
// Parallelism as in walk.h: the reference's worker is memoryless whenever it stands on a word-prefix
// position with tokens_since_prefix == 0, and here a position is *certain* to be such a stop when it is
// not a space and it is the text start, a spacing char itself, or preceded by a space or a punctuation
// char: a segment never reaches across a spacing char (fast.cpp:57-60), a punctuation char is a
// segment of its own (fast.cpp:56) and spaces are skipped (fast.cpp:94-96).  The one exception is a
// non-spacing char behind a CJK char: the CJK char's segment continues over it (is_spacing_char but not
// is_punctuation), so that position belongs to the CJK char's anchor.  One lane walks from each anchor
// to the next; words longer than kMaxAnchorGap go through the pointer-doubling kernels of walk.h.
#pragma once
#include "primitives.h"
#include "walk.h"

namespace wp {

struct TrieView {
  const unsigned long long *key;  // parent << 32 | code point, ~0 = free
  const uint32_t *child;
  const int32_t *id;  // per node: vocab line or -1
  uint32_t mask;
};

__device__ __forceinline__ uint32_t trie_hash_dev(uint64_t key) {  // == HostVocab::trie_hash
  key ^= key >> 33;
  key *= 0xff51afd7ed558ccdull;
  key ^= key >> 29;
  return static_cast<uint32_t>(key);
}

// child of `node` by code point c, or 0xffffffff
__device__ __forceinline__ uint32_t trie_step(const TrieView &t, uint32_t node, uint32_t c) {
  const uint64_t key = (static_cast<uint64_t>(node) << 32) | c;
  uint32_t h = trie_hash_dev(key) & t.mask;
  for (;;) {
    const uint64_t k = t.key[h];
    if (k == key) return t.child[h];
    if (k == ~0ull) return 0xffffffffu;
    h = (h + 1) & t.mask;
  }
}

struct FastArgs {
  const uint32_t *cps;  // code points of the text
  const uint8_t *cls;   // class bytes (decode.h)
  size_t n_text;
  TrieView trie;
  const int32_t *tok_len;
  int32_t unk_id;
  uint32_t max_len;  // min(longest eligible token, text length): fast.cpp:30,37
  int32_t *emit;
};

__device__ __forceinline__ bool f_word_prefix(const FastArgs &a, size_t p) {  // fast.cpp:39-42
  return p == 0 || (a.cls[p] & kClsSpacing) || (a.cls[p - 1] & kClsSpacing);
}
__device__ __forceinline__ bool f_anchor(const FastArgs &a, size_t p) {
  const uint8_t c = a.cls[p];
  if (c & kClsSpace) return false;
  return p == 0 || (c & kClsSpacing) || (a.cls[p - 1] & (kClsSpace | kClsPunct));
}

// length of the reference's segment at p (fast.cpp:55-61), computed only where it is needed (a failed
// lookup advances by it)
__device__ __forceinline__ size_t f_word_len(const FastArgs &a, size_t p) {
  size_t wl = 1;
  if (!(a.cls[p] & kClsPunct)) {
    const size_t lim = min(static_cast<size_t>(a.max_len), a.n_text - p);
    while (wl < lim && !(a.cls[p + wl] & kClsSpacing)) ++wl;
  }
  return wl;
}

// longest token of p's class that is a prefix of the segment at p: id or -1 (fast.cpp:63-78)
__device__ __forceinline__ int32_t f_match(const FastArgs &a, size_t p) {
  uint32_t node = f_word_prefix(a, p) ? 0u : 1u;
  const size_t lim = (a.cls[p] & kClsPunct) ? 1 : min(static_cast<size_t>(a.max_len), a.n_text - p);
  int32_t best = -1;
  for (size_t k = 0; k < lim; k++) {
    if (k > 0 && (a.cls[p + k] & kClsSpacing)) break;  // the segment ends in front of the next spacing char
    node = trie_step(a.trie, node, a.cps[p + k]);
    if (node == 0xffffffffu) break;
    const int32_t id = a.trie.id[node];
    if (id >= 0) best = id;
  }
  return best;
}

// one token (or the [UNK] of its word) of a lane's walk; true when the lane's stretch of text ends
template <typename Out>
__device__ __forceinline__ bool fast_walk_step(const FastArgs &a, WalkState &s, Out &o) {
  const size_t end = a.n_text;
  size_t p = s.p;
  const int32_t id = f_match(a, p);
  if (id != -1) {
    o.push(p, id);
    p += static_cast<size_t>(a.tok_len[id]);
    if (p < end && f_word_prefix(a, p)) {  // fast.cpp:90-92
      s.since = p;
      o.word_start();
    }
  } else {  // fast.cpp:80-89: roll the word's tokens back, [UNK], skip the rest of the word
    o.rollback(s.since, p);
    o.push(p, a.unk_id);
    p += f_word_len(a, p);
    while (p < end && !f_word_prefix(a, p)) ++p;
    s.since = p;
    o.word_start();
  }
  s.p = p;
  // behind a space the next non-space position is an anchor of its own (fast.cpp:94-96 skips the run)
  return p >= end || (a.cls[p] & kClsSpace) || f_anchor(a, p);
}

template <typename Out>
__device__ inline void fast_walk_from(const FastArgs &a, size_t p, Out &o) {
  WalkState s{p, p};
  o.word_start();
  while (s.p < a.n_text) {
    if (fast_walk_step(a, s, o)) return;
  }
}

struct FastStep {
  using State = WalkState;
  __device__ static __forceinline__ bool step(const FastArgs &a, WalkState &s, StagedOut &o) { return fast_walk_step(a, s, o); }
};

// 16 class bytes per lane -> anchors as a 16-bit mask (same layout as anchor_mask16 of walk.h)
__device__ __forceinline__ uint32_t fast_anchor_mask16(const uint8_t *__restrict__ cls, size_t n, size_t i) {
  if (i >= n) return 0u;
  uint32_t m = 0;
  uint8_t prev = i > 0 ? cls[i - 1] : static_cast<uint8_t>(kClsSpace);  // position 0 counts as preceded by a space
  uint32_t w[4] = {0, 0, 0, 0};
  if (i + 16 <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(cls + i);
    w[0] = v.x;
    w[1] = v.y;
    w[2] = v.z;
    w[3] = v.w;
  } else {
    for (int j = 0; j < 16 && i + j < n; j++) w[j >> 2] |= static_cast<uint32_t>(cls[i + j]) << (8 * (j & 3));
  }
#pragma unroll
  for (int j = 0; j < 16; j++) {
    const uint8_t c = static_cast<uint8_t>(w[j >> 2] >> (8 * (j & 3)));
    const bool anchor = i + j < n && !(c & kClsSpace) && ((c & kClsSpacing) || (prev & (kClsSpace | kClsPunct)));
    m |= anchor ? (1u << j) : 0u;
    prev = c;
  }
  return m;
}

__global__ __launch_bounds__(kBlock) void fast_anchor_count_kernel(const uint8_t *__restrict__ cls, size_t n,
                                                                   uint32_t *__restrict__ tile_counts) {
  __shared__ uint32_t sm[8];
  const size_t i = static_cast<size_t>(blockIdx.x) * kAnchorTile + static_cast<size_t>(threadIdx.x) * kAnchorBytes;
  const uint32_t c = __popc(fast_anchor_mask16(cls, n, i));
  uint32_t tot;
  (void)block_excl_sum(c, sm, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void fast_anchor_write_kernel(const uint8_t *__restrict__ cls, size_t n,
                                                                   const uint32_t *__restrict__ tile_prefix,
                                                                   uint32_t *__restrict__ anchors) {
  __shared__ uint32_t sm[8];
  const size_t i = static_cast<size_t>(blockIdx.x) * kAnchorTile + static_cast<size_t>(threadIdx.x) * kAnchorBytes;
  uint32_t m = fast_anchor_mask16(cls, n, i);
  uint32_t tot;
  uint32_t o = tile_prefix[blockIdx.x] + block_excl_sum(static_cast<uint32_t>(__popc(m)), sm, tot);
  while (m) {
    const int j = __ffs(static_cast<int>(m)) - 1;
    m &= m - 1;
    anchors[o++] = static_cast<uint32_t>(i + j);
  }
}

// largest distance from an anchor to the first space behind it (or to the next anchor)
__global__ __launch_bounds__(kBlock) void fast_anchor_gap_kernel(const uint32_t *__restrict__ anchors,
                                                                 const uint32_t *__restrict__ n_anchors_dev,
                                                                 size_t n_text, const uint8_t *__restrict__ cls,
                                                                 uint32_t *__restrict__ max_gap) {
  __shared__ int32_t sm[8];
  const size_t na = *n_anchors_dev;
  uint32_t g = 0;
  for (size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; k < na;
       k += static_cast<size_t>(gridDim.x) * kBlock) {
    const uint32_t lo = anchors[k];
    const uint32_t hi = k + 1 < na ? anchors[k + 1] : static_cast<uint32_t>(n_text);
    uint32_t d = hi - lo;
    if (d > kMaxAnchorGap) {  // a lane stops at the first space: a blank run behind a word is nobody's walk
      uint32_t q = lo;
      const uint32_t stop = min(hi, lo + static_cast<uint32_t>(kMaxAnchorGap) + 1u);
      while (q < stop && !(cls[q] & kClsSpace)) q++;
      d = q - lo;
    }
    g = max(g, d);
  }
  const int32_t m = -block_reduce_min(-static_cast<int32_t>(g), sm);
  if (threadIdx.x == 0 && m > 0) atomicMax(max_gap, static_cast<uint32_t>(m));
}

// Long words (walk.h, "long words"): ranges [anchor, next anchor) with more than kMaxAnchorGap
// non-space positions and no word-prefix position inside (a CJK char followed by a long run of
// non-spacing chars stays with its lane).  flags[k] = 1 for the anchors taken out of the lane walk.
__global__ __launch_bounds__(kBlock) void fast_long_word_collect_kernel(uint32_t *__restrict__ anchors,
                                                                        const uint32_t *__restrict__ n_anchors_dev,
                                                                        size_t n_text, const uint8_t *__restrict__ cls,
                                                                        LongWord *__restrict__ list, uint32_t cap,
                                                                        uint32_t *__restrict__ count) {
  const size_t na = *n_anchors_dev;
  for (size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; k < na;
       k += static_cast<size_t>(gridDim.x) * kBlock) {
    const uint32_t lo = anchors[k] & ~kAnchorSkip;
    const uint32_t hi = k + 1 < na ? (anchors[k + 1] & ~kAnchorSkip) : static_cast<uint32_t>(n_text);
    if (hi - lo <= kMaxAnchorGap) continue;
    uint32_t q = lo;
    const uint32_t stop = min(hi, lo + static_cast<uint32_t>(kMaxAnchorGap) + 1u);
    while (q < stop && !(cls[q] & kClsSpace)) q++;
    if (q - lo <= kMaxAnchorGap) continue;       // a short word in front of a long blank run
    if (cls[lo] & kClsSpacing) continue;         // CJK char + run: internal word-prefix position, lane walk
    const uint32_t i = atomicAdd(count, 1u);
    if (i < cap) list[i] = LongWord{lo, hi};
    anchors[k] = lo | kAnchorSkip;
  }
}

// successor / id per position of the long words' compact index space (cf. long_word_next_kernel)
__global__ __launch_bounds__(kBlock) void fast_long_word_next_kernel(FastArgs a, const LongWord *__restrict__ list,
                                                                     const uint32_t *__restrict__ off, uint32_t nw,
                                                                     uint32_t total, int32_t *__restrict__ id_out,
                                                                     uint32_t *__restrict__ jump,
                                                                     uint8_t *__restrict__ mark) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= total) return;
  const uint32_t w = long_word_of(off, nw, j);
  const LongWord lw = list[w];
  const size_t p = static_cast<size_t>(lw.begin) + (j - off[w]);
  int32_t id = -2;
  uint32_t nx = j;
  if (!(a.cls[p] & kClsSpace)) {
    id = f_match(a, p);
    if (id != -1) {
      const size_t q = p + static_cast<size_t>(a.tok_len[id]);
      if (q < lw.end) nx = j + static_cast<uint32_t>(a.tok_len[id]);
    }
  }
  id_out[j] = id;
  jump[j] = nx;
  mark[j] = j == off[w] ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void fast_long_word_emit_kernel(FastArgs a, const LongWord *__restrict__ list,
                                                                     const uint32_t *__restrict__ off, uint32_t nw,
                                                                     uint32_t total, const int32_t *__restrict__ id,
                                                                     const uint8_t *__restrict__ mark,
                                                                     const uint32_t *__restrict__ word_fail) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j >= total || !mark[j]) return;
  const uint32_t w = long_word_of(off, nw, j);
  const size_t p = static_cast<size_t>(list[w].begin) + (j - off[w]);
  const int32_t t = id[j];
  if (word_fail[w]) {
    if (t == -1) a.emit[p] = a.unk_id;
  } else if (t >= 0) {
    a.emit[p] = t;
  }
}

__global__ __launch_bounds__(kBlock) void fast_walk_kernel(FastArgs a, const uint32_t *__restrict__ anchors,
                                                           const uint32_t *__restrict__ n_anchors_dev, size_t cap) {
  const size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= cap || k >= *n_anchors_dev) return;
  const uint32_t start = anchors[k];
  if (start & kAnchorSkip) return;  // a long word: fast_long_word_* kernels
  SparseOut o{a.emit, a.tok_len};
  fast_walk_from(a, start, o);
}

}  // namespace wp
