// walk.h — the greedy WordPiece walk (linear.cpp:215-274 match_word_piece) and id compaction.
//
// The reference walks the text sequentially (per thread chunk).  The walk's state is memoryless
// whenever it stands on a word-prefix position (tokens_since_prefix == 0), and a position q is
// *certain* to be visited when it is a non-space word-prefix position whose spacing neighbour can
// never lie inside a longer match — i.e. that spacing char does not occur inside any eligible
// multi-char vocab token ("hard" spacing char; for every sane vocabulary all of them are hard).
// Such positions are anchors: one thread starts at each anchor and walks until it reaches the
// next one, so the union of all threads' steps is exactly the reference's sequential walk.
// Tokens are emitted into a text-order array (one slot per code point) and stream-compacted.
#pragma once
#include "primitives.h"
#include "scanline.h"
#include "suffix_array.h"

namespace wp {

constexpr int32_t kNoEmit = static_cast<int32_t>(0x80808080u);  // hipMemset(0x80) pattern; ids are >= -1

struct WalkArgs {
  const uint8_t *cls;
  size_t n_text;
  const RankEntry *rank;
  StepTable steps;
  const int32_t *tok_len;
  int32_t unk_id;
  int32_t *emit;
  int dbg;  // timing experiments only (WP_WALK_DBG): 1 = no emit stores, 2 = no step lookup, 4 = no rank load
};

__device__ __forceinline__ bool w_space(const WalkArgs &a, size_t p) { return a.cls[p] & kClsSpace; }
__device__ __forceinline__ bool w_word_prefix(const WalkArgs &a, size_t p) {  // linear.cpp:215-219
  return p == 0 || (a.cls[p] & kClsSpacing) || (a.cls[p - 1] & kClsSpacing);
}
__device__ __forceinline__ bool w_hard(uint8_t c) { return (c & kClsSpacing) && !(c & kClsSoft); }
__device__ __forceinline__ bool w_anchor(const WalkArgs &a, size_t p) {
  const uint8_t c = a.cls[p];
  if (c & kClsSpace) return false;
  return p == 0 || w_hard(c) || w_hard(a.cls[p - 1]);
}

__device__ inline void walk_from(const WalkArgs &a, size_t p) {
  const size_t end = a.n_text;
  size_t since = p;  // start of the tokens counted by tokens_since_prefix
  while (p < end) {
    const bool prefix = w_word_prefix(a, p);
    const uint32_t r = (a.dbg & 4) ? static_cast<uint32_t>(p) : rank_of(a.rank[p]);
    const int k = (a.dbg & 2) ? 0 : step_lookup(a.steps, r);
    const int32_t id = (a.dbg & 2) ? static_cast<int32_t>(5 + (r & 63)) : (prefix ? a.steps.pval_prefix[k] : a.steps.pval_suffix[k]);
    if (id != -1) {
      if (!(a.dbg & 1)) a.emit[p] = id;
      p += static_cast<size_t>(a.tok_len[id]);
      if (p < end && w_word_prefix(a, p)) since = p;
    } else {
      // roll back this word's tokens (linear.cpp:257-262), then [UNK]
      size_t q = since;
      while (q < p) {
        const int32_t t = a.emit[q];
        a.emit[q] = kNoEmit;
        q += static_cast<size_t>(a.tok_len[t]);
      }
      a.emit[p] = a.unk_id;
      ++p;
      while (p < end && !w_word_prefix(a, p)) ++p;
      since = p;
    }
    while (p < end && w_space(a, p)) ++p;
    if (p >= end || w_anchor(a, p)) return;
    // after skipped spaces p is a word-prefix position: counter restarts
    if (w_word_prefix(a, p)) since = p;
  }
}

// The walk is latency bound (about six dependent loads per token), and only one text position in
// five is an anchor, so anchors are first compacted into a list: every lane of the walk kernel
// then owns one anchor (= one word) and the waves are dense.
// Both anchor kernels give every lane 16 consecutive class bytes (one 16-byte load) plus the byte in
// front of them, and return the lane's anchors as a 16-bit mask.
constexpr int kAnchorBytes = 16;
constexpr int kAnchorTile = kBlock * kAnchorBytes;  // 4096 positions per workgroup

__device__ __forceinline__ uint32_t anchor_mask16(const uint8_t *__restrict__ cls, size_t n, size_t i) {
  if (i >= n) return 0u;
  uint32_t w[4];
  if (i + kAnchorBytes <= n) {
    const uint4 v = *reinterpret_cast<const uint4 *>(cls + i);  // cls is 256-byte aligned, i a multiple of 16
    w[0] = v.x;
    w[1] = v.y;
    w[2] = v.z;
    w[3] = v.w;
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      w[q] = 0;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const size_t p = i + 4 * q + j;
        if (p < n) w[q] |= static_cast<uint32_t>(cls[p]) << (8 * j);
      }
    }
  }
  uint32_t space = 0, hard = 0;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    // one bit per byte -> 4-bit field (byte j of the word -> bit j)
    const uint32_t sp = w[q] & 0x01010101u;
    const uint32_t hd = (w[q] >> 1) & ~(w[q] >> 2) & 0x01010101u;
    space |= ((sp * 0x01020408u) >> 24) << (4 * q);
    hard |= ((hd * 0x01020408u) >> 24) << (4 * q);
  }
  uint32_t before = 1u;  // position 0 counts as preceded by a hard boundary
  if (i > 0) before = w_hard(cls[i - 1]) ? 1u : 0u;
  uint32_t valid = 0xffffu;
  if (i + kAnchorBytes > n) valid = (1u << (n - i)) - 1u;
  return ~space & (hard | (hard << 1) | before) & valid;
}

__global__ __launch_bounds__(kBlock) void anchor_count_kernel(const uint8_t *__restrict__ cls, size_t n,
                                                              uint32_t *__restrict__ tile_counts) {
  __shared__ uint32_t sm[8];
  const size_t i = static_cast<size_t>(blockIdx.x) * kAnchorTile + static_cast<size_t>(threadIdx.x) * kAnchorBytes;
  const uint32_t c = __popc(anchor_mask16(cls, n, i));
  uint32_t tot;
  (void)block_excl_sum(c, sm, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// the list stays in text order: lane offsets come from a workgroup scan of the per-lane counts
__global__ __launch_bounds__(kBlock) void anchor_write_kernel(const uint8_t *__restrict__ cls, size_t n,
                                                              const uint32_t *__restrict__ tile_prefix,
                                                              uint32_t *__restrict__ anchors) {
  __shared__ uint32_t sm[8];
  const size_t i = static_cast<size_t>(blockIdx.x) * kAnchorTile + static_cast<size_t>(threadIdx.x) * kAnchorBytes;
  uint32_t m = anchor_mask16(cls, n, i);
  uint32_t tot;
  uint32_t o = tile_prefix[blockIdx.x] + block_excl_sum(static_cast<uint32_t>(__popc(m)), sm, tot);
  while (m) {
    const int j = __ffs(static_cast<int>(m)) - 1;
    m &= m - 1;
    anchors[o++] = static_cast<uint32_t>(i + j);
  }
}

__global__ __launch_bounds__(kBlock) void walk_kernel(WalkArgs a, const uint32_t *__restrict__ anchors,
                                                      const uint32_t *__restrict__ n_anchors_dev, size_t cap) {
  const size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k == 0) {
    // the reference skips leading whitespace first (linear.cpp:227-229); if the first real
    // position is not an anchor by itself, this thread owns it
    size_t q = 0;
    while (q < a.n_text && w_space(a, q)) ++q;
    if (q < a.n_text && q != 0 && !w_anchor(a, q)) walk_from(a, q);
  }
  if (k >= cap || k >= *n_anchors_dev) return;
  walk_from(a, anchors[k]);
}

// ---- compaction of emit[] into the id stream ----------------------------------------------------
__global__ __launch_bounds__(kBlock) void emit_count_kernel(const int32_t *__restrict__ emit, size_t n,
                                                            uint32_t *__restrict__ tile_counts) {
  __shared__ uint32_t sm[8];
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; j++) {
    size_t i = base + static_cast<size_t>(j) * kBlock + threadIdx.x;
    if (i < n && emit[i] != kNoEmit) c++;
  }
  uint32_t tot;
  (void)block_excl_sum(c, sm, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

__global__ __launch_bounds__(kBlock) void emit_write_kernel(const int32_t *__restrict__ emit, size_t n,
                                                            const uint32_t *__restrict__ tile_prefix,
                                                            int32_t *__restrict__ ids) {
  __shared__ uint32_t sm[8];
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile + static_cast<size_t>(threadIdx.x) * kScanItems;
  int32_t v[kScanItems];
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; j++) {
    size_t i = base + j;
    v[j] = i < n ? emit[i] : kNoEmit;
    c += v[j] != kNoEmit;
  }
  uint32_t tot;
  size_t o = static_cast<size_t>(block_excl_sum(c, sm, tot)) + tile_prefix[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanItems; j++) {
    if (v[j] != kNoEmit) ids[o++] = v[j];
  }
}

}  // namespace wp
