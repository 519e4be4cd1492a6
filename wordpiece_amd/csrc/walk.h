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

constexpr int kReachTile = kBlock * 8;  // tile of the coverage-rule kernels (a power of two)
constexpr size_t kMaxAnchorGap = 2048;  // longer gaps between class-rule anchors switch the anchor rule
constexpr int32_t kNoEmit = static_cast<int32_t>(0x80808080u);  // hipMemset(0x80) pattern; ids are >= -1

struct WalkArgs {
  const uint8_t *cls;
  size_t n_text;
  const RankEntry *rank;
  StepTable steps;
  const int32_t *tok_len;
  int32_t unk_id;
  int32_t *emit;
  const uint8_t *aflags;  // != nullptr: anchors are given per position (coverage-based anchors, below)
  const uint32_t *wp_from_tile;  // with aflags: first word-prefix position >= t * kReachTile (n_text: none)
  const uint32_t *ns_from_tile;  // with aflags: first non-space position >= t * kReachTile (n_text: none)
  int all_hard;  // no spacing char occurs inside an eligible multi-char token (every sane vocabulary)
  int32_t n_tokens;  // vocab lines (debug build: range check of the ids that come out of the step table)
};

__device__ __forceinline__ bool w_space(const WalkArgs &a, size_t p) { return a.cls[p] & kClsSpace; }
__device__ __forceinline__ bool w_word_prefix(const WalkArgs &a, size_t p) {  // linear.cpp:215-219
  return p == 0 || (a.cls[p] & kClsSpacing) || (a.cls[p - 1] & kClsSpacing);
}
__device__ __forceinline__ bool w_hard(uint8_t c) { return (c & kClsSpacing) && !(c & kClsSoft); }
__device__ __forceinline__ bool w_anchor(const WalkArgs &a, size_t p) {
  if (a.aflags) return a.aflags[p] != 0;
  const uint8_t c = a.cls[p];
  if (c & kClsSpace) return false;
  return p == 0 || w_hard(c) || w_hard(a.cls[p - 1]);
}

// ---- where a lane's ids go ---------------------------------------------------------------------------
// SparseOut: emit[p] = id at the token's first position; emit_count / emit_write compact the array afterwards
// (needed when several kernels contribute ids: long words, coverage anchors).
// StagedOut: the ids of a word stay with it — the first kStageIds in LDS, the rest at emit[p0 + j], inside the
// stretch of text only this word's lane walks — and the workgroup appends them, word after word, to a compact list that
// starts at the position of its first anchor (a token consumes at least one position, so the list fits in
// front of the next workgroup's first anchor).  emit_gather_kernel then moves whole lists: the id stream is
// written and read once, 4 bytes per id, instead of a cleared 4-byte slot per text position.
constexpr int kStageIds = 4;

struct SparseOut {
  int32_t *emit;
  const int32_t *tok_len;
  __device__ __forceinline__ void push(size_t p, int32_t id) { emit[p] = id; }
  __device__ __forceinline__ void word_start() {}
  __device__ __forceinline__ void rollback(size_t since, size_t p) {  // linear.cpp:257-262, fast.cpp:80-89
    size_t q = since;
    while (q < p) {
      const int32_t t = emit[q];
      emit[q] = kNoEmit;
      q += static_cast<size_t>(tok_len[t]);
    }
  }
};

struct StagedOut {
  int32_t *stage;  // LDS: the column of this lane (or of the word the lane is on), rows `stride` apart
  int32_t *spill;  // emit + p0
  uint32_t c = 0, mark = 0;
  uint32_t stride = kBlock;
  __device__ __forceinline__ void push(size_t, int32_t id) {
    if (c < static_cast<uint32_t>(kStageIds)) {
      stage[c * stride] = id;
    } else {
      spill[c] = id;
    }
    c++;
  }
  __device__ __forceinline__ void word_start() { mark = c; }
  __device__ __forceinline__ void rollback(size_t, size_t) { c = mark; }
  __device__ __forceinline__ int32_t get(uint32_t j) const {
    return j < static_cast<uint32_t>(kStageIds) ? stage[j * stride] : spill[j];
  }
};

// One step of a lane's walk: the token at s.p (or the [UNK] of its word) and what follows it up to the next
// token start.  Returns true when the lane's stretch of text ends (end of text, or the next anchor).
struct WalkState {
  size_t p, since;  // position; start of the tokens counted by tokens_since_prefix
};

// The class bytes of positions p-1 .. p+14 in two registers, loaded next to the rank (a token is a chain of
// dependent loads: every class test behind the token's end would otherwise add a link to it)
struct StepWin {
  uint64_t w0, w1;
  size_t wbase;
  bool win;
};
__device__ __forceinline__ StepWin step_window(const WalkArgs &a, size_t p) {
  StepWin W{0, 0, p ? p - 1 : 0, false};
  W.win = W.wbase + 16 <= a.n_text;
  if (W.win) {
    __builtin_memcpy(&W.w0, a.cls + W.wbase, 8);
    __builtin_memcpy(&W.w1, a.cls + W.wbase + 8, 8);
  }
  return W;
}
__device__ __forceinline__ uint8_t step_cb(const WalkArgs &a, const StepWin &W, size_t q) {
  const size_t d = q - W.wbase;
  if (W.win && d < 16) return static_cast<uint8_t>((d < 8 ? W.w0 >> (8 * d) : W.w1 >> (8 * (d - 8))) & 0xffu);
  return a.cls[q];
}
__device__ __forceinline__ bool step_word_prefix(const WalkArgs &a, const StepWin &W, size_t q) {
  return q == 0 || (step_cb(a, W, q) & kClsSpacing) || (step_cb(a, W, q - 1) & kClsSpacing);
}

// the part of a step behind the lookup: raw = the step table's value for s.p (step_raw)
template <typename Out>
__device__ __forceinline__ bool walk_finish(const WalkArgs &a, WalkState &s, Out &o, const StepWin &W, int32_t raw) {
  const size_t end = a.n_text;
  size_t p = s.p;
  auto cb = [&](size_t q) -> uint8_t { return step_cb(a, W, q); };
  auto word_prefix = [&](size_t q) { return step_word_prefix(a, W, q); };
  auto space = [&](size_t q) { return (cb(q) & kClsSpace) != 0; };
  auto anchor = [&](size_t q) {
    if (a.aflags) return a.aflags[q] != 0;
    const uint8_t c = cb(q);
    if (c & kClsSpace) return false;
    return q == 0 || w_hard(c) || w_hard(cb(q - 1));
  };
  int32_t id = step_id(a.steps, raw);
  if (!wp_in_bounds(id >= -1 && id < a.n_tokens, kSiteTokenId)) id = -1;
  if (id != -1) {
    o.push(p, id);
    p += static_cast<size_t>(step_len(a.steps, raw, a.tok_len));
    if (p < end && word_prefix(p)) {
      s.since = p;
      o.word_start();
    }
  } else {
    // roll back this word's tokens (linear.cpp:257-262), then [UNK]
    o.rollback(s.since, p);
    o.push(p, a.unk_id);
    ++p;
    while (p < end && !word_prefix(p)) {
      ++p;
      // coverage mode (texts with very long words): jump over whole tiles without a word-prefix
      // position instead of stepping through them (a 10 M-char word otherwise costs a lane 2 s)
      if (a.wp_from_tile && (p & (kReachTile - 1)) == 0 && p < end) {
        p = min(static_cast<size_t>(a.wp_from_tile[p / kReachTile]), end);
        break;
      }
    }
    s.since = p;
    o.word_start();
  }
  s.p = p;
  if (p < end && space(p)) {
    // class rule with only hard spacing chars: the first position behind the spaces is an anchor of
    // its own, whatever it is — no need to step through the run (a megabyte of blanks otherwise
    // stalls this lane for 0.2 s)
    if (!a.aflags && a.all_hard) return true;
    while (p < end && space(p)) {
      ++p;
      if (a.ns_from_tile && (p & (kReachTile - 1)) == 0 && p < end) {  // coverage mode: jump over blank tiles
        p = min(static_cast<size_t>(a.ns_from_tile[p / kReachTile]), end);
        break;
      }
    }
    s.p = p;
  }
  if (p >= end || anchor(p)) return true;
  // after skipped spaces p is a word-prefix position: counter restarts
  if (word_prefix(p)) {
    s.since = p;
    o.word_start();
  }
  return false;
}

template <typename Out>
__device__ __forceinline__ bool walk_step(const WalkArgs &a, WalkState &s, Out &o) {
  const StepWin W = step_window(a, s.p);
  const uint32_t r = rank_of(a.rank[s.p]);
  return walk_finish(a, s, o, W, step_raw(a.steps, r, step_word_prefix(a, W, s.p)));
}

template <typename Out>
__device__ inline void walk_from(const WalkArgs &a, size_t p, Out &o) {
  WalkState s{p, p};
  o.word_start();
  while (s.p < a.n_text) {
    if (walk_step(a, s, o)) return;
  }
}

// The walk is latency bound (about six dependent loads per token), and only one text position in
// five is an anchor, so anchors are first compacted into a list: every lane of the walk kernel
// then owns one anchor (= one word) and the waves are dense.
// Both anchor kernels give every lane 16 consecutive class bytes (one 16-byte load) plus the byte in
// front of them, and return the lane's anchors as a 16-bit mask.
constexpr int kAnchorBytes = 16;
constexpr int kAnchorTile = kBlock * kAnchorBytes;  // 4096 positions per workgroup

__device__ __forceinline__ uint32_t anchor_mask16(const uint8_t *__restrict__ cls, const uint8_t *__restrict__ aflags,
                                                  size_t n, size_t i) {
  if (i >= n) return 0u;
  uint32_t w[4];
  if (aflags) cls = aflags;  // per-position flags (0/1) instead of the class rule
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
  uint32_t valid = 0xffffu;
  if (i + kAnchorBytes > n) valid = (1u << (n - i)) - 1u;
  if (aflags) {
    uint32_t m = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) m |= (((w[q] & 0x01010101u) * 0x01020408u) >> 24) << (4 * q);
    return m & valid;
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
  return ~space & (hard | (hard << 1) | before) & valid;
}

__global__ __launch_bounds__(kBlock) void anchor_count_kernel(const uint8_t *__restrict__ cls,
                                                              const uint8_t *__restrict__ aflags, size_t n,
                                                              uint32_t *__restrict__ tile_counts) {
  __shared__ uint32_t sm[8];
  const size_t i = static_cast<size_t>(blockIdx.x) * kAnchorTile + static_cast<size_t>(threadIdx.x) * kAnchorBytes;
  const uint32_t c = __popc(anchor_mask16(cls, aflags, n, i));
  uint32_t tot;
  (void)block_excl_sum(c, sm, tot);
  if (threadIdx.x == 0) tile_counts[blockIdx.x] = tot;
}

// the list stays in text order: lane offsets come from a workgroup scan of the per-lane counts.  The kernel also merges
// what it knows into the class bytes (kClsWordPrefix, kClsAnchor): the walk then needs one byte of the position it
// lands on instead of that byte, its neighbour and (coverage rule) the flag array.
__global__ __launch_bounds__(kBlock) void anchor_write_kernel(uint8_t *__restrict__ cls, const uint8_t *__restrict__ aflags,
                                                              size_t n, const uint32_t *__restrict__ tile_prefix,
                                                              uint32_t *__restrict__ anchors) {
  __shared__ uint32_t sm[8];
  const size_t i = static_cast<size_t>(blockIdx.x) * kAnchorTile + static_cast<size_t>(threadIdx.x) * kAnchorBytes;
  uint32_t m = anchor_mask16(cls, aflags, n, i);
  if (i < n) {
    uint32_t w[4] = {0, 0, 0, 0};
    const int cnt = static_cast<int>(min(static_cast<size_t>(kAnchorBytes), n - i));
    if (cnt == kAnchorBytes) {
      const uint4 v = *reinterpret_cast<const uint4 *>(cls + i);
      w[0] = v.x;
      w[1] = v.y;
      w[2] = v.z;
      w[3] = v.w;
    } else {
      for (int j = 0; j < cnt; j++) w[j >> 2] |= static_cast<uint32_t>(cls[i + j]) << (8 * (j & 3));
    }
    uint32_t spc = 0;  // bit j: position i + j is a spacing char
#pragma unroll
    for (int q = 0; q < 4; q++) spc |= ((((w[q] >> 1) & 0x01010101u) * 0x01020408u) >> 24) << (4 * q);
    const uint32_t before = i == 0 ? 1u : ((cls[i - 1] & kClsSpacing) ? 1u : 0u);  // (the neighbour's low bits never change)
    const uint32_t wp = (spc | (spc << 1) | before) & 0xffffu;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      // 4 mask bits -> bit 0 of 4 bytes
      const uint32_t wq = (((wp >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u;
      const uint32_t aq = (((m >> (4 * q)) & 15u) * 0x00204081u) & 0x01010101u;
      w[q] = (w[q] & 0x0f0f0f0fu) | (wq << 4) | (aq << 5);
    }
    if (cnt == kAnchorBytes) {
      *reinterpret_cast<uint4 *>(cls + i) = uint4{w[0], w[1], w[2], w[3]};
    } else {
      for (int j = 0; j < cnt; j++) cls[i + j] = static_cast<uint8_t>(w[j >> 2] >> (8 * (j & 3)));
    }
  }
  uint32_t tot;
  uint32_t o = tile_prefix[blockIdx.x] + block_excl_sum(static_cast<uint32_t>(__popc(m)), sm, tot);
  while (m) {
    const int j = __ffs(static_cast<int>(m)) - 1;
    m &= m - 1;
    anchors[o++] = static_cast<uint32_t>(i + j);
  }
}

// ---- coverage-based anchors --------------------------------------------------------------------
// The class rule above yields no anchors inside text whose spacing chars are all "soft" (e.g. CJK
// text with a vocabulary that holds multi-char CJK tokens): one lane would walk the whole stretch.
// The encoder measures the largest gap between anchors and, if it is long, derives the anchors from
// the matches themselves.  The walk's position sequence is a function of the position alone: from q
// it moves by len(best token at q), or (no token) to the next word-prefix position, then over
// spaces.  So a non-space word-prefix position p that no match starting before it reaches across
// (max over q < p of q + len(q) <= p) is always landed on, with the per-word state reset — exactly
// what an anchor needs.  Cost: one match lookup per text position instead of one per visited one.
__global__ __launch_bounds__(kBlock) void anchor_gap_kernel(const uint32_t *__restrict__ anchors,
                                                            const uint32_t *__restrict__ n_anchors_dev, size_t n_text,
                                                            const uint8_t *__restrict__ cls, int all_hard,
                                                            uint32_t *__restrict__ max_gap) {
  __shared__ int32_t sm[8];
  const size_t na = *n_anchors_dev;
  uint32_t g = 0;
  for (size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; k <= na;
       k += static_cast<size_t>(gridDim.x) * kBlock) {
    const uint32_t hi = k < na ? anchors[k] : static_cast<uint32_t>(n_text);
    const uint32_t lo = k > 0 ? anchors[k - 1] : 0u;
    uint32_t d = hi - lo;
    if (d > kMaxAnchorGap && all_hard) {
      // with only hard spacing chars a lane stops at the first space (walk_from), so a long blank run
      // behind a word is not a long walk: measure up to the first space (k == 0: leading blanks are
      // nobody's walk)
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

// The coverage rule costs a match lookup per text position, and only the stretches the class rule leaves without
// anchors need it: gaps of more than kMaxAnchorGap positions between two class-rule anchors (a Chinese paragraph
// without blanks next to English, Russian or Japanese text that the class rule handles fine).  A tile of kReachTile
// <= kMaxAnchorGap positions meets at most two such gaps — one that holds its first position, one that holds its
// last — so per tile two numbers say where the coverage rule applies: [tile start, gap_a_end) and [gap_b_start, tile
// end); everywhere else the class rule stands.  (No match crosses a class-rule anchor — a hard spacing char occurs
// inside no token — so the cover of a gap starts at its own first position.)  all != 0: the whole text (WP_OPT_COVER_ANCHORS).
static_assert(kReachTile <= static_cast<int>(kMaxAnchorGap), "a tile meets at most two long gaps");
__global__ __launch_bounds__(kBlock) void gap_tiles_kernel(const uint32_t *__restrict__ anchors,
                                                           const uint32_t *__restrict__ n_anchors_dev, size_t n_text,
                                                           unsigned tiles, int all, uint32_t *__restrict__ gap_a_end,
                                                           uint32_t *__restrict__ gap_b_start) {
  const unsigned t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= tiles) return;
  const uint32_t t0 = t * static_cast<uint32_t>(kReachTile);
  const uint32_t t1 = static_cast<uint32_t>(min(static_cast<size_t>(t0) + kReachTile, n_text));
  if (all) {
    gap_a_end[t] = t1;
    gap_b_start[t] = t0;
    return;
  }
  const uint32_t na = *n_anchors_dev;
  auto gap_of = [&](uint32_t p, uint32_t &lo, uint32_t &hi) {  // the stretch between two anchors that holds p
    uint32_t a = 0, b = na;                                    // first anchor > p
    while (a < b) {
      const uint32_t md = (a + b) >> 1;
      if (anchors[md] <= p) a = md + 1; else b = md;
    }
    lo = a ? anchors[a - 1] : 0u;
    hi = a < na ? anchors[a] : static_cast<uint32_t>(n_text);
  };
  uint32_t lo, hi;
  gap_of(t0, lo, hi);
  gap_a_end[t] = hi - lo > kMaxAnchorGap ? min(hi, t1) : t0;
  gap_of(t1 - 1, lo, hi);
  gap_b_start[t] = hi - lo > kMaxAnchorGap ? max(lo, t0) : t1;
}

// reach[q] = q + length of the token the walk would take at q (q itself: none, or a space); tiles the coverage
// rule does not apply to are skipped
__global__ __launch_bounds__(kBlock) void reach_kernel(WalkArgs a, uint32_t *__restrict__ reach,
                                                       uint32_t *__restrict__ tile_max,
                                                       const uint32_t *__restrict__ gap_a_end,
                                                       const uint32_t *__restrict__ gap_b_start) {
  __shared__ int32_t sm[8];
  const size_t base = static_cast<size_t>(blockIdx.x) * kReachTile;
  {
    const uint32_t t0 = static_cast<uint32_t>(base);
    const uint32_t t1 = static_cast<uint32_t>(min(base + kReachTile, a.n_text));
    if (gap_a_end[blockIdx.x] <= t0 && gap_b_start[blockIdx.x] >= t1) {  // (uniform)
      if (threadIdx.x == 0) tile_max[blockIdx.x] = 0u;
      return;
    }
  }
  uint32_t mx = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const size_t q = base + static_cast<size_t>(j) * kBlock + threadIdx.x;
    if (q < a.n_text) {
      uint32_t r = static_cast<uint32_t>(q);
      if (!w_space(a, q)) {
        const int32_t raw = step_raw(a.steps, rank_of(a.rank[q]), (a.cls[q] & kClsWordPrefix) != 0);
        if (step_id(a.steps, raw) != -1) r += static_cast<uint32_t>(step_len(a.steps, raw, a.tok_len));
      }
      reach[q] = r;
      mx = max(mx, r);
    }
  }
  const int32_t m = -block_reduce_min(-static_cast<int32_t>(mx), sm);
  if (threadIdx.x == 0) tile_max[blockIdx.x] = static_cast<uint32_t>(m);
}

// in place: tile_max[t] <- max over tiles before t (single workgroup)
__global__ __launch_bounds__(1024) void reach_spine_kernel(uint32_t *__restrict__ tile_max, size_t tiles) {
  __shared__ int32_t wm[2][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int kTrip = 4;  // (steps per trip, their loads issued together: suffix_min_kernel)
  int32_t carry = 0;
  int buf = 0;
  for (size_t base = 0; base < tiles; base += 1024 * kTrip) {
    int32_t v[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; k++) {
      const size_t i = base + static_cast<size_t>(k) * 1024 + threadIdx.x;
      v[k] = i < tiles ? static_cast<int32_t>(tile_max[i]) : 0;
    }
#pragma unroll
    for (int k = 0; k < kTrip; k++) {
      const size_t i = base + static_cast<size_t>(k) * 1024 + threadIdx.x;
      const int32_t inc = wave_incl_max(v[k]);
      if (lane == kWave - 1) wm[buf][w] = inc;
      __syncthreads();  // (two buffers taking turns)
      int32_t before = carry, all = carry;
      for (int q = 0; q < 16; q++) {
        if (q < w) before = max(before, wm[buf][q]);
        all = max(all, wm[buf][q]);
      }
      const int32_t up = __shfl_up(inc, 1, kWave);
      if (i < tiles) tile_max[i] = static_cast<uint32_t>(lane == 0 ? before : max(before, up));
      carry = all;
      buf ^= 1;
    }
  }
}

// aflags[p] = 1 iff p is a non-space word-prefix position that no earlier match reaches across (inside the long
// gaps, gap_tiles_kernel), or a class-rule anchor (everywhere else)
__global__ __launch_bounds__(kBlock) void cover_flags_kernel(const uint8_t *__restrict__ cls,
                                                             const uint32_t *__restrict__ reach,
                                                             const uint32_t *__restrict__ tile_before, size_t n,
                                                             uint8_t *__restrict__ aflags,
                                                             uint32_t *__restrict__ tile_first_wp,
                                                             uint32_t *__restrict__ tile_first_ns,
                                                             const uint32_t *__restrict__ gap_a_end,
                                                             const uint32_t *__restrict__ gap_b_start,
                                                             uint32_t *__restrict__ anchor_cnt) {
  // anchor_cnt (cleared by the caller): the flags set per tile of kAnchorTile positions — what anchor_count_kernel
  // would count in a pass of its own
  static_assert(kAnchorTile % kReachTile == 0, "a tile of this kernel lies inside one tile of the anchor list");
  __shared__ int32_t wm[4];
  __shared__ int32_t sm_min[8];
  uint32_t set = 0;
  int32_t first_wp = 0x7fffffff, first_ns = 0x7fffffff;  // first word-prefix / non-space position of this thread
  const int lane = lane_id(), w = wave_id();
  const size_t p0 = static_cast<size_t>(blockIdx.x) * kReachTile + static_cast<size_t>(threadIdx.x) * 8;
  const uint32_t ga = gap_a_end[blockIdx.x], gb = gap_b_start[blockIdx.x];
  const uint32_t tile0 = blockIdx.x * static_cast<uint32_t>(kReachTile);
  const bool any_cover = ga > tile0 || static_cast<size_t>(gb) < min(static_cast<size_t>(tile0) + kReachTile, n);  // (uniform)
  int32_t r[8], mx = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    r[j] = (any_cover && p0 + j < n) ? static_cast<int32_t>(reach[p0 + j]) : 0;
    mx = max(mx, r[j]);
  }
  const int32_t inc = wave_incl_max(mx);
  if (lane == kWave - 1) wm[w] = inc;
  __syncthreads();
  int32_t cover = static_cast<int32_t>(tile_before[blockIdx.x]);  // reach of everything before this thread
  for (int q = 0; q < w; q++) cover = max(cover, wm[q]);
  const int32_t up = __shfl_up(inc, 1, kWave);
  if (lane > 0) cover = max(cover, up);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const size_t p = p0 + j;
    if (p < n) {
      const uint8_t c = cls[p];
      const uint8_t cp = p ? cls[p - 1] : 0;
      const bool wp = p == 0 || (c & kClsSpacing) || (cp & kClsSpacing);
      const bool covered_rule = p < ga || p >= gb;
      const bool anchor = covered_rule ? (wp && cover <= static_cast<int32_t>(p)) : (p == 0 || w_hard(c) || w_hard(cp));
      const bool flag = !(c & kClsSpace) && anchor;
      aflags[p] = flag ? 1 : 0;
      set += flag ? 1u : 0u;
      cover = max(cover, r[j]);
      if (wp) first_wp = min(first_wp, static_cast<int32_t>(p));
      if (!(c & kClsSpace)) first_ns = min(first_ns, static_cast<int32_t>(p));
    }
  }
  const int32_t m = block_reduce_min(first_wp, sm_min);
  const int32_t m2 = block_reduce_min(first_ns, sm_min);
  uint32_t nset = 0;
  (void)block_excl_sum(set, reinterpret_cast<uint32_t *>(sm_min), nset);  // (starts with a barrier of its own)
  if (threadIdx.x == 0) {
    if (anchor_cnt && nset) atomicAdd(&anchor_cnt[blockIdx.x / (kAnchorTile / kReachTile)], nset);
    tile_first_wp[blockIdx.x] = m == 0x7fffffff ? static_cast<uint32_t>(n) : static_cast<uint32_t>(m);
    tile_first_ns[blockIdx.x] = m2 == 0x7fffffff ? static_cast<uint32_t>(n) : static_cast<uint32_t>(m2);
  }
}

// in place: t[i] <- min over tiles >= i (one workgroup per array, from the back).  Four steps of 1024 tiles per trip:
// their loads are issued together (the loop is a chain of load -> scan -> store, one memory latency per step
// otherwise: 0.28 ms for the 270 K tiles of config 3).
__global__ __launch_bounds__(1024) void suffix_min_kernel(uint32_t *__restrict__ t0, uint32_t *__restrict__ t1, size_t tiles) {
  uint32_t *__restrict__ t = blockIdx.x == 0 ? t0 : t1;
  __shared__ uint32_t wm[2][16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int kTrip = 4;
  uint32_t carry = 0xffffffffu;
  int buf = 0;
  for (size_t done = 0; done < tiles; done += 1024 * kTrip) {
    uint32_t v[kTrip];
    size_t idx[kTrip];
    bool ok[kTrip];
#pragma unroll
    for (int k = 0; k < kTrip; k++) {
      // step k covers indices [hi - 1024, hi) from the back; thread x owns index hi - 1 - x
      const size_t off = done + static_cast<size_t>(k) * 1024;
      const size_t hi = off < tiles ? tiles - off : 0;
      ok[k] = threadIdx.x < hi;
      idx[k] = ok[k] ? hi - 1 - threadIdx.x : 0;
      v[k] = ok[k] ? t[idx[k]] : 0xffffffffu;
    }
#pragma unroll
    for (int k = 0; k < kTrip; k++) {
      uint32_t x = v[k];
#pragma unroll
      for (int d = 1; d < kWave; d <<= 1) {
        const uint32_t u = __shfl_up(x, d, kWave);
        if (lane >= d) x = min(x, u);
      }
      if (lane == kWave - 1) wm[buf][w] = x;
      __syncthreads();  // (two buffers taking turns: the next step's writes cannot meet this step's reads)
      uint32_t before = carry, all = carry;
      for (int q = 0; q < 16; q++) {
        if (q < w) before = min(before, wm[buf][q]);
        all = min(all, wm[buf][q]);
      }
      if (ok[k]) t[idx[k]] = min(x, before);
      carry = all;
      buf ^= 1;
    }
  }
}

// ---- long words -------------------------------------------------------------------------------------
// With only hard spacing chars every word is walked on its own, but a lane still takes its word token
// by token: a megabyte of base64 or the reference's 10 M-character single-word stress
// (tests.cpp:266-272) would keep one lane busy for 0.1-1 s.  Words longer than kMaxAnchorGap are
// therefore taken out of the lane walk: the positions of all such words are laid out in one compact
// index space, every position gets its successor j + len(best token) (terminal: no token = the
// word fails, or the word's end), the chain from each word start is marked by pointer doubling
// (log2(longest word) rounds of "mark my successor; successor <- successor of successor"), and the
// marked positions emit their ids in parallel — or, if the chain reached a position without a token,
// the single [UNK] of linear.cpp:257-266.
struct LongWord {
  uint32_t begin, end;  // [this anchor, next anchor): the word and the blanks behind it
};
constexpr uint32_t kAnchorSkip = 0x80000000u;  // flag in the anchor list: not for the lane walk

__global__ __launch_bounds__(kBlock) void long_word_collect_kernel(uint32_t *__restrict__ anchors,
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
    if (q - lo <= kMaxAnchorGap) continue;  // a short word in front of a long blank run
    const uint32_t i = atomicAdd(count, 1u);
    if (i < cap) list[i] = LongWord{lo, hi};
    anchors[k] = lo | kAnchorSkip;
  }
}

__device__ __forceinline__ uint32_t long_word_of(const uint32_t *__restrict__ off, uint32_t nw, uint32_t j) {
  uint32_t lo = 0, hi = nw;  // last word with off[w] <= j
  while (hi - lo > 1) {
    const uint32_t md = (lo + hi) >> 1;
    if (off[md] <= j) lo = md; else hi = md;
  }
  return lo;
}

// id[j]: >= 0 token, -1 no token (the word fails here), -2 blank (the word is over); jump[j]: successor
// in the compact index space, j itself for the terminal cases; mark[j] = 1 at the word starts
__global__ __launch_bounds__(kBlock) void long_word_next_kernel(WalkArgs a, const LongWord *__restrict__ list,
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
  if (!w_space(a, p)) {
    const int32_t raw = step_raw(a.steps, rank_of(a.rank[p]), (a.cls[p] & kClsWordPrefix) != 0);
    id = step_id(a.steps, raw);
    if (id != -1) {
      const uint32_t len = static_cast<uint32_t>(step_len(a.steps, raw, a.tok_len));
      if (p + len < lw.end) nx = j + len;  // (p + len == end: the last token of the range)
    }
  }
  id_out[j] = id;
  jump[j] = nx;
  mark[j] = j == off[w] ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void long_word_mark_kernel(const uint32_t *__restrict__ jump, uint32_t total,
                                                                uint8_t *__restrict__ mark) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j < total && mark[j]) mark[jump[j]] = 1;
}

__global__ __launch_bounds__(kBlock) void long_word_double_kernel(const uint32_t *__restrict__ jump_in, uint32_t total,
                                                                  uint32_t *__restrict__ jump_out) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j < total) jump_out[j] = jump_in[jump_in[j]];
}

__global__ __launch_bounds__(kBlock) void long_word_fail_kernel(const int32_t *__restrict__ id, const uint8_t *__restrict__ mark,
                                                                const uint32_t *__restrict__ off, uint32_t nw,
                                                                uint32_t total, uint32_t *__restrict__ word_fail) {
  const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
  if (j < total && mark[j] && id[j] == -1) word_fail[long_word_of(off, nw, j)] = 1u;
}

__global__ __launch_bounds__(kBlock) void long_word_emit_kernel(WalkArgs a, const LongWord *__restrict__ list,
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
    if (t == -1) a.emit[p] = a.unk_id;  // the one [UNK] of the word, where its chain broke
  } else if (t >= 0) {
    a.emit[p] = t;
  }
}

__global__ __launch_bounds__(kBlock) void walk_kernel(WalkArgs a, const uint32_t *__restrict__ anchors,
                                                      const uint32_t *__restrict__ n_anchors_dev, size_t cap) {
  const size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k == 0 && !a.aflags && !a.all_hard) {
    // the reference skips leading whitespace first (linear.cpp:227-229); if the first real
    // position is not an anchor by itself, this thread owns it.  (Only with soft spacing chars under
    // the class rule: otherwise that position always is an anchor — behind a hard space, or, under
    // the coverage rule, a word-prefix position no match reaches, since spaces match nothing.)
    size_t q = 0;
    while (q < a.n_text && w_space(a, q)) ++q;
    SparseOut o{a.emit, a.tok_len};
    if (q < a.n_text && q != 0 && !w_anchor(a, q)) walk_from(a, q, o);
  }
  if (k >= cap || k >= *n_anchors_dev) return;
  const uint32_t start = anchors[k];
  if (start & kAnchorSkip) return;  // a long word: long_word_* kernels
  SparseOut o{a.emit, a.tok_len};
  walk_from(a, start, o);
}

// ---- wide walk: a whole wave per long word ---------------------------------------------------------------------
// A lane takes its word token by token, and a token is a chain of ~5 dependent loads: a 512-character word of
// one-character pieces (config 5: 200 tokens per word, every word long) keeps its lane busy for 200 such chains,
// whatever the other lanes do.  Under the class rule with only hard spacing chars a stretch [anchor, next anchor)
// is one word and the blanks behind it, so a stretch of more than kWideMin positions is handed to a whole wave: it
// looks up the token of EVERY position of the word, 256 positions per round trip (four independent chains per
// lane), finds the positions the greedy walk stands on by pointer doubling over their successors in LDS
// (8 steps for 256 positions), and the lanes on the chain write their ids, in order, into the word's own
// stretch of the scratch array — where the lane walk's list assembly picks them up (count | kWideFlag in
// wide_cnt[anchor index]).  A position of the chain without a token makes the whole word [UNK] (linear.cpp:257-272).
constexpr uint32_t kWideMin = 48;
constexpr uint32_t kWideFlag = 0x80000000u;
constexpr int kWidePerLane = 4;
constexpr int kWideWindow = kWave * kWidePerLane;  // positions per round trip

__global__ __launch_bounds__(kBlock) void wide_collect_kernel(const uint32_t *__restrict__ anchors,
                                                              const uint32_t *__restrict__ n_anchors_dev, size_t n_text,
                                                              uint32_t *__restrict__ list, uint32_t *__restrict__ count) {
  const size_t na = *n_anchors_dev;
  for (size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; k < na;
       k += static_cast<size_t>(gridDim.x) * kBlock) {
    const uint32_t lo = anchors[k];
    const uint32_t hi = k + 1 < na ? anchors[k + 1] : static_cast<uint32_t>(n_text);
    if (hi - lo > kWideMin) list[atomicAdd(count, 1u)] = static_cast<uint32_t>(k);
  }
}

__global__ __launch_bounds__(kBlock) void walk_wide_kernel(WalkArgs a, const uint32_t *__restrict__ anchors,
                                                           const uint32_t *__restrict__ n_anchors_dev,
                                                           const uint32_t *__restrict__ list,
                                                           const uint32_t *__restrict__ count,
                                                           uint32_t *__restrict__ wide_cnt) {
  constexpr int WAVES = kBlock / kWave;
  constexpr uint32_t kEnd = 0xffffu;  // successor of a position behind the word's end
  __shared__ uint16_t s_jump_mem[WAVES][2][kWideWindow];
  __shared__ uint8_t s_mark_mem[WAVES][kWideWindow];
  // (volatile: the lanes of a wave talk through these arrays — a value another lane wrote must not be served from a register)
  volatile uint16_t (*s_jump)[2][kWideWindow] = s_jump_mem;
  volatile uint8_t (*s_mark)[kWideWindow] = s_mark_mem;
  const int lane = lane_id(), wv = wave_id();
  const size_t na = *n_anchors_dev;
  const uint32_t n_wide = *count;
  const uint64_t lt = (1ull << lane) - 1ull;
  for (size_t wi = static_cast<size_t>(blockIdx.x) * WAVES + wv; wi < n_wide; wi += static_cast<size_t>(gridDim.x) * WAVES) {
    const uint32_t k = list[wi];
    const uint32_t start = anchors[k];
    const uint32_t next = static_cast<size_t>(k) + 1 < na ? anchors[k + 1] : static_cast<uint32_t>(a.n_text);
    // the word ends where the blanks of the stretch begin (under this rule a stretch is one word and the blanks behind
    // it: looked for from the back — usually one load instead of one per 64 characters of the word)
    uint32_t e = next;
    while (e > start) {
      const uint32_t q = e - 1 - static_cast<uint32_t>(lane);
      const bool in = e - start > static_cast<uint32_t>(lane);
      const uint64_t m = __ballot(in && !(a.cls[q] & kClsSpace));  // bit j: position e-1-j is not a blank
      if (m) {
        e -= static_cast<uint32_t>(__ffsll(static_cast<long long>(m)) - 1);
        break;
      }
      e -= min(e - start, static_cast<uint32_t>(kWave));
    }
    int32_t *out = a.emit + start;
    uint32_t c = 0;
    bool failed = false;
    uint32_t pos = start;
    while (pos < e && !failed) {
      int32_t id[kWidePerLane];
      uint32_t jp[kWidePerLane];
#pragma unroll
      for (int j = 0; j < kWidePerLane; j++) {  // (four independent chains of loads in flight)
        const uint32_t i = static_cast<uint32_t>(j) * kWave + lane;
        const uint32_t q = pos + i;
        id[j] = -1;
        jp[j] = kEnd;
        if (q < e) {
          const int32_t raw = step_raw(a.steps, rank_of(a.rank[q]), (a.cls[q] & kClsWordPrefix) != 0);
          int32_t t = step_id(a.steps, raw);
          if (!wp_in_bounds(t >= -1 && t < a.n_tokens, kSiteTokenId)) t = -1;
          id[j] = t;
          jp[j] = t == -1 ? i : i + static_cast<uint32_t>(step_len(a.steps, raw, a.tok_len));  // (no token: a terminal of its own)
        }
      }
      int cur = 0;
#pragma unroll
      for (int j = 0; j < kWidePerLane; j++) {
        const uint32_t i = static_cast<uint32_t>(j) * kWave + lane;
        s_jump[wv][0][i] = static_cast<uint16_t>(min(jp[j], kEnd));
        s_mark[wv][i] = i == 0 ? 1 : 0;
      }
      __builtin_amdgcn_wave_barrier();
      for (int span = 1; span < kWideWindow; span *= 2) {  // after r steps: the chain's first 2^r positions are marked
        uint16_t nj[kWidePerLane];
#pragma unroll
        for (int j = 0; j < kWidePerLane; j++) {
          const uint32_t i = static_cast<uint32_t>(j) * kWave + lane;
          const uint32_t t = s_jump[wv][cur][i];
          if (s_mark[wv][i] && t < static_cast<uint32_t>(kWideWindow)) s_mark[wv][t] = 1;
          nj[j] = t < static_cast<uint32_t>(kWideWindow) ? s_jump[wv][cur][t] : static_cast<uint16_t>(t);
        }
#pragma unroll
        for (int j = 0; j < kWidePerLane; j++) s_jump[wv][cur ^ 1][static_cast<uint32_t>(j) * kWave + lane] = nj[j];
        cur ^= 1;
        __builtin_amdgcn_wave_barrier();
      }
      // where the chain leaves the window (>= kWideWindow), ends (kEnd) or breaks (a position < kWideWindow)
      const uint32_t land = s_jump[wv][cur][0];
      failed = land < static_cast<uint32_t>(kWideWindow);
      if (!failed) {
#pragma unroll
        for (int j = 0; j < kWidePerLane; j++) {
          const uint32_t i = static_cast<uint32_t>(j) * kWave + lane;
          const bool on = s_mark[wv][i] != 0 && id[j] >= 0;
          const uint64_t m = __ballot(on);
          if (on) out[c + static_cast<uint32_t>(__popcll(m & lt))] = id[j];
          c += static_cast<uint32_t>(__popcll(m));
        }
        pos = land == kEnd ? e : pos + land;
      }
      __builtin_amdgcn_wave_barrier();
    }
    if (failed) {  // the one [UNK] of the word
      if (lane == 0) out[0] = a.unk_id;
      c = 1;
    }
    if (lane == 0) wide_cnt[k] = c | kWideFlag;
  }
}

// the lists of a workgroup's words, one behind the other, at ctmp[position of its first anchor ...]: thread t
// appends words kPer * t ... (cnt[w]: ids of word w, | kWideFlag: all of them in the word's own stretch of emit — those
// are copied by whole waves afterwards, coalesced: a wide word has hundreds of ids, woff keeps where they go)
template <int WORDS, bool WIDE>
__device__ __forceinline__ void assemble_word_lists(const int32_t *__restrict__ emit, const uint32_t *__restrict__ anchors, size_t a0,
                                                    size_t na, const int32_t *stage, const uint32_t *cnt, uint32_t *sm,
                                                    int32_t *__restrict__ ctmp, uint32_t *__restrict__ blk_cnt,
                                                    uint32_t *woff, const uint32_t *wstart) {
  constexpr int kPer = WORDS / kBlock;
  uint32_t mine = 0;
#pragma unroll
  for (int q = 0; q < kPer; q++) mine += cnt[threadIdx.x * kPer + q] & (WIDE ? ~kWideFlag : ~0u);
  uint32_t tot;
  uint32_t ex = block_excl_sum(mine, sm, tot);
  const size_t base = a0 < na ? anchors[a0] : 0;
#pragma unroll
  for (int q = 0; q < kPer; q++) {
    const int wq = threadIdx.x * kPer + q;
    const uint32_t c = cnt[wq] & (WIDE ? ~kWideFlag : ~0u);
    if (WIDE) woff[wq] = ex;
    if (c == 0) continue;
    if (!(WIDE && (cnt[wq] & kWideFlag))) {
      const int32_t *spill = c > static_cast<uint32_t>(kStageIds) ? emit + anchors[a0 + wq] : nullptr;
      for (uint32_t j = 0; j < c; j++) {
        ctmp[base + ex + j] = j < static_cast<uint32_t>(kStageIds) ? stage[j * WORDS + wq] : spill[j];
      }
    }
    ex += c;
  }
  if (WIDE) {
    __syncthreads();
    // two words per trip, all loads (<= 256 ids per word and trip) before the stores: a wave has 8 loads in flight
    // instead of one, and word starts come from LDS (wstart, filled when the word was dealt) — a copy loop with one
    // load per trip behind a global load of the word's start made this kernel slower than the thread-serial copy
    const int lane = lane_id();
    constexpr int kW = 2, kC = 4, kStride = kBlock / kWave;
    for (int w0 = wave_id(); w0 < WORDS; w0 += kW * kStride) {
      uint32_t c[kW], done = 0;
      const int32_t *src[kW];
      int32_t *dst[kW];
#pragma unroll
      for (int u = 0; u < kW; u++) {
        const int wq = w0 + u * kStride;
        const uint32_t cw = wq < WORDS ? cnt[wq] : 0u;  // (wave-uniform)
        c[u] = (cw & kWideFlag) ? (cw & ~kWideFlag) : 0u;
        src[u] = emit + (wq < WORDS ? wstart[wq] : 0u);
        dst[u] = ctmp + base + (wq < WORDS ? woff[wq] : 0u);
      }
      while (done < max(c[0], c[1])) {
        int32_t v[kW][kC];
#pragma unroll
        for (int u = 0; u < kW; u++) {
#pragma unroll
          for (int q = 0; q < kC; q++) {
            const uint32_t j = done + static_cast<uint32_t>(q) * kWave + lane;
            v[u][q] = j < c[u] ? src[u][j] : 0;
          }
        }
#pragma unroll
        for (int u = 0; u < kW; u++) {
#pragma unroll
          for (int q = 0; q < kC; q++) {
            const uint32_t j = done + static_cast<uint32_t>(q) * kWave + lane;
            if (j < c[u]) dst[u][j] = v[u][q];
          }
        }
        done += kC * kWave;
      }
    }
  }
  if (threadIdx.x == 0) blk_cnt[blockIdx.x] = tot;
}

// A lane per word makes every wave wait for its longest word: 1.2 tokens per word on average, 7 in the slowest of
// 64 lanes, and a token is a chain of ~5 dependent loads (the kernel ran 84 % waiting, 20 us per wave).  Here a
// wave owns kWbPerWave consecutive words and deals them out as lanes fall idle: every iteration each busy lane
// takes ONE token step (Step::step), then the idle lanes pick the next words in order (ballot + prefix count,
// no atomics).  Ids are staged per word (first kStageIds in LDS, the rest in the word's own stretch of emit[])
// and leave as one list per workgroup, in word order.
constexpr int kWbPerWave = 256;
constexpr int kWbWords = kWbPerWave * (kBlock / kWave);
struct LinearStep {
  using State = WalkState;
  __device__ static __forceinline__ bool step(const WalkArgs &a, WalkState &s, StagedOut &o) { return walk_step(a, s, o); }
};

template <typename Args, typename Step, bool WIDE = false>
__global__ __launch_bounds__(kBlock) void walk_balanced_kernel(Args a, const uint32_t *__restrict__ anchors,
                                                               const uint32_t *__restrict__ n_anchors_dev, size_t cap,
                                                               int32_t *__restrict__ ctmp, uint32_t *__restrict__ blk_cnt,
                                                               const uint32_t *__restrict__ wide_cnt) {
  // WIDE: stretches of more than kWideMin positions were walked by walk_wide_kernel — their ids sit in
  // the word's own stretch of a.emit, their count (| kWideFlag) in wide_cnt[anchor index]
  __shared__ int32_t stage[kStageIds * kWbWords];
  __shared__ uint32_t cnt[kWbWords];
  __shared__ uint32_t woff[WIDE ? kWbWords : 1], wstart[WIDE ? kWbWords : 1];  // (assemble_word_lists)
  __shared__ uint32_t sm[8];
  const int lane = lane_id(), w = wave_id();
  const size_t na = min(cap, static_cast<size_t>(*n_anchors_dev));
  const size_t a0 = static_cast<size_t>(blockIdx.x) * kWbWords;
  for (int q = threadIdx.x; q < kWbWords; q += kBlock) cnt[q] = 0;
  __syncthreads();
  const int wbase = w * kWbPerWave;
  int next = kWave;  // next word of the wave (relative to wbase) that no lane has taken
  int widx = wbase + lane;
  bool active = a0 + static_cast<size_t>(widx) < na;
  typename Step::State s{0, 0};
  StagedOut o{stage + widx, a.emit, 0, 0, kWbWords};
  auto taken_wide = [&](int wd, uint32_t start) {  // true: the word is not this lane's to walk
    if (!WIDE) return false;
    const size_t k = a0 + static_cast<size_t>(wd);
    const uint32_t hi = k + 1 < static_cast<size_t>(*n_anchors_dev) ? anchors[k + 1] : static_cast<uint32_t>(a.n_text);
    if (hi - start <= kWideMin) return false;
    cnt[wd] = wide_cnt[k];
    wstart[wd] = start;
    return true;
  };
  if (active) {
    const uint32_t start = anchors[a0 + widx];
    s = typename Step::State{start, start};
    o.spill = a.emit + start;
    if (taken_wide(widx, start)) active = false;
  }
  for (;;) {
    if (active) {
      const bool done = s.p >= a.n_text || Step::step(a, s, o);
      if (done) {
        cnt[widx] = o.c;
        active = false;
      }
    }
    const uint64_t idle = __ballot(!active);
    if (next < kWbPerWave) {  // (wave-uniform)
      if (!active) {
        const int cand = next + __popcll(idle & ((1ull << lane) - 1ull));
        if (cand < kWbPerWave && a0 + static_cast<size_t>(wbase + cand) < na) {
          widx = wbase + cand;
          const uint32_t start = anchors[a0 + widx];
          s = typename Step::State{start, start};
          o = StagedOut{stage + widx, a.emit + start, 0, 0, kWbWords};
          active = !taken_wide(widx, start);
        }
      }
      next += __popcll(idle);
    }
    // (a lane that was dealt a wide word is idle again at once: the wave is done when nobody walks AND no word is left)
    if (!__ballot(active) && (!WIDE || next >= kWbPerWave)) break;
  }
  __syncthreads();
  assemble_word_lists<kWbWords, WIDE>(a.emit, anchors, a0, na, stage, cnt, sm, ctmp, blk_cnt, woff, wstart);
}

// The Linear walk, lean.  The kernel above is generic (Fast and Linear steps) and its Linear step is instruction bound,
// not memory bound: 200 vector + 150 scalar instructions per token step (class tests through two 64-bit windows,
// 64-bit positions, the step table's search), with 7 waves per SIMD all wanting the issue port (SQ counters:
// 12 % of a wave's life issuing x 7 waves).  Here the common step is straight: the rank and 16 class bytes from p on
// are loaded together, the step table answers from its bucket entry (scanline.h, step_raw), the class byte of the
// landing position — which carries its word-prefix and anchor bits (anchor_write_kernel) — comes out of the 16 by
// one field extract, positions are 32-bit.  Everything else (no token: [UNK] and roll back; blank runs; a landing
// position beyond the window) goes through the generic walk_finish, which redoes the step from the same state.
template <bool WIDE>
__global__ __launch_bounds__(kBlock) void walk_lean_kernel(WalkArgs a, const uint32_t *__restrict__ anchors,
                                                           const uint32_t *__restrict__ n_anchors_dev, size_t cap,
                                                           int32_t *__restrict__ ctmp, uint32_t *__restrict__ blk_cnt,
                                                           const uint32_t *__restrict__ wide_cnt) {
  __shared__ int32_t stage[kStageIds * kWbWords];
  __shared__ uint32_t cnt[kWbWords];
  __shared__ uint32_t woff[WIDE ? kWbWords : 1], wstart[WIDE ? kWbWords : 1];  // (assemble_word_lists)
  __shared__ uint32_t sm[8];
  const int lane = lane_id(), w = wave_id();
  const size_t na = min(cap, static_cast<size_t>(*n_anchors_dev));
  const size_t a0 = static_cast<size_t>(blockIdx.x) * kWbWords;
  const uint32_t end = static_cast<uint32_t>(a.n_text);
  for (int q = threadIdx.x; q < kWbWords; q += kBlock) cnt[q] = 0;
  __syncthreads();
  const int wbase = w * kWbPerWave;
  const int mine = a0 + static_cast<size_t>(wbase) < na
                       ? static_cast<int>(min(static_cast<size_t>(kWbPerWave), na - a0 - static_cast<size_t>(wbase)))
                       : 0;  // words of this wave that exist
  const bool stop_at_blank = !a.aflags && a.all_hard;  // (walk_finish: the position behind the blanks is an anchor of its own)
  int widx = wbase + lane;
  uint32_t p = 0, since = 0;
  StagedOut o{stage + widx, a.emit, 0, 0, kWbWords};
  // a word is taken: where it starts; false: a wide word (walk_wide_kernel has its ids), nothing to walk
  auto take = [&](int wd) {
    widx = wd;
    const size_t k = a0 + static_cast<size_t>(wd);
    const uint32_t start = anchors[k];
    p = since = start;
    o = StagedOut{stage + wd, a.emit + start, 0, 0, kWbWords};
    if (WIDE) {
      const uint32_t hi = k + 1 < static_cast<size_t>(*n_anchors_dev) ? anchors[k + 1] : end;
      if (hi - start > kWideMin) {
        cnt[wd] = wide_cnt[k];
        wstart[wd] = start;
        return false;
      }
    }
    return true;
  };
  bool active = lane < mine && take(wbase + lane);
  int next = min(kWave, mine);  // next word of the wave (relative to wbase) that no lane has taken
  for (;;) {
    if (active) {
      bool done = true;
      if (p < end) {
        const uint32_t r = rank_of(a.rank[p]);
        uint32_t cw[4];  // class bytes of p .. p + 15 (the array has 16 bytes of slack behind the text)
        __builtin_memcpy(cw, a.cls + p, 16);
        const int32_t raw = step_raw(a.steps, r, (cw[0] & kClsWordPrefix) != 0);
        const int32_t id = step_id(a.steps, raw);
        bool fast = id >= 0 && wp_in_bounds(id < a.n_tokens, kSiteTokenId);
        uint32_t len = 0, p2 = 0, f = 0;
        if (fast) {
          len = static_cast<uint32_t>(step_len(a.steps, raw, a.tok_len));
          p2 = p + len;
          fast = len < 15 || p2 >= end;  // (the landing position and the one behind it inside the window)
        }
        if (fast && p2 < end) {
          const uint32_t d0 = len < 8 ? (len < 4 ? cw[0] : cw[1]) : (len < 12 ? cw[2] : cw[3]);
          f = __builtin_amdgcn_ubfe(d0, (len & 3u) * 8u, 8u);
          if (f & kClsSpace) {
            if (!stop_at_blank) {
              // one blank and the position behind it (the usual case between two words); longer runs: generic path
              const uint32_t l1 = len + 1;
              const uint32_t d1 = l1 < 8 ? (l1 < 4 ? cw[0] : cw[1]) : (l1 < 12 ? cw[2] : cw[3]);
              const uint32_t f1 = __builtin_amdgcn_ubfe(d1, (l1 & 3u) * 8u, 8u);
              if (p2 + 1 < end && !(f1 & kClsSpace)) {
                p2 += 1;
                f = f1;
              } else {
                fast = p2 + 1 >= end;  // (blanks up to the end of the text: done)
                f = kClsAnchor;
                if (fast) p2 = end;
              }
            } else {
              f = kClsAnchor;  // (done: the next anchor is the first position behind the blanks)
            }
          }
        }
        if (fast) {
          o.push(p, id);
          p = p2;
          done = p2 >= end || (f & kClsAnchor);
          if (!done && (f & kClsWordPrefix)) {
            since = p2;
            o.word_start();
          }
        } else {
          WalkState s{p, since};
          const StepWin W = step_window(a, p);
          done = walk_finish(a, s, o, W, raw);
          p = static_cast<uint32_t>(s.p);
          since = static_cast<uint32_t>(s.since);
        }
      }
      if (done) {
        cnt[widx] = o.c;
        active = false;
      }
    }
    // the idle lanes pick the next words in order (ballot + prefix count, no atomics)
    const uint64_t idle = __ballot(!active);
    if (next < mine) {  // (wave-uniform)
      if (!active) {
        const int cand = next + __popcll(idle & ((1ull << lane) - 1ull));
        if (cand < mine) active = take(wbase + cand);
      }
      next = min(next + static_cast<int>(__popcll(idle)), mine);
    }
    // (a lane that was dealt a wide word is idle again at once: the wave is done when nobody walks AND no word is left)
    if (!__ballot(active) && next >= mine) break;
  }
  __syncthreads();
  assemble_word_lists<kWbWords, WIDE>(a.emit, anchors, a0, na, stage, cnt, sm, ctmp, blk_cnt, woff, wstart);
}

// ids[blk_off[b] ...] = the list of workgroup b (of the walk kernel, which took `words` anchors per workgroup)
__global__ __launch_bounds__(kBlock) void emit_gather_kernel(const uint32_t *__restrict__ anchors,
                                                             const uint32_t *__restrict__ n_anchors_dev, size_t cap,
                                                             const int32_t *__restrict__ ctmp,
                                                             const uint32_t *__restrict__ blk_cnt,
                                                             const uint32_t *__restrict__ blk_off,
                                                             int32_t *__restrict__ ids, int words) {
  const size_t k0 = static_cast<size_t>(blockIdx.x) * words;
  if (k0 >= cap || k0 >= *n_anchors_dev) return;
  const size_t base = anchors[k0], off = blk_off[blockIdx.x];
  const uint32_t cnt = blk_cnt[blockIdx.x];
  for (uint32_t j = threadIdx.x; j < cnt; j += kBlock) ids[off + j] = ctmp[base + j];
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
