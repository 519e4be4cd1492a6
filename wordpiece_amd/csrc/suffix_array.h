// suffix_array.h — suffix array + rank + LCP by prefix doubling (replaces libsais_int,
// the inverse-SA loop and calcLcp: linear.cpp:118-149).
//
// Round 0 sorts every suffix by the first 63 bits of its codeword stream (code.h) with an LSD
// radix sort.  Round r >= 1 works on the *active list* only — the slots of groups that are still
// tied — sorting every group by rank[i + depth(group)] and splitting it; singletons (and, in the
// depth-capped mode, groups whose depth already exceeds the longest vocab token) retire.  rank[i]
// is the first SA slot of i's group, so a retired suffix already holds its final rank.  Depths are
// per group: depth(new subgroup) = depth(old group) + depth(group of the second key).
//
// LCP comes out of the same passes: a boundary that appears in round 0 gets its LCP from the two
// keys (complete codewords inside their common bit prefix); a boundary that appears later
// separates two suffixes that share the old group's depth d, so it is d + one direct symbol
// compare from offset d.  Boundaries that never appear (depth-capped mode) keep -1 = "LCP >= the
// depth cap".
#pragma once
#include "decode.h"
#include "primitives.h"
#include "radix_sort.h"

namespace wp {

constexpr int kRrItems = 8;
constexpr int kRrTile = kBlock * kRrItems;  // 2048 list entries per workgroup

// rank[i] = first SA slot of i's group.  The depth of a tied group (symbols its members are known
// to share) is kept per group at gdepth[first slot of the group].
using RankEntry = uint32_t;
__host__ __device__ inline uint32_t rank_of(RankEntry e) { return e; }
constexpr uint32_t kRankUnchanged = 0xffffffffu;  // never a rank: ranks are < n <= 2e9

struct RerankAgg {
  uint32_t last_flag;  // 1 + largest k in the tile that starts a group, 0 if none
  uint32_t n_active;   // entries that stay on the active list
  uint32_t n_heads;    // heads of groups that stay on the active list
};

struct DepthRule {
  uint32_t need;  // a tied group retires once its depth reaches this (depth-capped mode)
  int full;       // 1: only singletons retire (true suffix array)
  // Rounds >= 1 behind a pruned round 0: every group of the list carries the depth ITS tokens need (the longest
  // eligible token that carries the round-0 key, + 1; prune.h) instead of the longest token of the whole vocabulary,
  // handed down from a group to the subgroups it splits into.  gneed_in: by the group ids of the list being split
  // (nullptr: `need` for everyone), gneed_out: by the new group ids.
  const uint32_t *gneed_in;
  uint32_t *gneed_out;
  // final round (trie.h): the second keys are trie nodes, not ranks — every group retires after this split, no depth is
  // looked up, and second_out[slot] receives the entry's second key (ascending inside a group)
  int final_round;
  uint32_t *second_out;
};

// old_gid: group of the entry in the list being split (high word of its key in rounds >= 1)
__device__ __forceinline__ bool rr_stays_active(const DepthRule &rule, uint32_t nd, uint32_t old_gid) {
  if (rule.final_round) return false;
  if (rule.full) return true;
  return nd < (rule.gneed_in ? rule.gneed_in[old_gid] : rule.need);
}

__device__ __forceinline__ void rr_flags(const uint64_t *__restrict__ keys, size_t m, size_t k, bool &flag,
                                         bool &single) {
  const uint64_t me = keys[k];
  flag = (k == 0) || keys[k - 1] != me;
  single = flag && (k + 1 == m || keys[k + 1] != me);
}

// Tile layout for the rerank kernels: wave w of the workgroup owns list entries
// [tile_base + 512 w, +512), visited in 8 rounds of 64 consecutive entries (lane = entry), so every
// global access is a fully coalesced wave access and every ordered scan is a ballot + popcount.
constexpr int kRrRounds = kRrItems;
constexpr int kRrWaveSpan = kWave * kRrRounds;

// Pass 1 over the sorted list: group flags, the new depth of every tied entry (tdep) and the
// per-tile counts.  ROUND0: depth = complete codewords inside the 63-bit key; later rounds:
// depth = old depth + depth of the group the second key came from.
template <bool ROUND0>
__global__ __launch_bounds__(kBlock) void rerank_agg_kernel(const uint64_t *__restrict__ keys,
                                                            const uint32_t *__restrict__ vals, size_t m,
                                                            const uint32_t *__restrict__ adep,
                                                            const RankEntry *__restrict__ rd,
                                                            const uint32_t *__restrict__ gdepth, size_t n,
                                                            const uint8_t *__restrict__ first_len, int uniform_bits,
                                                            DepthRule rule, uint32_t *__restrict__ tdep,
                                                            RerankAgg *__restrict__ agg) {
  __shared__ uint32_t s_na[4], s_nh[4], s_last[4];
  __shared__ uint8_t s_fl[kDecodeTableBytes];
  if (ROUND0 && uniform_bits <= 0) {
    for (int q = threadIdx.x; q < kDecodeTableBytes / 4; q += kBlock) {
      reinterpret_cast<uint32_t *>(s_fl)[q] = reinterpret_cast<const uint32_t *>(first_len)[q];
    }
    __syncthreads();
  }
  const int lane = lane_id(), w = wave_id();
  const size_t wave_base = static_cast<size_t>(blockIdx.x) * kRrTile + static_cast<size_t>(w) * kRrWaveSpan;
  uint32_t na = 0, nh = 0, last = 0;  // last: 1 + list index of the last group head seen, 0 = none
#pragma unroll
  for (int r = 0; r < kRrRounds; r++) {
    const size_t k = wave_base + static_cast<size_t>(r) * kWave + lane;
    bool f = false, sg = true, act = false;
    if (k < m) {
      rr_flags(keys, m, k, f, sg);
      if (!sg) {
        uint32_t nd;
        if (ROUND0) {
          nd = static_cast<uint32_t>(count_key_symbols(keys[k], kKeyBits, s_fl, uniform_bits));
        } else {
          // the second key is 1 + rank of suffix vals[k] + d (0: past the end), i.e. 1 + the first slot of
          // that suffix's group, where its depth is kept: no need to go through rank[] again
          if (rule.final_round) {
            nd = 0u;
          } else {
            const uint32_t d = adep[k];
            const uint32_t r2 = static_cast<uint32_t>(keys[k]);
            const uint32_t dj = r2 ? gdepth[r2 - 1u] : 0u;
            nd = min(d + dj, 0x7fffffffu);
          }
        }
        tdep[k] = nd;
        act = rr_stays_active(rule, nd, ROUND0 ? 0u : static_cast<uint32_t>(keys[k] >> 32));
      }
    }
    const uint64_t bf = __ballot(f), ba = __ballot(act), bh = __ballot(f && act);
    na += __popcll(ba);
    nh += __popcll(bh);
    if (bf) last = static_cast<uint32_t>(wave_base + static_cast<size_t>(r) * kWave + (63 - __clzll(static_cast<long long>(bf))) + 1);
  }
  if (lane == 0) {
    s_na[w] = na;
    s_nh[w] = nh;
    s_last[w] = last;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    RerankAgg a;
    a.n_active = s_na[0] + s_na[1] + s_na[2] + s_na[3];
    a.n_heads = s_nh[0] + s_nh[1] + s_nh[2] + s_nh[3];
    a.last_flag = max(max(s_last[0], s_last[1]), max(s_last[2], s_last[3]));
    agg[blockIdx.x] = a;
  }
}

// The apply pass needs, per tile, the exclusive prefix of the tile aggregates (sums of the two
// counts, running max of last_flag).  Two small multi-workgroup kernels instead of a single-workgroup
// scan (0.23 ms for 48.9 k tiles): one reduces chunks of kRrChunk tiles, one turns the tile
// aggregates into exclusive prefixes chunk by chunk.
constexpr int kRrChunk = kBlock;

__device__ __forceinline__ RerankAgg block_reduce_agg(RerankAgg a, uint32_t (*sm)[4]) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) {
    a.n_active += __shfl_xor(a.n_active, d, kWave);
    a.n_heads += __shfl_xor(a.n_heads, d, kWave);
    a.last_flag = max(a.last_flag, static_cast<uint32_t>(__shfl_xor(a.last_flag, d, kWave)));
  }
  const int w = wave_id();
  __syncthreads();
  if (lane_id() == 0) {
    sm[0][w] = a.n_active;
    sm[1][w] = a.n_heads;
    sm[2][w] = a.last_flag;
  }
  __syncthreads();
  RerankAgg o;
  o.n_active = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
  o.n_heads = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
  o.last_flag = max(max(sm[2][0], sm[2][1]), max(sm[2][2], sm[2][3]));
  return o;
}

__global__ __launch_bounds__(kBlock) void rerank_chunk_kernel(const RerankAgg *__restrict__ agg, unsigned tiles,
                                                              RerankAgg *__restrict__ chunk_agg) {
  __shared__ uint32_t sm[3][4];
  const unsigned t = blockIdx.x * kRrChunk + threadIdx.x;
  RerankAgg a = {0, 0, 0};
  if (t < tiles) a = agg[t];
  a = block_reduce_agg(a, sm);
  if (threadIdx.x == 0) chunk_agg[blockIdx.x] = a;
}

// agg[t] <- exclusive prefix of tile t: every workgroup reduces the chunks before its own (one per
// thread and step) and scans the kRrChunk tile aggregates of its chunk
__global__ __launch_bounds__(kBlock) void rerank_prefix_kernel(RerankAgg *__restrict__ agg,
                                                               const RerankAgg *__restrict__ chunk_agg,
                                                               unsigned tiles) {
  __shared__ uint32_t sm[3][4];
  __shared__ uint32_t wa[4], wh[4], wm[4];
  const unsigned chunk = blockIdx.x;
  RerankAgg a = {0, 0, 0};
  for (unsigned c = threadIdx.x; c < chunk; c += kBlock) {
    const RerankAgg v = chunk_agg[c];
    a.n_active += v.n_active;
    a.n_heads += v.n_heads;
    a.last_flag = max(a.last_flag, v.last_flag);
  }
  const RerankAgg before = block_reduce_agg(a, sm);
  const unsigned t = chunk * kRrChunk + threadIdx.x;
  RerankAgg mine = {0, 0, 0};
  if (t < tiles) mine = agg[t];
  const int lane = lane_id(), w = wave_id();
  uint32_t ia = mine.n_active, ih = mine.n_heads, im = mine.last_flag;
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    const uint32_t ta = __shfl_up(ia, d, kWave), th = __shfl_up(ih, d, kWave), tm = __shfl_up(im, d, kWave);
    if (lane >= d) {
      ia += ta;
      ih += th;
      im = max(im, tm);
    }
  }
  if (lane == kWave - 1) {
    wa[w] = ia;
    wh[w] = ih;
    wm[w] = im;
  }
  __syncthreads();
  RerankAgg o = before;
  for (int q = 0; q < w; q++) {
    o.n_active += wa[q];
    o.n_heads += wh[q];
    o.last_flag = max(o.last_flag, wm[q]);
  }
  const uint32_t pm = __shfl_up(im, 1, kWave);
  o.n_active += ia - mine.n_active;
  o.n_heads += ih - mine.n_heads;
  if (lane > 0) o.last_flag = max(o.last_flag, pm);
  if (t < tiles) agg[t] = o;
}

// number of equal symbols at a+t, b+t for t < maxlen (positions >= n never match)
template <typename SymT>
__device__ __forceinline__ int32_t lcp_compare(const SymT *__restrict__ sym, size_t n, size_t a, size_t b,
                                               int32_t maxlen) {
  const size_t hi = a > b ? a : b;
  if (hi >= n) return 0;
  const int32_t lim = static_cast<int32_t>(min(static_cast<size_t>(maxlen), n - hi));
  int32_t t = 0;
  if (sizeof(SymT) == 1) {
    // 8 symbols per (unaligned) 64-bit load; the symbol buffer is padded by 16 bytes past n
    const uint8_t *pa = reinterpret_cast<const uint8_t *>(sym) + a, *pb = reinterpret_cast<const uint8_t *>(sym) + b;
    while (t < lim) {
      uint64_t wa, wb;
      __builtin_memcpy(&wa, pa + t, 8);
      __builtin_memcpy(&wb, pb + t, 8);
      const uint64_t x = wa ^ wb;
      if (x) {
        t += (__ffsll(static_cast<long long>(x)) - 1) >> 3;
        break;
      }
      t += 8;
    }
    return min(t, lim);
  }
  while (t < lim && sym[a + t] == sym[b + t]) t++;
  return t;
}

// xcd != 0: workgroups of one XCD take a contiguous range of the list (used after the list has
// been partitioned by destination, so that the stores of one XCD fall into one region and meet in
// its L2).  One entry per thread on purpose: the fewer entries an XCD has in flight, the narrower
// the slice of the partitioned list it is working on and the better its stores merge (measured for
// 1e8 entries: 0.79 ms with 1 entry per thread, 1.0 / 1.15 / 1.30 ms with 2 / 4 / 8).
#ifndef WP_SP_ITEMS
#define WP_SP_ITEMS 1
#endif
constexpr int kSpItems = WP_SP_ITEMS;
constexpr int kSpTile = kBlock * kSpItems;
__global__ __launch_bounds__(kBlock) void scatter_pairs_kernel(const uint32_t *__restrict__ dst,
                                                               const RankEntry *__restrict__ val, size_t m,
                                                               RankEntry *__restrict__ out, size_t out_n, int xcd) {
  unsigned b = blockIdx.x;
  if (xcd) {
    const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8;
    b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
  }
  const size_t base = static_cast<size_t>(b) * kSpTile + threadIdx.x;
  uint32_t d[kSpItems];
  RankEntry v[kSpItems];
#pragma unroll
  for (int j = 0; j < kSpItems; j++) {
    const size_t k = base + static_cast<size_t>(j) * kBlock;
    v[j] = k < m ? (val ? val[k] : static_cast<RankEntry>(k)) : kRankUnchanged;  // val == nullptr: the entry's own index
    d[j] = k < m ? dst[k] : 0u;
  }
#pragma unroll
  for (int j = 0; j < kSpItems; j++) {
    if (v[j] != kRankUnchanged && wp_in_bounds(d[j] < out_n, kSiteRankStore)) out[d[j]] = v[j];
  }
}

// The rank store of round 0 is a permutation (every slot of the rank table is written exactly once).  After
// the list has been partitioned by the destination bits above kWinBits, the entries of window w are exactly
// list[w << kWinBits, (w + 1) << kWinBits): one workgroup places them in an LDS image of the window and
// writes the window with full-width stores.  Measured (profiles/yardstick/scatter_probe.hip, 1.3e8 entries):
// 0.29 ms against 0.77 ms for the XCD-aware 4-byte scatter into 2 MB windows (request-rate bound: 0.64 ms
// even into 64 KB windows) and 3.7 ms unpartitioned.
constexpr int kWinBits = 15;
constexpr int kWinThreads = 1024;
constexpr size_t kWinLdsBytes = sizeof(RankEntry) << kWinBits;
__global__ __launch_bounds__(kWinThreads) void window_store_kernel(const uint32_t *__restrict__ dst,
                                                                   const RankEntry *__restrict__ val, size_t m,
                                                                   RankEntry *__restrict__ out) {
  extern __shared__ RankEntry win[];
  const size_t w0 = static_cast<size_t>(blockIdx.x) << kWinBits;
  const uint32_t cnt = static_cast<uint32_t>(min(static_cast<size_t>(1) << kWinBits, m - w0));
  const uint32_t lo = static_cast<uint32_t>(w0);
  // (dst, val, out: 16-byte aligned bases; w0 is a multiple of 4.  The range check stays on in every build:
  // a list that is not the partitioned permutation must not write outside the window image)
  auto place = [&](uint32_t d, RankEntry v) {
    const bool ok = d - lo < cnt;
    wp_in_bounds(ok, kSiteRankStore);
    if (ok) win[d - lo] = v;
  };
  for (uint32_t k = threadIdx.x * 4; k < cnt; k += kWinThreads * 4) {
    if (k + 4 <= cnt) {
      const uint4 d = *reinterpret_cast<const uint4 *>(dst + w0 + k);
      const uint4 v = *reinterpret_cast<const uint4 *>(val + w0 + k);
      place(d.x, v.x);
      place(d.y, v.y);
      place(d.z, v.z);
      place(d.w, v.w);
    } else {
      for (uint32_t j = k; j < cnt; j++) place(dst[w0 + j], val[w0 + j]);
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x * 4; k < cnt; k += kWinThreads * 4) {
    if (k + 4 <= cnt) {
      *reinterpret_cast<uint4 *>(out + w0 + k) = *reinterpret_cast<const uint4 *>(win + k);
    } else {
      for (uint32_t j = k; j < cnt; j++) out[w0 + j] = win[j];
    }
  }
}

// ---- the split of one round ------------------------------------------------------------------------
// ROUND0: list == all slots (slot k == k), keys are the packed codeword streams.  Later rounds:
// keys = (group id << 32 | second key), adep = depth of the entry's (old) group.
// A workgroup handles its tile in two halves with the tile prefix in between.  Everything the second
// half needs from the keys (the key itself, the common bit prefix with the previous key, "same old
// group as the previous entry") is taken in the first half and kept in registers / ballots: with
// ~190 tiles in flight per XCD the 16 KB of keys of a tile do not survive in the 4 MB L2 until the
// tile prefix is known, and a second read came from HBM.
struct RrTile {
  uint64_t bfs[kRrRounds], bas[kRrRounds], bss[kRrRounds], bos[kRrRounds];  // head / active / single / same old group
  uint64_t mes[kRrRounds];                                                   // the entry's key
  uint32_t nds[kRrRounds];  // new depth; ROUND0: | bits shared with the previous key << 16 (for heads)
  uint32_t na, nh, last;    // wave totals: active, active heads, 1 + last head
};

// OWN_DEPTH: the new depth is computed here (single-pass kernel) instead of read from tdep
template <bool ROUND0, bool OWN_DEPTH>
__device__ __forceinline__ void rr_first_half(RrTile &T, const uint64_t *__restrict__ keys, size_t m, size_t wave_base,
                                              const uint32_t *__restrict__ tdep, const uint32_t *__restrict__ vals,
                                              const uint32_t *__restrict__ adep, const RankEntry *__restrict__ rd,
                                              const uint32_t *__restrict__ gdepth_in, size_t n, const uint8_t *s_fl,
                                              int uniform_bits, DepthRule rule) {
  const int lane = lane_id();
  T.na = T.nh = T.last = 0;
#pragma unroll
  for (int r = 0; r < kRrRounds; r++) {
    const size_t k = wave_base + static_cast<size_t>(r) * kWave + lane;
    bool f = false, sg = false, act = false, same_old = false;
    uint64_t me = 0;
    uint32_t nd = 0;
    if (k < m) {
      me = keys[k];
      const uint64_t prev = k > 0 ? keys[k - 1] : ~me;
      const uint64_t next = k + 1 < m ? keys[k + 1] : ~me;
      f = prev != me;
      sg = f && next != me;
      if (!sg) {
        if (!OWN_DEPTH) {
          nd = tdep[k];
        } else if (ROUND0) {
          nd = static_cast<uint32_t>(count_key_symbols(me, kKeyBits, s_fl, uniform_bits));
        } else if (rule.final_round) {
          nd = 0u;
        } else {
          const uint32_t d = adep[k];
          const uint32_t r2 = static_cast<uint32_t>(me);
          const uint32_t dj = r2 ? gdepth_in[r2 - 1u] : 0u;
          nd = min(d + dj, 0x7fffffffu);
        }
        act = rr_stays_active(rule, nd, ROUND0 ? 0u : static_cast<uint32_t>(me >> 32));
      }
      if (ROUND0) {
        nd = min(nd, 0xffffu);
        if (f && k > 0) nd |= static_cast<uint32_t>(__clzll(static_cast<long long>(me ^ prev))) << 16;
      } else {
        same_old = k > 0 && (me >> 32) == (prev >> 32);
      }
    }
    T.mes[r] = me;
    T.nds[r] = nd;
    T.bfs[r] = __ballot(f);
    T.bas[r] = __ballot(act);
    T.bss[r] = __ballot(sg);
    T.bos[r] = __ballot(same_old);
    T.na += __popcll(T.bas[r]);
    T.nh += __popcll(T.bfs[r] & T.bas[r]);
    if (T.bfs[r]) {
      T.last = static_cast<uint32_t>(wave_base + static_cast<size_t>(r) * kWave + (63 - __clzll(static_cast<long long>(T.bfs[r]))) + 1);
    }
  }
}

// ea / eh / head1: entries that stay active, their heads, 1 + last head — all before this wave.
// gd != nullptr: park the new group depths in gd[] instead of storing them to gdepth (single-pass
// kernel, rounds >= 1, see there).
template <typename SymT, bool ROUND0>
__device__ __forceinline__ void rr_second_half(const RrTile &T, const uint64_t *__restrict__ keys, size_t m,
                                               size_t wave_base, uint32_t ea, uint32_t eh, uint32_t head1,
                                               const uint32_t *__restrict__ vals, const uint32_t *__restrict__ slots,
                                               const uint32_t *__restrict__ adep, const SymT *__restrict__ sym, size_t n,
                                               const uint8_t *s_fl, int uniform_bits, uint32_t *__restrict__ sa,
                                               RankEntry *__restrict__ hd, int32_t *__restrict__ lcp,
                                               uint32_t *__restrict__ nslots, uint32_t *__restrict__ nvals,
                                               uint32_t *__restrict__ ngid, uint32_t *__restrict__ ndep,
                                               uint32_t *__restrict__ ghead, uint32_t *__restrict__ gdepth,
                                               uint32_t *__restrict__ gd, const DepthRule &rule) {
  const int lane = lane_id();
  const uint64_t lt = (1ull << lane) - 1ull, le = lt | (1ull << lane);
  // rounds >= 1: does the group of the carried head continue the old group of the entry before it?
  // (then it is not the first subgroup of its old group and its rank changes)
  bool chg1 = true;
  if (!ROUND0) {
    chg1 = false;
    if (head1 > 1) chg1 = (keys[head1 - 1] >> 32) == (keys[head1 - 2] >> 32);
  }
#pragma unroll
  for (int r = 0; r < kRrRounds; r++) {
    const size_t round_base = wave_base + static_cast<size_t>(r) * kWave;
    const size_t k = round_base + lane;
    const uint64_t bf = T.bfs[r], ba = T.bas[r], bh = T.bfs[r] & T.bas[r];
    if (k < m) {
      const uint64_t mine = bf & le;
      const int hl = 63 - __clzll(static_cast<long long>(mine));  // lane of my head, if it is in this round
      const size_t head = mine ? round_base + hl : (head1 ? static_cast<size_t>(head1) - 1 : 0);
      const bool f = (bf >> lane) & 1ull;
      const bool act = (ba >> lane) & 1ull;
      const bool single = (T.bss[r] >> lane) & 1ull;
      const uint32_t v = vals[k];
      const uint32_t x = ROUND0 ? static_cast<uint32_t>(k) : slots[k];
      const uint32_t head_slot = ROUND0 ? static_cast<uint32_t>(head) : slots[head];
      // the suffix array itself is only kept for debug fetches / the Kasai kernel, and (text-only layout,
      // rounds >= 1) for the slots of the groups whose long tokens are located in it afterwards
      if (sa) sa[x] = v;
      if (!ROUND0 && rule.second_out) rule.second_out[x] = static_cast<uint32_t>(T.mes[r]) - 1u;
      // new rank entry of suffix v, scattered to the rank table afterwards.  In rounds >= 1 the
      // first subgroup of an old group keeps its rank (its head is the old head): left unchanged.
      // (final round behind a round 0 whose ranks are the suffixes' own slots, not group heads: every entry is stored)
      bool changed = true;
      if (!ROUND0 && !rule.final_round) changed = mine ? ((T.bos[r] >> hl) & 1ull) != 0 : chg1;
      const uint32_t nd = single ? 0u : (ROUND0 ? (T.nds[r] & 0xffffu) : T.nds[r]);
      hd[k] = changed ? head_slot : kRankUnchanged;
      if (gd) {
        gd[k] = (f && !single) ? nd : kRankUnchanged;
      } else if (f && !single && !rule.final_round) {
        gdepth[x] = nd;  // x is the first slot of this (still tied) group
      }
      if (!lcp) {
        // (text-only layout: nothing reads the LCPs, and the symbol compare below is two gathers per boundary)
      } else if (ROUND0) {
        if (k > 0) {
          int32_t l = -1;
          if (f) l = count_key_symbols(T.mes[r], static_cast<int>(T.nds[r] >> 16) - (64 - kKeyBits), s_fl, uniform_bits);
          lcp[x - 1] = l;
        }
      } else if (f && ((T.bos[r] >> lane) & 1ull)) {
        // a new boundary inside an old group (x-1 is the previous list entry's slot): the two
        // suffixes share the old group's depth and then differ within the second keys' reach
        const uint32_t d = adep[k];
        lcp[x - 1] = static_cast<int32_t>(d)
                     + lcp_compare(sym, n, static_cast<size_t>(vals[k - 1]) + d, static_cast<size_t>(v) + d,
                                   0x7fffffff);
      }
      if (act) {
        const uint32_t pos = ea + __popcll(ba & lt);
        const uint32_t g = eh + __popcll(bh & le) - 1;
        nslots[pos] = x;
        nvals[pos] = v;
        ngid[pos] = g;
        ndep[pos] = nd;
        if (f) {
          ghead[g] = pos;
          if (!ROUND0 && rule.gneed_in) rule.gneed_out[g] = rule.gneed_in[static_cast<uint32_t>(keys[k] >> 32)];
        }
      }
    }
    ea += __popcll(ba);
    eh += __popcll(bh);
    if (bf) {
      const int hl = 63 - __clzll(static_cast<long long>(bf));
      head1 = static_cast<uint32_t>(round_base + hl + 1);
      if (!ROUND0) chg1 = (T.bos[r] >> hl) & 1ull;
    }
  }
}

// Pass 2 of the three-kernel form: tile prefix from the aggregates of pass 1, then the split.
template <typename SymT, bool ROUND0>
__global__ __launch_bounds__(kBlock) void rerank_apply_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ slots,
    const uint32_t *__restrict__ adep, const uint32_t *__restrict__ tdep, size_t m,
    const RerankAgg *__restrict__ agg, const SymT *__restrict__ sym, size_t n,
    const uint8_t *__restrict__ first_len, int uniform_bits, DepthRule rule, uint32_t *__restrict__ sa,
    RankEntry *__restrict__ hd, int32_t *__restrict__ lcp, uint32_t *__restrict__ nslots,
    uint32_t *__restrict__ nvals, uint32_t *__restrict__ ngid, uint32_t *__restrict__ ndep,
    uint32_t *__restrict__ ghead, uint32_t *__restrict__ gdepth, uint32_t *__restrict__ totals) {
  __shared__ uint32_t s_na[4], s_nh[4], s_last[4];
  __shared__ uint8_t s_fl[kDecodeTableBytes];
  if (ROUND0 && uniform_bits <= 0) {
    for (int q = threadIdx.x; q < kDecodeTableBytes / 4; q += kBlock) {
      reinterpret_cast<uint32_t *>(s_fl)[q] = reinterpret_cast<const uint32_t *>(first_len)[q];
    }
  }
  const int lane = lane_id(), w = wave_id();
  const size_t wave_base = static_cast<size_t>(blockIdx.x) * kRrTile + static_cast<size_t>(w) * kRrWaveSpan;
  RrTile T;
  rr_first_half<ROUND0, false>(T, keys, m, wave_base, tdep, nullptr, nullptr, nullptr, nullptr, n, s_fl, uniform_bits,
                               rule);
  if (lane == 0) {
    s_na[w] = T.na;
    s_nh[w] = T.nh;
    s_last[w] = T.last;
  }
  const RerankAgg pre = agg[blockIdx.x];  // exclusive prefix (rerank_prefix_kernel)
  __syncthreads();
  if (blockIdx.x + 1 == gridDim.x && threadIdx.x == 0) {  // the last tile knows the totals of the round
    const uint32_t ta = pre.n_active + s_na[0] + s_na[1] + s_na[2] + s_na[3];
    const uint32_t th = pre.n_heads + s_nh[0] + s_nh[1] + s_nh[2] + s_nh[3];
    totals[0] = ta;
    totals[1] = th;
    ghead[th] = ta;  // sentinel: group g of the next list is [ghead[g], ghead[g+1])
  }
  uint32_t ea = pre.n_active, eh = pre.n_heads, head1 = pre.last_flag;  // head1: 1-based list index
  for (int i = 0; i < w; i++) {
    ea += s_na[i];
    eh += s_nh[i];
    head1 = max(head1, s_last[i]);
  }
  rr_second_half<SymT, ROUND0>(T, keys, m, wave_base, ea, eh, head1, vals, slots, adep, sym, n, s_fl, uniform_bits, sa,
                               hd, lcp, nslots, nvals, ngid, ndep, ghead, gdepth, nullptr, rule);
}

// full-depth mode with 32-bit round-0 keys: the count / prefix / apply kernels below take 64-bit keys
__global__ __launch_bounds__(kBlock) void widen_keys_kernel(const Key0 *__restrict__ in, uint64_t *__restrict__ out,
                                                            size_t n) {
  for (size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<size_t>(gridDim.x) * kBlock) {
    out[i] = static_cast<uint64_t>(in[i]);
  }
}

// ---- round 0 in the depth-capped mode: ranks and LCPs only ---------------------------------------------
// With the pruning of prune.h the next active list is written by the kernel that finds the needed
// groups, so round 0 needs no counting pass, no tile prefix and no compaction: one streaming kernel turns
// the sorted keys into rank entries (the first slot of every suffix's group), the LCP of every boundary
// (complete codewords inside the common bits of two neighbouring keys; -1 = "same key") and the depth
// of every tied group (kept at gdepth[first slot] for the rounds that double through it).
// The one dependency across tiles — where does the group that reaches into this wave's first entry
// start — is answered by the keys themselves: a look at the 64 entries in front of the wave, and a
// binary search in the sorted keys for the rare group that is longer than that.
// Layout: a lane owns kR0Vec consecutive entries per step (16-byte loads and stores of keys, ranks and
// LCPs; the 4-byte-per-lane form of the same kernel ran at 2.3 TB/s), a wave kR0Steps steps of 64 x kR0Vec
// entries; neighbours across lanes come by shuffles, the head carried across lanes by one ballot + one
// variable-lane shuffle.  lcp[k] (boundary between slots k and k+1) is written by the owner of slot k.
constexpr int kR0Vec = 16 / static_cast<int>(sizeof(Key0)) * 2;  // 8 entries (u32 keys: 2 x uint4) or 4 (u64 keys: 2 x 16 B)
constexpr int kR0Steps = 2;
constexpr int kR0WaveSpan = kWave * kR0Vec * kR0Steps;
constexpr int kR0Spans = 4;  // spans a wave takes one after the other: the decode table is staged once per workgroup
constexpr int kR0Tile = (kBlock / kWave) * kR0WaveSpan * kR0Spans;
// LCP = false: no LCP array (the text-only layout takes the tokens' reach from their ranges in the sorted keys,
// nothing reads the LCPs): 4 bytes read and 4 written per entry.
template <bool LCP>
__global__ __launch_bounds__(kBlock) void round0_rank_kernel(const Key0 *__restrict__ keys,
                                                             const uint32_t *__restrict__ vals, size_t n,
                                                             const uint8_t *__restrict__ first_len, int uniform_bits,
                                                             uint32_t *__restrict__ sa_dbg, RankEntry *__restrict__ hd,
                                                             int32_t *__restrict__ lcp, uint32_t *__restrict__ gdepth) {
  __shared__ uint8_t s_fl[kDecodeTableBytes];
  // Heads of tied groups need their key's symbol count (a table walk of ~14 steps).  Done where the head is found,
  // every wave ran that walk 16 times per span with a few lanes active (137 VALU instructions per entry, the
  // kernel was VALU bound); instead the heads of a step are listed in LDS and the wave walks the list densely.
  // (a tied head is followed by an entry that is no head: at most half of a step's entries)
  constexpr int kHeadCap = kWave * kR0Vec / 2;
  __shared__ uint32_t s_hpos[kBlock / kWave][kHeadCap];
  __shared__ Key0 s_hkey[kBlock / kWave][kHeadCap];
  const int lane = lane_id(), w = wave_id();
  volatile uint32_t *hpos = s_hpos[w];
  volatile Key0 *hkey = s_hkey[w];
  if (uniform_bits <= 0) {
    for (int q = threadIdx.x; q < kDecodeTableBytes / 4; q += kBlock) {
      reinterpret_cast<uint32_t *>(s_fl)[q] = reinterpret_cast<const uint32_t *>(first_len)[q];
    }
  }
  __syncthreads();
  // (keys stay in registers of their own width; entries past the end are masked by index, never by value)
  auto wide = [](Key0 k) { return static_cast<uint64_t>(k); };
  for (int sp = 0; sp < kR0Spans; sp++) {
    const size_t wave_base = static_cast<size_t>(blockIdx.x) * kR0Tile +
                             (static_cast<size_t>(sp) * (kBlock / kWave) + w) * kR0WaveSpan;
    if (wave_base >= n) return;  // (wave-uniform; no barrier behind this point)
    // all loads of the span are issued before anything waits
    Key0 me[kR0Steps][kR0Vec];
#pragma unroll
    for (int t = 0; t < kR0Steps; t++) {
      const size_t k0 = wave_base + (static_cast<size_t>(t) * kWave + lane) * kR0Vec;
      if (k0 + kR0Vec <= n) {  // (the key buffer is 256-byte aligned and k0 a multiple of kR0Vec: aligned 16-byte loads)
        const uint4 *src = reinterpret_cast<const uint4 *>(keys + k0);
        uint4 a = src[0], b = src[1];
        __builtin_memcpy(&me[t][0], &a, 16);
        __builtin_memcpy(reinterpret_cast<char *>(&me[t][0]) + 16, &b, 16);
      } else {
#pragma unroll
        for (int j = 0; j < kR0Vec; j++) me[t][j] = k0 + j < n ? keys[k0 + j] : static_cast<Key0>(0);
      }
    }
    const bool have_front = wave_base >= 1 + static_cast<size_t>(lane);
    const Key0 front = have_front ? keys[wave_base - 1 - lane] : static_cast<Key0>(0);  // lane l: the key l + 1 entries in front
    const size_t after_idx = wave_base + kR0WaveSpan;
    const Key0 after = after_idx < n ? keys[after_idx] : static_cast<Key0>(0);  // (one broadcast load)
    size_t carry = wave_base;  // head of the group of the entries in front of the first head seen by this wave
    {
      const Key0 me0 = __shfl(me[0][0], 0, kWave);
      const uint64_t neq = ~__ballot(have_front && front == me0);
      if (neq & 1ull) {
        carry = wave_base;  // the entry in front has another key (or there is none): the wave starts a group
      } else if (neq) {
        carry = wave_base - static_cast<size_t>(__ffsll(static_cast<long long>(neq)) - 1);
      } else {
        // a group of more than 64 entries reaches in.  Its first entry: ONE load with the lanes 64, 128, 256, ...
        // entries further back (27 probes reach past 2^32) brackets it between two probes, then the wave search
        // inside the bracket — where a binary search from the front of the array took 27 dependent loads and made
        // this kernel a latency chain (most spans start inside a large group), and 64 evenly spaced probes per
        // step pulled in 8 KB of foreign cache lines per 4 KB span
        size_t hi = wave_base - kWave, lo = 0;  // keys[hi] == me0
        {
          const size_t back = static_cast<size_t>(kWave) << (lane < 40 ? lane : 40);
          const bool valid = lane < 40 && hi >= back;
          const bool eq = valid && keys[hi - back] == me0;
          const uint64_t nm = ~__ballot(eq);  // (never 0: the high lanes are not valid)
          const int t = __ffsll(static_cast<long long>(nm)) - 1;  // the nearest probe with another key, or in front of the array
          const size_t back_t = static_cast<size_t>(kWave) << t, back_in = t ? static_cast<size_t>(kWave) << (t - 1) : 0;
          lo = hi >= back_t ? hi - back_t + 1 : 0;
          hi -= back_in;
        }
        carry = wave_key_lower_bound(keys, lo, hi, static_cast<uint64_t>(me0), 0);
      }
    }
    Key0 prev_last = __shfl(front, 0, kWave);  // key of the entry in front of the current step
#pragma unroll
    for (int t = 0; t < kR0Steps; t++) {
      const size_t k0 = wave_base + (static_cast<size_t>(t) * kWave + lane) * kR0Vec;
      const Key0 up = __shfl_up(me[t][kR0Vec - 1], 1, kWave), dn = __shfl_down(me[t][0], 1, kWave);
      const Key0 next_first = t + 1 < kR0Steps ? __shfl(me[t + 1 < kR0Steps ? t + 1 : t][0], 0, kWave) : after;
      const Key0 prevk = lane == 0 ? prev_last : up;
      const Key0 nextk = lane == kWave - 1 ? next_first : dn;
      // flags of the lane's entries; the last head inside the lane
      bool f[kR0Vec];
      int last = -1;
#pragma unroll
      for (int j = 0; j < kR0Vec; j++) {
        const Key0 p = j == 0 ? prevk : me[t][j - 1];
        f[j] = k0 + j < n && (k0 + j == 0 || p != me[t][j]);
        if (f[j]) last = j;
      }
      // head carried into the lane: the last head of the nearest lower lane that has one, else the wave's carry
      const uint64_t bh = __ballot(last >= 0);
      const uint64_t lower = bh & ((1ull << lane) - 1ull);
      const int src_lane = lower ? 63 - __clzll(static_cast<long long>(lower)) : 0;
      const uint32_t my_last_pos = static_cast<uint32_t>(k0 + (last >= 0 ? last : 0));
      const uint32_t from_lane = __shfl(my_last_pos, src_lane, kWave);
      size_t head = lower ? static_cast<size_t>(from_lane) : carry;
      uint32_t hv[kR0Vec];
      int32_t lv[kR0Vec];
      uint32_t tied = 0;  // the lane's heads of tied groups
#pragma unroll
      for (int j = 0; j < kR0Vec; j++) {
        if (f[j]) head = k0 + j;
        hv[j] = static_cast<uint32_t>(head);
        // boundary between slot k0 + j and the next one
        const Key0 nx = j + 1 < kR0Vec ? me[t][j + 1 < kR0Vec ? j + 1 : j] : nextk;
        const bool has_next = k0 + j + 1 < n;
        int32_t l = -1;
        if (LCP && has_next && nx != me[t][j]) {
          l = count_key_symbols(wide(nx), __clzll(static_cast<long long>(wide(nx) ^ wide(me[t][j]))) - (64 - kKeyBits), s_fl,
                                uniform_bits);
        }
        lv[j] = l;
        if (f[j] && has_next && nx == me[t][j]) tied |= 1u << j;  // (the next entry has the same key)
      }
      {  // depth of the tied groups that start in this step
        const uint32_t cnt = __popc(tied);
        const uint32_t incl = wave_incl_sum(cnt);
        const uint32_t total = __shfl(incl, kWave - 1, kWave);
        uint32_t o = incl - cnt;
#pragma unroll
        for (int j = 0; j < kR0Vec; j++) {
          if ((tied >> j) & 1u) {
            hpos[o] = static_cast<uint32_t>(k0 + j);
            hkey[o] = me[t][j];
            o++;
          }
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q = lane; q < total; q += kWave) {
          gdepth[hpos[q]] = static_cast<uint32_t>(count_key_symbols(wide(hkey[q]), kKeyBits, s_fl, uniform_bits));
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (k0 + kR0Vec <= n) {
        uint4 *hdst = reinterpret_cast<uint4 *>(hd + k0), *ldst = reinterpret_cast<uint4 *>(lcp + k0);
#pragma unroll
        for (int q = 0; q < kR0Vec / 4; q++) {
          hdst[q] = make_uint4(hv[4 * q], hv[4 * q + 1], hv[4 * q + 2], hv[4 * q + 3]);
          if (LCP) {
            ldst[q] = make_uint4(static_cast<uint32_t>(lv[4 * q]), static_cast<uint32_t>(lv[4 * q + 1]),
                                 static_cast<uint32_t>(lv[4 * q + 2]), static_cast<uint32_t>(lv[4 * q + 3]));
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < kR0Vec; j++) {
          if (k0 + j < n) {
            hd[k0 + j] = hv[j];
            if (LCP && k0 + j + 1 < n) lcp[k0 + j] = lv[j];
          }
        }
      }
      if (sa_dbg) {
#pragma unroll
        for (int j = 0; j < kR0Vec; j++) {
          if (k0 + j < n) sa_dbg[k0 + j] = vals[k0 + j];
        }
      }
      // wave carry for the next step: the last head of the highest lane that has one
      if (bh) {
        const int hl = 63 - __clzll(static_cast<long long>(bh));
        carry = static_cast<size_t>(__shfl(my_last_pos, hl, kWave));
      }
      prev_last = __shfl(me[t][kR0Vec - 1], kWave - 1, kWave);
    }
  }
}

// ---- chunked Kasai (linear.cpp:18-41), optional alternative LCP builder --------------------
// One thread per chunk of consecutive text positions, restarting with prefix_len = 0 exactly
// as the reference's per-thread chunks do.  Needs the full-depth SA (rank is a permutation).
template <typename SymT>
__global__ __launch_bounds__(kBlock) void kasai_kernel(const SymT *__restrict__ sym, const uint32_t *__restrict__ sa,
                                                       const RankEntry *__restrict__ rank, size_t n, size_t chunk,
                                                       int32_t *__restrict__ lcp) {
  const size_t c = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  const size_t begin = c * chunk;
  if (begin >= n) return;
  const size_t end = min(n, begin + chunk);
  size_t pl = 0;
  for (size_t i = begin; i < end; i++) {
    const size_t r = rank_of(rank[i]);
    if (r + 1 != n) {
      const size_t j = sa[r + 1];
      const size_t mx = i > j ? i : j;
      while (mx + pl < n && sym[i + pl] == sym[j + pl]) pl++;
      lcp[r] = static_cast<int32_t>(pl);
      if (pl > 0) pl--;
    }
  }
}

}  // namespace wp
