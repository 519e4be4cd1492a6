// local_sort.h — segmented sort of the active list for prefix-doubling rounds r >= 1.
//
// In round r every still-tied group (contiguous in the active list) must be sorted by
// rank[i + depth(group)].  Most groups are small (after 9 symbols of English text 79 % of the active
// entries sit in groups of <= 2048), so a workgroup takes a window of whole groups into LDS,
// gathers their second keys and sorts the window there with 10-bit LSD passes on the composite
// (local group number, rank[i+h]+1): no HBM pass per digit, one read and one write per entry.
// Only groups larger than kLsMaxGroup take the global radix path (extract -> sort -> write back).
//
// Window rule: workgroup t owns the groups whose head lies in list range [t*T, (t+1)*T).  With
// kLsMaxGroup >= T at most the last owned group can be "large", so a window holds < T + kLsMaxGroup
// entries.
#pragma once
#include "primitives.h"
#include "radix_sort.h"
#include "suffix_array.h"

namespace wp {

#ifndef WP_LS_T
#define WP_LS_T 2048
#endif
#ifndef WP_LS_MAXGROUP
#define WP_LS_MAXGROUP 2048
#endif
constexpr int kLsT = WP_LS_T;                // nominal list entries per workgroup
constexpr int kLsMaxGroup = WP_LS_MAXGROUP;  // groups above this size use the global path
constexpr int kLsCap = kLsT + kLsMaxGroup;  // LDS capacity in entries
constexpr int kLsItems = kLsCap / kBlock;   // 16
constexpr int kLsBits = 10;
constexpr int kLsBins = 1 << kLsBits;
constexpr int kLsGroupBits = 12;  // local group number < kLsCap

// Table of the large groups (size > kLsMaxGroup) of the list: lg_head[i] = list position of the i-th
// large group, lg_off[i] = its offset in the large list, lg_off[n_large_groups] = n_large.  Large groups
// are few, so the table is appended to by atomics instead of two device-wide scans: one 64-bit atomic
// hands out the table index (low word) and the offset (high word) together, so lg_off ascends with the
// index whatever order the groups arrive in (the extract kernel searches lg_off; nothing needs lg_head
// sorted).  counters: two adjacent uint32 {n_large_groups, n_large}, zeroed by the caller.
__global__ __launch_bounds__(kBlock) void large_groups_kernel(const uint32_t *__restrict__ ghead,
                                                              const uint32_t *__restrict__ n_groups_dev,
                                                              unsigned long long *__restrict__ counters,
                                                              uint32_t *__restrict__ lg_head,
                                                              uint32_t *__restrict__ lg_off) {
  const size_t ng = *n_groups_dev;
  for (size_t g = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; g < ng;
       g += static_cast<size_t>(gridDim.x) * kBlock) {
    const uint32_t h = ghead[g];
    const uint32_t s = ghead[g + 1] - h;
    if (s > kLsMaxGroup) {
      const unsigned long long got = atomicAdd(counters, (static_cast<unsigned long long>(s) << 32) | 1ull);
      const uint32_t i = static_cast<uint32_t>(got);
      lg_head[i] = h;
      lg_off[i] = static_cast<uint32_t>(got >> 32);
    }
  }
}
__global__ void large_groups_close_kernel(const uint32_t *__restrict__ counters, uint32_t *__restrict__ lg_off) {
  lg_off[counters[0]] = counters[1];
}

// entries of large groups -> (key, val, list position) in the large list; key = dense large-group
// number << rbits | second key.  A wave owns kLxSpan consecutive positions of the large list: one
// search for its first group, then it only moves forward (large groups are long runs).
constexpr int kLxSpan = 64;  // (2048: 32 dependent gather rounds per wave, 94 us for 5e5 entries; 256: 69 us)
__global__ __launch_bounds__(kBlock) void large_extract_kernel(
    const uint32_t *__restrict__ aval, const uint32_t *__restrict__ adep, const uint32_t *__restrict__ lg_head,
    const uint32_t *__restrict__ lg_off, uint32_t n_lg, size_t n_large, const RankEntry *__restrict__ rank, size_t n,
    int rbits, uint64_t *__restrict__ lkey, uint32_t *__restrict__ lval, uint32_t *__restrict__ lpos,
    const uint32_t *__restrict__ second_key) {
  // second_key != nullptr: the entries are sorted by second_key[list position] (trie.h: the end node of the suffix)
  // instead of by the rank of the suffix `depth` symbols further on
  const int lane = lane_id();
  const size_t wave = static_cast<size_t>(blockIdx.x) * (kBlock / kWave) + wave_id();
  const size_t j0 = wave * kLxSpan;
  if (j0 >= n_large) return;
  // last group with lg_off[i] <= j0
  uint32_t lo = 0, hi = n_lg;
  while (hi - lo > 1) {
    const uint32_t md = (lo + hi) >> 1;
    if (lg_off[md] <= j0) lo = md; else hi = md;
  }
  uint32_t i = lo;
  const size_t j1 = min(n_large, j0 + kLxSpan);
  for (size_t j = j0 + lane; j < j1; j += kWave) {
    while (j >= lg_off[i + 1]) i++;
    const uint32_t k = lg_head[i] + static_cast<uint32_t>(j - lg_off[i]);
    const uint32_t v = aval[k];
    uint32_t r2;
    if (second_key) {
      r2 = second_key[k] + 1u;
    } else {
      const size_t t = static_cast<size_t>(v) + adep[k];
      r2 = t < n ? rank_of(rank[t]) + 1u : 0u;
    }
    lkey[j] = (static_cast<uint64_t>(i) << rbits) | r2;
    lval[j] = v;
    lpos[j] = k;
  }
}

// sorted large list -> back to the list positions (lpos[j]: the list position behind large-list
// position j; the sort keeps every group's entries inside the group's range), with the original group
// id in the high word again
__global__ __launch_bounds__(kBlock) void large_writeback_kernel(const uint64_t *__restrict__ lkey,
                                                                 const uint32_t *__restrict__ lval,
                                                                 const uint32_t *__restrict__ lpos, size_t nl,
                                                                 const uint32_t *__restrict__ agid, int rbits,
                                                                 uint64_t *__restrict__ kout,
                                                                 uint32_t *__restrict__ vout) {
  const uint64_t rmask = (1ull << rbits) - 1ull;
  for (size_t j = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x; j < nl;
       j += static_cast<size_t>(gridDim.x) * kBlock) {
    const uint32_t k = lpos[j];
    kout[k] = (static_cast<uint64_t>(agid[k]) << 32) | (lkey[j] & rmask);
    vout[k] = lval[j];
  }
}

__global__ __launch_bounds__(kBlock) void local_sort_kernel(const uint32_t *__restrict__ aval,
                                                            const uint32_t *__restrict__ agid,
                                                            const uint32_t *__restrict__ adep,
                                                            const uint32_t *__restrict__ sizes_dev,
                                                            const uint32_t *__restrict__ ghead,
                                                            const RankEntry *__restrict__ rank, size_t n, int rbits,
                                                            uint64_t *__restrict__ kout,
                                                            uint32_t *__restrict__ vout,
                                                            const uint32_t *__restrict__ second_key) {
  // second_key: see large_extract_kernel
  // sizes_dev[0] = list length, sizes_dev[1] = number of groups: read on the device, so that the
  // kernel can be queued (with a grid for the largest possible list) while the host is still waiting
  // for the same numbers; workgroups behind the list leave at once
  const size_t m = sizes_dev[0];
  const uint32_t n_groups = sizes_dev[1];
  if (static_cast<size_t>(blockIdx.x) * kLsT >= m) return;
  constexpr int WAVES = kBlock / kWave;
  __shared__ uint64_t skey[kLsCap];
  __shared__ uint32_t sval[kLsCap];
  __shared__ uint32_t wc[WAVES][kLsBins];
  __shared__ uint32_t ssum[8];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t lo = static_cast<size_t>(blockIdx.x) * kLsT;
  const size_t hi = min(m, lo + kLsT);
  // owned groups [g_first, g_end)
  const uint32_t g0 = agid[lo];
  const uint32_t g_first = ghead[g0] == lo ? g0 : g0 + 1;
  uint32_t g_end = n_groups;
  if (hi < m) {
    const uint32_t g1 = agid[hi];
    g_end = ghead[g1] == hi ? g1 : g1 + 1;
  }
  if (g_first >= g_end) return;
  const uint32_t a = ghead[g_first];
  uint32_t b = ghead[g_end];
  if (b - ghead[g_end - 1] > kLsMaxGroup) b = ghead[g_end - 1];  // the last owned group is large
  const uint32_t cnt = b - a;
  if (cnt == 0) return;

  // wave w owns window entries [w*WSPAN, (w+1)*WSPAN) in nr rounds of 64; nr adapts to the window
  const int nr = static_cast<int>((cnt + kBlock - 1) / kBlock);
  const uint32_t WSPAN = static_cast<uint32_t>(nr) * kWave;
  const uint64_t rmask = (1ull << rbits) - 1ull;
  uint64_t key[kLsItems];
  uint32_t val[kLsItems];
#pragma unroll
  for (int r = 0; r < kLsItems; r++) {
    const uint32_t i = w * WSPAN + r * kWave + lane;
    key[r] = ~0ull;
    val[r] = 0;
    if (r < nr && i < cnt) {
      const uint32_t v = aval[a + i];
      uint64_t r2;
      if (second_key) {
        r2 = static_cast<uint64_t>(second_key[a + i]) + 1u;
      } else {
        const size_t t = static_cast<size_t>(v) + adep[a + i];
        r2 = t < n ? rank_of(rank[t]) + 1u : 0u;
      }
      key[r] = (static_cast<uint64_t>(agid[a + i] - g_first) << rbits) | r2;
      val[r] = v;
    }
  }
  const int total_bits = rbits + kLsGroupBits;
  for (int shift = 0; shift < total_bits; shift += kLsBits) {
    for (int q = tid; q < WAVES * kLsBins; q += kBlock) (&wc[0][0])[q] = 0;
    __syncthreads();
    uint32_t rnk[kLsItems];
    volatile uint32_t *mycnt = wc[w];
#pragma unroll
    for (int r = 0; r < kLsItems; r++) {
      rnk[r] = 0;
      if (r < nr) {  // wave-uniform
        const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & (kLsBins - 1);
        rnk[r] = wave_rank_digit<kLsBits>(mycnt, d, lane);
      }
    }
    __syncthreads();
    // bins 4*tid .. 4*tid+3: exclusive over waves, then over bins
    uint32_t tot4 = 0, base4[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int bin = 4 * tid + q;
      uint32_t t = 0;
#pragma unroll
      for (int i = 0; i < WAVES; i++) {
        const uint32_t c = wc[i][bin];
        wc[i][bin] = t;
        t += c;
      }
      base4[q] = tot4;
      tot4 += t;
    }
    uint32_t all;
    const uint32_t ex = block_excl_sum(tot4, ssum, all);
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int bin = 4 * tid + q;
#pragma unroll
      for (int i = 0; i < WAVES; i++) wc[i][bin] += ex + base4[q];
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kLsItems; r++) {
      if (r < nr) {
        const uint32_t d = static_cast<uint32_t>(key[r] >> shift) & (kLsBins - 1);
        const uint32_t pos = wc[w][d] + rnk[r];
        skey[pos] = key[r];
        sval[pos] = val[r];
      }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kLsItems; r++) {
      if (r < nr) {
        const uint32_t i = w * WSPAN + r * kWave + lane;
        key[r] = skey[i];
        val[r] = sval[i];
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int r = 0; r < kLsItems; r++) {
    const uint32_t i = w * WSPAN + r * kWave + lane;
    if (r < nr && i < cnt) {
      const uint64_t g = static_cast<uint64_t>(g_first) + (key[r] >> rbits);
      kout[a + i] = (g << 32) | (key[r] & rmask);
      vout[a + i] = val[r];
    }
  }
}

}  // namespace wp
