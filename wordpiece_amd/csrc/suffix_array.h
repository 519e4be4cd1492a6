// suffix_array.h — suffix array + rank + LCP by prefix doubling (replaces libsais_int,
// the inverse-SA loop and calcLcp: linear.cpp:118-149).
//
// Round 0 sorts every suffix by its first K symbols packed into one 64-bit key (LSD radix).
// Round r >= 1 works on the *active list* only — the slots of groups that are still tied —
// sorting (dense group id, rank[i+h]) and splitting groups; singletons retire.  rank[i] is
// the first SA slot of i's group, so a retired suffix already holds its final rank.
//
// LCP comes out of the same passes: a boundary that appears in round 0 gets its LCP from the two
// packed keys (count of equal leading symbols); a boundary that appears in the round with offset
// h separates two suffixes that agree on exactly h symbols plus whatever their (i+h, j+h)
// continuations share, which is < h, so it is one bounded symbol compare.  Boundaries that never
// appear (depth-capped mode) keep -1 = "LCP >= sorted depth".
#pragma once
#include "primitives.h"

namespace wp {

constexpr int kRrItems = 8;
constexpr int kRrTile = kBlock * kRrItems;  // 2048 list entries per workgroup

struct RerankAgg {
  uint32_t last_flag;  // 1 + largest k in the tile that starts a group, 0 if none
  uint32_t n_active;   // entries of non-singleton groups
  uint32_t n_heads;    // heads of non-singleton groups
};

// keys for round r >= 1 over the active list: (dense group id << 32) | (rank[i+h]+1, 0 past the end)
__global__ __launch_bounds__(kBlock) void build_keys_round_kernel(const uint32_t *__restrict__ aval,
                                                                  const uint32_t *__restrict__ agid, size_t n_act,
                                                                  const uint32_t *__restrict__ rank, uint32_t h,
                                                                  size_t n, uint64_t *__restrict__ keys) {
  size_t k = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= n_act) return;
  const size_t j = static_cast<size_t>(aval[k]) + h;
  const uint32_t r2 = j < n ? rank[j] + 1u : 0u;
  keys[k] = (static_cast<uint64_t>(agid[k]) << 32) | r2;
}

__device__ __forceinline__ void rr_flags(const uint64_t *__restrict__ keys, size_t m, size_t k, bool &flag,
                                         bool &single) {
  const uint64_t me = keys[k];
  flag = (k == 0) || keys[k - 1] != me;
  single = flag && (k + 1 == m || keys[k + 1] != me);
}

__global__ __launch_bounds__(kBlock) void rerank_agg_kernel(const uint64_t *__restrict__ keys, size_t m,
                                                            RerankAgg *__restrict__ agg) {
  __shared__ uint32_t sm[8];
  __shared__ int32_t smx[8];
  const size_t base = static_cast<size_t>(blockIdx.x) * kRrTile + static_cast<size_t>(threadIdx.x) * kRrItems;
  int32_t last = -1;
  uint32_t na = 0, nh = 0;
#pragma unroll
  for (int j = 0; j < kRrItems; j++) {
    size_t k = base + j;
    if (k < m) {
      bool f, s;
      rr_flags(keys, m, k, f, s);
      if (f) last = static_cast<int32_t>(k - static_cast<size_t>(blockIdx.x) * kRrTile);
      na += !s;
      nh += (f && !s);
    }
  }
  uint32_t ta, th;
  (void)block_excl_sum(na, sm, ta);
  (void)block_excl_sum(nh, sm, th);
  int32_t mx = block_incl_max(last, smx);
  if (threadIdx.x == kBlock - 1) {
    RerankAgg a;
    a.last_flag = mx < 0 ? 0u : static_cast<uint32_t>(static_cast<size_t>(blockIdx.x) * kRrTile + mx + 1);
    a.n_active = ta;
    a.n_heads = th;
    agg[blockIdx.x] = a;
  }
}

// single block: exclusive prefix over tiles (running max of last_flag, sums of the counts);
// totals[0] = n_active, totals[1] = n_heads
__global__ __launch_bounds__(kBlock) void rerank_spine_kernel(RerankAgg *__restrict__ agg, size_t tiles,
                                                              uint32_t *__restrict__ totals) {
  __shared__ uint32_t sm[8];
  __shared__ int32_t smx[8];
  uint32_t ca = 0, ch = 0;
  int32_t cm = 0;
  for (size_t base = 0; base < tiles; base += kBlock) {
    size_t i = base + threadIdx.x;
    RerankAgg a = {0, 0, 0};
    if (i < tiles) a = agg[i];
    uint32_t ta, th;
    uint32_t ea = block_excl_sum(a.n_active, sm, ta);
    uint32_t eh = block_excl_sum(a.n_heads, sm, th);
    // exclusive running max: shift by one thread
    int32_t inc = block_incl_max(static_cast<int32_t>(a.last_flag), smx);
    __shared__ int32_t shifted[kBlock];
    __syncthreads();
    shifted[threadIdx.x] = inc;
    __syncthreads();
    int32_t exm = threadIdx.x == 0 ? 0 : shifted[threadIdx.x - 1];
    int32_t blockmax = shifted[kBlock - 1];
    if (i < tiles) {
      RerankAgg o;
      o.last_flag = static_cast<uint32_t>(max(cm, exm));
      o.n_active = ca + ea;
      o.n_heads = ch + eh;
      agg[i] = o;
    }
    ca += ta;
    ch += th;
    cm = max(cm, blockmax);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    totals[0] = ca;
    totals[1] = ch;
  }
}

template <typename SymT>
__device__ __forceinline__ int32_t lcp_compare(const SymT *__restrict__ sym, size_t n, size_t a, size_t b,
                                               int32_t maxlen) {
  int32_t t = 0;
  while (t < maxlen && a + t < n && b + t < n && sym[a + t] == sym[b + t]) t++;
  return t;
}

// Applies one round's split.  ROUND0: list == all slots (slot k == k), keys are packed symbols.
template <typename SymT, bool ROUND0>
__global__ __launch_bounds__(kBlock) void rerank_apply_kernel(
    const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ slots,
    size_t m, const RerankAgg *__restrict__ agg, const SymT *__restrict__ sym, size_t n, uint32_t h, int K,
    int bits, uint32_t *__restrict__ sa, uint32_t *__restrict__ rank, int32_t *__restrict__ lcp,
    uint32_t *__restrict__ nslots, uint32_t *__restrict__ nvals, uint32_t *__restrict__ ngid) {
  __shared__ uint32_t sm[8];
  __shared__ int32_t smx[8];
  const size_t tile_base = static_cast<size_t>(blockIdx.x) * kRrTile;
  const size_t base = tile_base + static_cast<size_t>(threadIdx.x) * kRrItems;
  const RerankAgg pre = agg[blockIdx.x];

  bool f[kRrItems], s[kRrItems];
  uint32_t na = 0, nh = 0;
  int32_t last = -1;
#pragma unroll
  for (int j = 0; j < kRrItems; j++) {
    size_t k = base + j;
    f[j] = false;
    s[j] = true;
    if (k < m) {
      rr_flags(keys, m, k, f[j], s[j]);
      if (f[j]) last = static_cast<int32_t>(k - tile_base);
      na += !s[j];
      nh += (f[j] && !s[j]);
    }
  }
  uint32_t ta, th;
  uint32_t ea = block_excl_sum(na, sm, ta) + pre.n_active;
  uint32_t eh = block_excl_sum(nh, sm, th) + pre.n_heads;
  // head of the group of the entry just before this thread's first entry
  int32_t inc = block_incl_max(last, smx);
  __shared__ int32_t shifted[kBlock];
  __syncthreads();
  shifted[threadIdx.x] = inc;
  __syncthreads();
  const int32_t exm = threadIdx.x == 0 ? -1 : shifted[threadIdx.x - 1];
  // current head as an index into the list (k-space); pre.last_flag is 1-based
  size_t head = exm >= 0 ? tile_base + exm : (pre.last_flag ? static_cast<size_t>(pre.last_flag) - 1 : 0);

#pragma unroll
  for (int j = 0; j < kRrItems; j++) {
    size_t k = base + j;
    if (k >= m) break;
    if (f[j]) head = k;
    const uint32_t v = vals[k];
    const uint32_t x = ROUND0 ? static_cast<uint32_t>(k) : slots[k];
    const uint32_t head_slot = ROUND0 ? static_cast<uint32_t>(head) : slots[head];
    sa[x] = v;
    if (ROUND0) {
      rank[v] = head_slot;
      if (k > 0) {
        int32_t l = -1;
        if (f[j]) {
          const uint64_t d = keys[k] ^ keys[k - 1];
          const int lead = __clzll(static_cast<long long>(d)) - (64 - K * bits);
          l = lead / bits;
        }
        lcp[x - 1] = l;
      }
    } else {
      // the rank changes only when the (new) head is not the old group head
      const bool head_is_new = head > 0 && (keys[head] >> 32) == (keys[head - 1] >> 32);
      if (head_is_new) rank[v] = head_slot;
      if (f[j] && k > 0 && (keys[k] >> 32) == (keys[k - 1] >> 32)) {
        // x-1 is the previous list entry's slot: both belong to one old group
        lcp[x - 1] = static_cast<int32_t>(h)
                     + lcp_compare(sym, n, static_cast<size_t>(vals[k - 1]) + h, static_cast<size_t>(v) + h,
                                   static_cast<int32_t>(h));
      }
    }
    if (!s[j]) {
      if (f[j]) eh++;
      nslots[ea] = x;
      nvals[ea] = v;
      ngid[ea] = eh - 1;
      ea++;
    }
  }
}

// ---- chunked Kasai (linear.cpp:18-41), optional alternative LCP builder --------------------
// One thread per chunk of consecutive text positions, restarting with prefix_len = 0 exactly
// as the reference's per-thread chunks do.  Needs the full-depth SA (rank is a permutation).
template <typename SymT>
__global__ __launch_bounds__(kBlock) void kasai_kernel(const SymT *__restrict__ sym, const uint32_t *__restrict__ sa,
                                                       const uint32_t *__restrict__ rank, size_t n, size_t chunk,
                                                       int32_t *__restrict__ lcp) {
  const size_t c = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  const size_t begin = c * chunk;
  if (begin >= n) return;
  const size_t end = min(n, begin + chunk);
  size_t pl = 0;
  for (size_t i = begin; i < end; i++) {
    const size_t r = rank[i];
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
