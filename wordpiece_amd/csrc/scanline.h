// scanline.h — the monotone-stack scanlines of linear.cpp:161-213 (get_closest x4), tiled.
//
// The reference runs four sequential passes over SA order (left->right / right->left, prefix /
// suffix tokens).  A token sits on the stack from its own slot until the first boundary whose LCP
// is smaller than its length ("reach").  Here the SA slots are cut into tiles of 4096:
//   1. sl_summary : per tile, the minimum interior LCP, and for every vocab mark in the tile its
//                   reach inside the tile in both directions (one wave per mark, 64 boundaries per
//                   ballot) -> which marks are still on the stack when the scan leaves the tile;
//   2. sl_carry   : four independent chains (class x direction), one wave each: the stack that
//                   enters every tile = (stack entering the previous tile, popped by that tile's
//                   minimum LCP via ballot) ++ that tile's survivors;
//   3. sl_resolve : per slot, the nearest covering mark of the tile, else the deepest entry of
//                   the incoming stack not popped by the running prefix-minimum LCP; the left and
//                   right answers are merged with the reference's rule (linear.cpp:243-250).
// Marks (vocab token starts) are a sorted list of (slot, id, len, class), not a dense who[] array.
#pragma once
#include "primitives.h"
#include "suffix_array.h"

namespace wp {

constexpr int kSlItems = 16;
constexpr int kSlTile = kBlock * kSlItems;  // 4096 SA slots per workgroup
constexpr int32_t kLcpInf = 0x7fffffff;
constexpr uint32_t kMarkLenMask = 0x0fffffffu;
constexpr uint32_t kMarkSurvBwd = 1u << 28;  // on the stack when the right->left scan leaves the tile
constexpr uint32_t kMarkSurvFwd = 1u << 29;  // on the stack when the left->right scan leaves the tile
constexpr int kMarkClsShift = 30;            // 0 = prefix-class token, 1 = ##suffix-class token
constexpr int kCarryWin = 1024;
constexpr int kStackLds = 256;
constexpr int kMarkLds = 256;

__device__ __forceinline__ int pad16(int q) { return q + (q >> 4); }  // stride-16 access without bank conflicts

// boundary LCP between SA slots x and x+1; -1 (never split, depth-capped) and out-of-range = +inf
__device__ __forceinline__ int32_t boundary_lcp(const int32_t *__restrict__ lcp, size_t n, long long x) {
  if (x < 0 || static_cast<size_t>(x) + 1 >= n) return kLcpInf;
  const int32_t v = lcp[x];
  return v < 0 ? kLcpInf : v;
}

// who marks: slot of every eligible token's first symbol (linear.cpp:153-160)
__global__ __launch_bounds__(kBlock) void mark_slots_kernel(const uint32_t *__restrict__ tok_start, int M,
                                                            size_t vocab_base, const RankEntry *__restrict__ rank,
                                                            uint32_t *__restrict__ slot, uint32_t *__restrict__ idx) {
  int m = blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  slot[m] = rank_of(rank[vocab_base + tok_start[m]]);
  idx[m] = static_cast<uint32_t>(m);
}

__global__ __launch_bounds__(kBlock) void mark_gather_kernel(const uint32_t *__restrict__ order, int M,
                                                             const int32_t *__restrict__ tok_id,
                                                             const uint32_t *__restrict__ tok_info,
                                                             int32_t *__restrict__ mid, uint32_t *__restrict__ minfo) {
  int m = blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  mid[m] = tok_id[order[m]];
  minfo[m] = tok_info[order[m]];
}

__global__ __launch_bounds__(kBlock) void tile_mlo_kernel(const uint32_t *__restrict__ mslot, int M, size_t n,
                                                          unsigned ntiles, uint32_t *__restrict__ tile_mlo) {
  unsigned t = blockIdx.x * kBlock + threadIdx.x;
  if (t > ntiles) return;
  const size_t s = static_cast<size_t>(t) * kSlTile;
  int lo = 0, hi = M;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (mslot[mid] < s) lo = mid + 1; else hi = mid;
  }
  tile_mlo[t] = (t == ntiles || s >= n) ? M : lo;
}

// ---- 1. per-tile summary ---------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sl_summary_kernel(const int32_t *__restrict__ lcp, size_t n,
                                                            const uint32_t *__restrict__ mslot,
                                                            uint32_t *__restrict__ minfo,
                                                            const uint32_t *__restrict__ tile_mlo,
                                                            int32_t *__restrict__ interior,
                                                            int32_t *__restrict__ reach_fwd,
                                                            int32_t *__restrict__ reach_bwd) {
  __shared__ int32_t bl[kSlTile + 1];
  __shared__ int32_t smin[8];
  const size_t s = static_cast<size_t>(blockIdx.x) * kSlTile;
  const int cnt = static_cast<int>(min(static_cast<size_t>(kSlTile), n - s));
  int32_t mn = kLcpInf;
  for (int q = threadIdx.x; q <= cnt; q += kBlock) {
    const int32_t v = boundary_lcp(lcp, n, static_cast<long long>(s) - 1 + q);
    bl[q] = v;  // bl[q] = boundary just before local slot q
    if (q >= 1 && q <= cnt - 1) mn = min(mn, v);
  }
  mn = block_reduce_min(mn, smin);  // contains the __syncthreads that publishes bl[]
  if (threadIdx.x == 0) interior[blockIdx.x] = mn;

  const int lane = lane_id(), w = wave_id();
  const uint32_t lo = tile_mlo[blockIdx.x], hi = tile_mlo[blockIdx.x + 1];
  for (uint32_t m = lo + w; m < hi; m += kBlock / kWave) {
    const int i = static_cast<int>(mslot[m] - s);
    const uint32_t info = minfo[m];
    const int32_t len = static_cast<int32_t>(info & kMarkLenMask);
    int jf = cnt;  // first local slot > i not covered (popped at its leading boundary)
    for (int start = i + 1; start < cnt; start += kWave) {
      const int j = start + lane;
      const uint64_t b = __ballot(j < cnt && bl[j] < len);
      if (b) {
        jf = start + __ffsll(static_cast<long long>(b)) - 1;
        break;
      }
    }
    int jb = -1;  // last local slot < i not covered
    for (int top = i - 1; top >= 0; top -= kWave) {
      const int j = top - lane;
      const uint64_t b = __ballot(j >= 0 && bl[j + 1] < len);
      if (b) {
        jb = top - (__ffsll(static_cast<long long>(b)) - 1);
        break;
      }
    }
    if (lane == 0) {
      reach_fwd[m] = static_cast<int32_t>(s + jf);
      reach_bwd[m] = static_cast<int32_t>(static_cast<long long>(s) + jb);
      minfo[m] = (info & ~(kMarkSurvFwd | kMarkSurvBwd)) | (jf == cnt ? kMarkSurvFwd : 0u)
                 | (jb == -1 ? kMarkSurvBwd : 0u);
    }
  }
}

// ---- 2. carry chains ---------------------------------------------------------------------------
// chain = cls*2 + dir (dir 0: left->right, 1: right->left).  Two levels, so that the sequential part is
// short: tiles are grouped by 64.
//   sl_carry_local : one wave per (group, chain) composes its 64 tiles starting from an EMPTY stack.
//       For every tile it stores the stack entering the tile *relative to the group start*
//       (lpool/ldepth) and lmin = the minimum LCP popped since the group start; at the end the
//       group's summary (gmin, surviving pushes) is stored.
//   sl_carry_group : one wave per chain walks the groups: stack entering group g+1 =
//       (stack entering g popped by gmin[g]) ++ summary[g].
// The stack entering a tile is then  { group stack entries with len <= lmin[tile] } ++ local list,
// which sl_resolve queries as two levels without materialising it.
constexpr int kSlGroup = 64;

struct CarryStack {
  int32_t *slen, *sid;
  int depth;
};

// pops: lengths ascend, so the popped entries are a suffix of the stack (ballot over the top 64)
__device__ __forceinline__ void carry_pop(CarryStack &s, int32_t m, int lane) {
  while (s.depth > 0) {
    const int base = max(0, s.depth - kWave);
    const int q = base + lane;
    const uint64_t b = __ballot(q < s.depth && s.slen[q] > m);
    if (!b) break;
    const int first = __ffsll(static_cast<long long>(b)) - 1;
    s.depth = base + first;
    if (first > 0) break;
  }
}

__global__ __launch_bounds__(kWave) void sl_carry_local_kernel(
    const int32_t *__restrict__ lcp, size_t n, unsigned ntiles, const int32_t *__restrict__ interior,
    const uint32_t *__restrict__ tile_mlo, const int32_t *__restrict__ mid, const uint32_t *__restrict__ minfo,
    int M, int D, int2 *__restrict__ lpool, uint32_t *__restrict__ ldepth, int32_t *__restrict__ lmin,
    unsigned ngroups, int2 *__restrict__ gsum_pool, uint32_t *__restrict__ gsum_depth, int32_t *__restrict__ gmin,
    uint32_t *__restrict__ overflow) {
  extern __shared__ int32_t dyn[];
  CarryStack stk{dyn, dyn + D, 0};
  int32_t *win_id = dyn + 2 * D;
  uint32_t *win_info = reinterpret_cast<uint32_t *>(dyn + 2 * D + kCarryWin);
  const unsigned group = blockIdx.x;
  const int chain = blockIdx.y, cls = chain >> 1, dir = chain & 1;
  const int lane = threadIdx.x;
  const uint64_t lt = (1ull << lane) - 1ull;
  const unsigned t_first = group * kSlGroup;
  const unsigned t_count = min(static_cast<unsigned>(kSlGroup), ntiles - t_first);
  int wlo = 0, whi = 0;  // marks [wlo, whi) are in the LDS window
  // one lane per tile: preload the tile scalars (lane = position in scan order)
  int32_t r_m = kLcpInf;
  uint32_t r_lo = 0, r_hi = 0;
  if (static_cast<unsigned>(lane) < t_count) {
    const unsigned t = dir ? t_first + t_count - 1 - lane : t_first + lane;
    const size_t s = static_cast<size_t>(t) * kSlTile;
    const size_t e = min(n, s + kSlTile);
    const int32_t bnd = dir ? boundary_lcp(lcp, n, static_cast<long long>(e) - 1)
                            : boundary_lcp(lcp, n, static_cast<long long>(s) - 1);
    r_m = min(interior[t], bnd);
    r_lo = tile_mlo[t];
    r_hi = tile_mlo[t + 1];
  }
  int32_t run_min = kLcpInf;
  for (unsigned step = 0; step < t_count; step++) {
    const unsigned tile = dir ? t_first + t_count - 1 - step : t_first + step;
    const int32_t m = __shfl(r_m, step, kWave);
    const int lo = static_cast<int>(__shfl(r_lo, step, kWave));
    const int hi = static_cast<int>(__shfl(r_hi, step, kWave));
    // publish the (group-relative) stack entering this tile
    const size_t pbase = (static_cast<size_t>(chain) * ntiles + tile) * D;
    if (lane == 0) {
      ldepth[static_cast<size_t>(chain) * ntiles + tile] = stk.depth;
      lmin[static_cast<size_t>(chain) * ntiles + tile] = run_min;
    }
    for (int q = lane; q < stk.depth; q += kWave) lpool[pbase + q] = make_int2(stk.slen[q], stk.sid[q]);
    carry_pop(stk, m, lane);
    run_min = min(run_min, m);
    // pushes: this tile's marks of our class that survive to the tile edge, in scan order
    for (int c0 = 0; c0 < hi - lo; c0 += kWave) {
      const int k = c0 + lane;
      const int mm = dir ? hi - 1 - k : lo + k;
      const bool in = k < hi - lo;
      const int a = dir ? max(lo, hi - c0 - kWave) : lo + c0;            // lowest mark index of the chunk
      const int b = dir ? hi - 1 - c0 : min(hi - 1, lo + c0 + kWave - 1);  // highest
      if (a < wlo || b >= whi) {  // slide the window (uniform decision)
        if (dir) {
          whi = b + 1;
          wlo = max(0, whi - kCarryWin);
        } else {
          wlo = a;
          whi = min(M, wlo + kCarryWin);
        }
        for (int q = lane; q < whi - wlo; q += kWave) {
          win_id[q] = mid[wlo + q];
          win_info[q] = minfo[wlo + q];
        }
        __builtin_amdgcn_wave_barrier();
      }
      uint32_t info = 0;
      int32_t id = 0;
      if (in) {
        info = win_info[mm - wlo];
        id = win_id[mm - wlo];
      }
      const bool push = in && static_cast<int>(info >> kMarkClsShift) == cls
                        && (info & (dir ? kMarkSurvBwd : kMarkSurvFwd));
      const uint64_t bm = __ballot(push);
      const int np = __popcll(bm);
      if (stk.depth + np > D) {
        if (lane == 0) atomicOr(overflow, 1u);
        break;
      }
      if (push) {
        const int pos = stk.depth + __popcll(bm & lt);
        stk.slen[pos] = static_cast<int32_t>(info & kMarkLenMask);
        stk.sid[pos] = id;
      }
      stk.depth += np;
      __builtin_amdgcn_wave_barrier();
    }
  }
  // group summary
  const size_t gi = static_cast<size_t>(chain) * ngroups + group;
  if (lane == 0) {
    gsum_depth[gi] = stk.depth;
    gmin[gi] = run_min;
  }
  for (int q = lane; q < stk.depth; q += kWave) gsum_pool[gi * D + q] = make_int2(stk.slen[q], stk.sid[q]);
}

__global__ __launch_bounds__(kWave) void sl_carry_group_kernel(unsigned ngroups, int D,
                                                               const int2 *__restrict__ gsum_pool,
                                                               const uint32_t *__restrict__ gsum_depth,
                                                               const int32_t *__restrict__ gmin,
                                                               int2 *__restrict__ gin_pool,
                                                               uint32_t *__restrict__ gin_depth,
                                                               uint32_t *__restrict__ overflow) {
  extern __shared__ int32_t dyn[];
  CarryStack stk{dyn, dyn + D, 0};
  const int chain = blockIdx.x, dir = chain & 1;
  const int lane = threadIdx.x;
  int32_t r_min = kLcpInf;
  uint32_t r_depth = 0;
  for (unsigned step = 0; step < ngroups; step++) {
    if ((step & 63u) == 0) {
      const unsigned st = step + lane;
      if (st < ngroups) {
        const unsigned g = dir ? ngroups - 1 - st : st;
        r_min = gmin[static_cast<size_t>(chain) * ngroups + g];
        r_depth = gsum_depth[static_cast<size_t>(chain) * ngroups + g];
      }
    }
    const unsigned g = dir ? ngroups - 1 - step : step;
    const size_t gi = static_cast<size_t>(chain) * ngroups + g;
    const int32_t m = __shfl(r_min, step & 63, kWave);
    const int add = static_cast<int>(__shfl(r_depth, step & 63, kWave));
    if (lane == 0) gin_depth[gi] = stk.depth;
    for (int q = lane; q < stk.depth; q += kWave) gin_pool[gi * D + q] = make_int2(stk.slen[q], stk.sid[q]);
    carry_pop(stk, m, lane);
    if (stk.depth + add > D) {
      if (lane == 0) atomicOr(overflow, 1u);
      return;
    }
    for (int q = lane; q < add; q += kWave) {
      const int2 e = gsum_pool[gi * D + q];
      stk.slen[stk.depth + q] = e.x;
      stk.sid[stk.depth + q] = e.y;
    }
    stk.depth += add;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- 3. resolve ----------------------------------------------------------------------------------
struct StackView {
  const int32_t *len_lds;  // first kStackLds entries staged in LDS
  const int32_t *id_lds;
  const int2 *glob;        // the full stack in the pool
  int depth;
};

// deepest entry with len <= pm (entries ascend in len); returns its id and length
__device__ __forceinline__ bool stack_lookup(const StackView &sv, int32_t pm, int32_t &id, int32_t &len) {
  int lo = 0, hi = sv.depth;  // first index with len > pm
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t l = mid < kStackLds ? sv.len_lds[mid] : sv.glob[mid].x;
    if (l <= pm) lo = mid + 1; else hi = mid;
  }
  if (lo == 0) return false;
  const int q = lo - 1;
  if (q < kStackLds) {
    id = sv.id_lds[q];
    len = sv.len_lds[q];
  } else {
    const int2 e = sv.glob[q];
    len = e.x;
    id = e.y;
  }
  return true;
}

__global__ __launch_bounds__(kBlock) void sl_resolve_kernel(
    const int32_t *__restrict__ lcp, size_t n, unsigned ntiles, const uint32_t *__restrict__ tile_mlo,
    const uint32_t *__restrict__ mslot, const int32_t *__restrict__ mid, const uint32_t *__restrict__ minfo,
    const int32_t *__restrict__ reach_fwd, const int32_t *__restrict__ reach_bwd, const int2 *__restrict__ lpool,
    const uint32_t *__restrict__ ldepth, const int32_t *__restrict__ lmin, unsigned ngroups,
    const int2 *__restrict__ gin_pool, const uint32_t *__restrict__ gin_depth, int D,
    int32_t *__restrict__ best_prefix, int32_t *__restrict__ best_suffix) {
  __shared__ int32_t bl[kSlTile + kSlTile / 16 + 2];
  __shared__ int32_t pml[kSlTile + kSlTile / 16 + 2];
  __shared__ int32_t st_len[8][kStackLds], st_id[8][kStackLds];  // [chain] local lists, [4+chain] group stacks
  __shared__ int32_t mg_len[4][kStackLds], mg_id[4][kStackLds];  // merged stack entering the tile, per chain
  __shared__ int mg_depth[4];                                    // -1: too deep for LDS, use the two-level lookup
  __shared__ uint32_t mk_slot[kMarkLds], mk_info[kMarkLds];
  __shared__ int32_t mk_id[kMarkLds], mk_rf[kMarkLds], mk_rb[kMarkLds];
  __shared__ int32_t wmin[2][8];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const size_t s = static_cast<size_t>(blockIdx.x) * kSlTile;
  const int cnt = static_cast<int>(min(static_cast<size_t>(kSlTile), n - s));
  for (int q = tid; q <= cnt; q += kBlock) bl[pad16(q)] = boundary_lcp(lcp, n, static_cast<long long>(s) - 1 + q);
  for (int q = cnt + 1 + tid; q <= kSlTile; q += kBlock) bl[pad16(q)] = kLcpInf;

  const unsigned group = blockIdx.x / kSlGroup;
  int depth[8];
  int32_t lm[4];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    const size_t ti = static_cast<size_t>(c) * ntiles + blockIdx.x;
    const size_t gi = static_cast<size_t>(c) * ngroups + group;
    depth[c] = static_cast<int>(ldepth[ti]);
    depth[4 + c] = static_cast<int>(gin_depth[gi]);
    lm[c] = lmin[ti];
    for (int q = tid; q < min(depth[c], kStackLds); q += kBlock) {
      const int2 e = lpool[ti * D + q];
      st_len[c][q] = e.x;
      st_id[c][q] = e.y;
    }
    for (int q = tid; q < min(depth[4 + c], kStackLds); q += kBlock) {
      const int2 e = gin_pool[gi * D + q];
      st_len[4 + c][q] = e.x;
      st_id[4 + c][q] = e.y;
    }
  }
  const int mlo = static_cast<int>(tile_mlo[blockIdx.x]), mhi = static_cast<int>(tile_mlo[blockIdx.x + 1]);
  const int nm = mhi - mlo;
  for (int q = tid; q < min(nm, kMarkLds); q += kBlock) {
    mk_slot[q] = mslot[mlo + q];
    mk_info[q] = minfo[mlo + q];
    mk_id[q] = mid[mlo + q];
    mk_rf[q] = reach_fwd[mlo + q];
    mk_rb[q] = reach_bwd[mlo + q];
  }
  __syncthreads();

  // merged stack per chain: { group entries with len <= lmin } ++ local list (all ascending in len)
  if (tid < 4) {
    const int c = tid;
    int ng = 0;
    const int dg = depth[4 + c], dl = depth[c];
    int md = -1;
    if (dg <= kStackLds && dl <= kStackLds) {
      while (ng < dg && st_len[4 + c][ng] <= lm[c]) ng++;
      if (ng + dl <= kStackLds) {
        for (int q = 0; q < ng; q++) {
          mg_len[c][q] = st_len[4 + c][q];
          mg_id[c][q] = st_id[4 + c][q];
        }
        for (int q = 0; q < dl; q++) {
          mg_len[c][ng + q] = st_len[c][q];
          mg_id[c][ng + q] = st_id[c][q];
        }
        md = ng + dl;
      }
    }
    mg_depth[c] = md;
  }
  __syncthreads();

  // running minima: pml(j) = min bl[0..j], pmr(j) = min bl[j+1..cnt].  Scanned with each thread on 16
  // consecutive slots (padded LDS index), then stored back so the per-slot phase can run lane-striped
  // (coalesced stores of the two result arrays).
  const int j0 = tid * kSlItems;
  int32_t cf = kLcpInf, cb = kLcpInf;
#pragma unroll
  for (int q = 0; q < kSlItems; q++) {
    cf = min(cf, bl[pad16(j0 + q)]);
    cb = min(cb, bl[pad16(j0 + q + 1)]);
  }
  int32_t inf_f = wave_incl_min(cf);
  int32_t ex_f = __shfl_up(inf_f, 1, kWave);
  if (lane == 0) ex_f = kLcpInf;
  int32_t rb = __shfl(cb, 63 - lane, kWave);  // backward: scan the mirrored lanes
  int32_t inr = wave_incl_min(rb);
  int32_t ex_r = __shfl_up(inr, 1, kWave);
  if (lane == 0) ex_r = kLcpInf;
  int32_t ex_b = __shfl(ex_r, 63 - lane, kWave);
  if (lane == 63) wmin[0][w] = inf_f;  // wave totals
  if (lane == 63) wmin[1][w] = inr;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kBlock / kWave; i++) {
    if (i < w) ex_f = min(ex_f, wmin[0][i]);
    if (i > w) ex_b = min(ex_b, wmin[1][i]);
  }
  {
    int32_t vf[kSlItems], vb[kSlItems];
    int32_t run = ex_f;
#pragma unroll
    for (int q = 0; q < kSlItems; q++) {
      run = min(run, bl[pad16(j0 + q)]);
      vf[q] = run;
    }
    run = ex_b;
#pragma unroll
    for (int q = kSlItems - 1; q >= 0; q--) {
      run = min(run, bl[pad16(j0 + q + 1)]);
      vb[q] = run;
    }
    __syncthreads();  // every read of bl[] is done: reuse it for pmr
#pragma unroll
    for (int q = 0; q < kSlItems; q++) {
      pml[pad16(j0 + q)] = vf[q];
      bl[pad16(j0 + q)] = vb[q];
    }
  }
  __syncthreads();
  const int32_t *pmr = bl;

  StackView sv[8];
#pragma unroll
  for (int c = 0; c < 4; c++) {
    sv[c].len_lds = st_len[c];
    sv[c].id_lds = st_id[c];
    sv[c].glob = lpool + (static_cast<size_t>(c) * ntiles + blockIdx.x) * D;
    sv[c].depth = depth[c];
    sv[4 + c].len_lds = st_len[4 + c];
    sv[4 + c].id_lds = st_id[4 + c];
    sv[4 + c].glob = gin_pool + (static_cast<size_t>(c) * ngroups + group) * D;
    sv[4 + c].depth = depth[4 + c];
  }

  for (int j = tid; j < cnt; j += kBlock) {
    const int32_t runf = pml[pad16(j)], runb = pmr[pad16(j)];
    const uint32_t slot = static_cast<uint32_t>(s + j);
    int32_t out[2];
    // position among the tile's marks: first mark with slot > `slot`
    int ub = 0;
    if (nm > 0) {
      int lo = 0, hi = nm;
      while (lo < hi) {
        const int md = (lo + hi) >> 1;
        const uint32_t ms = md < kMarkLds ? mk_slot[md] : mslot[mlo + md];
        if (ms <= slot) lo = md + 1; else hi = md;
      }
      ub = lo;
    }
#pragma unroll
    for (int cls = 0; cls < 2; cls++) {
      int32_t xid = -1, xlen = 0, yid = -1, ylen = 0;
      bool fx = false, fy = false;
      // left->right scan: nearest mark at or before the slot that still covers it
      for (int q2 = ub - 1; q2 >= 0; q2--) {
        const uint32_t info = q2 < kMarkLds ? mk_info[q2] : minfo[mlo + q2];
        if (static_cast<int>(info >> kMarkClsShift) != cls) continue;
        const int32_t rf = q2 < kMarkLds ? mk_rf[q2] : reach_fwd[mlo + q2];
        if (rf > static_cast<int32_t>(slot)) {
          xid = q2 < kMarkLds ? mk_id[q2] : mid[mlo + q2];
          xlen = static_cast<int32_t>(info & kMarkLenMask);
          fx = true;
          break;
        }
      }
      if (!fx) {
        const int c = cls * 2 + 0, md = mg_depth[c];
        if (md >= 0) {
          for (int q2 = md - 1; q2 >= 0; q2--) {
            if (mg_len[c][q2] <= runf) {
              xid = mg_id[c][q2];
              xlen = mg_len[c][q2];
              fx = true;
              break;
            }
          }
        } else {
          fx = stack_lookup(sv[c], runf, xid, xlen);
          if (!fx) fx = stack_lookup(sv[4 + c], min(runf, lm[c]), xid, xlen);
        }
      }
      // right->left scan: nearest mark at or after the slot that still covers it
      int lb = ub;
      if (ub > 0) {
        const uint32_t ps = (ub - 1) < kMarkLds ? mk_slot[ub - 1] : mslot[mlo + ub - 1];
        if (ps == slot) lb = ub - 1;  // a mark on this very slot counts for both directions
      }
      for (int q2 = lb; q2 < nm; q2++) {
        const uint32_t info = q2 < kMarkLds ? mk_info[q2] : minfo[mlo + q2];
        if (static_cast<int>(info >> kMarkClsShift) != cls) continue;
        const int32_t rbk = q2 < kMarkLds ? mk_rb[q2] : reach_bwd[mlo + q2];
        if (rbk < static_cast<int32_t>(slot)) {
          yid = q2 < kMarkLds ? mk_id[q2] : mid[mlo + q2];
          ylen = static_cast<int32_t>(info & kMarkLenMask);
          fy = true;
          break;
        }
      }
      if (!fy) {
        const int c = cls * 2 + 1, md = mg_depth[c];
        if (md >= 0) {
          for (int q2 = md - 1; q2 >= 0; q2--) {
            if (mg_len[c][q2] <= runb) {
              yid = mg_id[c][q2];
              ylen = mg_len[c][q2];
              fy = true;
              break;
            }
          }
        } else {
          fy = stack_lookup(sv[c], runb, yid, ylen);
          if (!fy) fy = stack_lookup(sv[4 + c], min(runb, lm[c]), yid, ylen);
        }
      }
      // linear.cpp:243-250: both -> x iff strictly longer, else y; one -> that one
      int32_t r = -1;
      if (fx && fy) r = xlen > ylen ? xid : yid;
      else if (fx) r = xid;
      else if (fy) r = yid;
      out[cls] = r;
    }
    best_prefix[s + j] = out[0];
    best_suffix[s + j] = out[1];
  }
}

}  // namespace wp
