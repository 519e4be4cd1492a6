// scanline.h — the monotone-stack scanlines of linear.cpp:161-213 (get_closest x4), tiled.
//
// The reference runs four sequential passes over SA order (left->right / right->left, prefix /
// suffix tokens).  A token sits on the stack from its own slot until the scan crosses the first
// boundary whose LCP is smaller than its length (its "reach"); the stack top at a slot is the
// nearest mark before (after) it whose reach still covers the slot.  Here:
//   1. sl_summary      : SA slots in tiles of 4096: per tile the minimum LCP a scan crosses, and per
//                        vocab mark its reach inside the tile (one wave per mark, 64 boundaries per
//                        ballot = the stack pop test of 64 slots at once);
//   2. sl_reach_global : marks still on the stack at the tile edge continue over whole tiles /
//                        groups of tiles by their minima (ballots again) to the exact pop boundary;
//   3. pieces          : the four result arrays of the reference are step functions of the slot
//                        with <= 3M+1 steps (M = eligible vocab tokens); each step is evaluated
//                        once with the reference's left/right merge rule (linear.cpp:243-250) and
//                        the walk looks slots up through a bucket index.  step_expand_kernel
//                        materialises the per-slot arrays for the parity tests.
// Marks (vocab token starts) are a sorted list of (slot, id, len, class), not a dense who[] array.
#pragma once
#include "primitives.h"
#include "suffix_array.h"

namespace wp {

constexpr int kSlItems = 16;
constexpr int kSlTile = kBlock * kSlItems;  // 4096 SA slots per workgroup
constexpr int32_t kLcpInf = 0x7fffffff;
constexpr uint32_t kMarkLenMask = 0x0fffffffu;
// step values with the token length packed above the id: ids < 2^20 lines, lengths < 2^11 symbols (the sign bit stays clear)
#ifndef WP_STEP_BUCKET_BITS
#define WP_STEP_BUCKET_BITS 18
#endif
constexpr int kStepBucketBits = WP_STEP_BUCKET_BITS;  // the step table's index: 2^18 .. 2^21 buckets of SA slots (linear_path.h)
#ifndef WP_STEP_BUCKET_BITS_MAX
#define WP_STEP_BUCKET_BITS_MAX 21
#endif
constexpr int kStepBucketBitsMax = WP_STEP_BUCKET_BITS_MAX;
constexpr int kStepIdBits = 20;
constexpr int kStepMaxLen = 1 << 11;
constexpr uint32_t kMarkSurvBwd = 1u << 28;  // on the stack when the right->left scan leaves the tile
constexpr uint32_t kMarkSurvFwd = 1u << 29;  // on the stack when the left->right scan leaves the tile
constexpr int kMarkClsShift = 30;            // 0 = prefix-class token, 1 = ##suffix-class token

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
// tmin_f[t] = min LCP over the leading boundaries of the tile's slots (what a left->right scan
// crosses while passing the tile), tmin_b[t] = min over the trailing boundaries (right->left).
// For every mark of the tile: its reach inside the tile in both directions (one wave per mark, 64
// boundaries per ballot) and whether it is still on the stack at the tile edge.
__global__ __launch_bounds__(kBlock) void sl_summary_kernel(const int32_t *__restrict__ lcp, size_t n,
                                                            const uint32_t *__restrict__ mslot,
                                                            uint32_t *__restrict__ minfo,
                                                            const uint32_t *__restrict__ tile_mlo,
                                                            int32_t *__restrict__ tmin_f, int32_t *__restrict__ tmin_b,
                                                            int32_t *__restrict__ reach_fwd,
                                                            int32_t *__restrict__ reach_bwd) {
  __shared__ int32_t bl[kSlTile + 1];
  __shared__ int32_t smin[8];
  const size_t s = static_cast<size_t>(blockIdx.x) * kSlTile;
  const int cnt = static_cast<int>(min(static_cast<size_t>(kSlTile), n - s));
  int32_t mf = kLcpInf, mb = kLcpInf;
  for (int q = threadIdx.x; q <= cnt; q += kBlock) {
    const int32_t v = boundary_lcp(lcp, n, static_cast<long long>(s) - 1 + q);
    bl[q] = v;  // bl[q] = boundary just before local slot q
    if (q <= cnt - 1) mf = min(mf, v);
    if (q >= 1) mb = min(mb, v);
  }
  mf = block_reduce_min(mf, smin);  // contains the __syncthreads that publishes bl[]
  mb = block_reduce_min(mb, smin);
  if (threadIdx.x == 0) {
    tmin_f[blockIdx.x] = mf;
    tmin_b[blockIdx.x] = mb;
  }

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

constexpr int kSlGroup = 64;
__global__ __launch_bounds__(kBlock) void sl_group_min_kernel(const int32_t *__restrict__ tmin_f,
                                                              const int32_t *__restrict__ tmin_b, unsigned ntiles,
                                                              unsigned ngroups, int32_t *__restrict__ gmin_f,
                                                              int32_t *__restrict__ gmin_b) {
  const unsigned g = (blockIdx.x * kBlock + threadIdx.x) >> 6;  // one wave per group
  const int lane = lane_id();
  if (g >= ngroups) return;
  const unsigned t = g * kSlGroup + lane;
  int32_t f = t < ntiles ? tmin_f[t] : kLcpInf, b = t < ntiles ? tmin_b[t] : kLcpInf;
  f = wave_reduce_min(f);
  b = wave_reduce_min(b);
  if (lane == 0) {
    gmin_f[g] = f;
    gmin_b[g] = b;
  }
}

// ---- 2. global reach ------------------------------------------------------------------------------
// A mark that survives its tile keeps its place on the reference's stack until the scan crosses a
// boundary with LCP < len.  One wave per mark: skip tiles (then groups of 64 tiles) whose minimum
// cannot pop it, 64 candidates per ballot, then locate the boundary inside the stopping tile.
__global__ __launch_bounds__(kBlock) void sl_reach_global_kernel(
    const int32_t *__restrict__ lcp, size_t n, unsigned ntiles, unsigned ngroups, const uint32_t *__restrict__ mslot,
    const uint32_t *__restrict__ minfo, int M, const int32_t *__restrict__ tmin_f, const int32_t *__restrict__ tmin_b,
    const int32_t *__restrict__ gmin_f, const int32_t *__restrict__ gmin_b, int32_t *__restrict__ reach_fwd,
    int32_t *__restrict__ reach_bwd) {
  const int m = static_cast<int>((static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6);
  const int lane = lane_id();
  if (m >= M) return;
  const uint32_t info = minfo[m];
  const int32_t len = static_cast<int32_t>(info & kMarkLenMask);
  const unsigned t0 = mslot[m] / kSlTile;
  if (info & kMarkSurvFwd) {
    long long stop = -1;  // first tile after t0 whose leading boundaries pop the mark
    const unsigned g0 = t0 / kSlGroup;
    {
      const unsigned t = g0 * kSlGroup + lane;
      const uint64_t b = __ballot(t > t0 && t < ntiles && tmin_f[t] < len);
      if (b) stop = g0 * kSlGroup + __ffsll(static_cast<long long>(b)) - 1;
    }
    if (stop < 0) {
      for (unsigned gb = g0 + 1; gb < ngroups && stop < 0; gb += kWave) {
        const unsigned g = gb + lane;
        const uint64_t b = __ballot(g < ngroups && gmin_f[g] < len);
        if (b) {
          const unsigned gs = gb + __ffsll(static_cast<long long>(b)) - 1;
          const unsigned t = gs * kSlGroup + lane;
          const uint64_t b2 = __ballot(t < ntiles && tmin_f[t] < len);
          stop = gs * kSlGroup + __ffsll(static_cast<long long>(b2)) - 1;
        }
      }
    }
    long long r = static_cast<long long>(n);
    if (stop >= 0) {
      const size_t s = static_cast<size_t>(stop) * kSlTile;
      const size_t e = min(n, s + kSlTile);
      for (size_t j0 = s; j0 < e; j0 += kWave) {
        const size_t j = j0 + lane;
        const uint64_t b = __ballot(j < e && boundary_lcp(lcp, n, static_cast<long long>(j) - 1) < len);
        if (b) {
          r = static_cast<long long>(j0) + __ffsll(static_cast<long long>(b)) - 1;
          break;
        }
      }
    }
    if (lane == 0) reach_fwd[m] = static_cast<int32_t>(r);
  }
  if (info & kMarkSurvBwd) {
    long long stop = -1;  // last tile before t0 whose trailing boundaries pop the mark
    const unsigned g0 = t0 / kSlGroup;
    {
      const unsigned t = g0 * kSlGroup + lane;
      const uint64_t b = __ballot(t < t0 && tmin_b[t] < len);
      if (b) stop = g0 * kSlGroup + (63 - __clzll(static_cast<long long>(b)));
    }
    if (stop < 0) {
      for (long long gb = static_cast<long long>(g0) - 1; gb >= 0 && stop < 0; gb -= kWave) {
        const long long g = gb - lane;
        const uint64_t b = __ballot(g >= 0 && gmin_b[g] < len);
        if (b) {
          const long long gs = gb - (__ffsll(static_cast<long long>(b)) - 1);
          const unsigned t = static_cast<unsigned>(gs) * kSlGroup + lane;
          const uint64_t b2 = __ballot(t < ntiles && tmin_b[t] < len);
          stop = gs * kSlGroup + (63 - __clzll(static_cast<long long>(b2)));
        }
      }
    }
    long long r = -1;
    if (stop >= 0) {
      const long long s = stop * kSlTile;
      const long long e = static_cast<long long>(min(n, static_cast<size_t>(s) + kSlTile));
      for (long long top = e - 1; top >= s; top -= kWave) {
        const long long j = top - lane;
        const uint64_t b = __ballot(j >= s && boundary_lcp(lcp, n, j) < len);
        if (b) {
          r = top - (__ffsll(static_cast<long long>(b)) - 1);
          break;
        }
      }
    }
    if (lane == 0) reach_bwd[m] = static_cast<int32_t>(r);
  }
}

// ---- 3. the answer as a step function of the SA slot ---------------------------------------------
// The reference fills best_left/right_prefix/suffix for every slot.  Those arrays only change at a
// mark's slot or where a mark's reach ends, so they are step functions with <= 3M+1 steps.  The
// steps are evaluated here (3M+1 slots instead of n) and the walk looks its slots up in them.
struct MarkView {
  const uint32_t *mslot;
  const int32_t *mid;
  const uint32_t *minfo;
  const int32_t *reach_fwd, *reach_bwd;
  int M;
  // per class c: cover_f[c*M + q] = max reach_fwd over marks <= q of class c (no mark at or before q
  // covers slot j when it is <= j); cover_b[c*M + q] = min reach_bwd over marks >= q of class c
  const int32_t *cover_f, *cover_b;
};

// grid = 4 workgroups (class x direction): running max / min per class over the (sorted) marks.
// 1024 threads x kCoverItems consecutive marks per step, so a step of the serial carry chain covers
// 8192 marks (the 256-thread, one-mark version took 0.1 ms for 29 k marks and 3.5 ms for 1 M).
constexpr int kCoverThreads = 1024;
constexpr int kCoverItems = 8;
__global__ __launch_bounds__(kCoverThreads) void mark_cover_kernel(const uint32_t *__restrict__ minfo,
                                                                   const int32_t *__restrict__ reach_fwd,
                                                                   const int32_t *__restrict__ reach_bwd, int M,
                                                                   int32_t *__restrict__ cover_f,
                                                                   int32_t *__restrict__ cover_b) {
  constexpr int WAVES = kCoverThreads / kWave;
  __shared__ int32_t wm[WAVES];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = blockIdx.x >> 1, back = blockIdx.x & 1;
  constexpr int32_t kNone = -2147483647 - 1;
  int32_t carry = kNone;
  for (int base = 0; base < M; base += kCoverThreads * kCoverItems) {
    // forward: running max of reach_fwd; backward: running min of reach_bwd as a max of negated values,
    // taken over the marks in reverse order (scan index i <-> mark M-1-i)
    // (all loads of a step issued before the first use, none behind a branch: eight marks per thread were eight
    // chains of info -> reach in a row, 62 us for 29 k marks)
    int32_t v[kCoverItems], mx = kNone;
    uint32_t info[kCoverItems];
    int32_t rv[kCoverItems];
    const int32_t *__restrict__ reach = back ? reach_bwd : reach_fwd;
#pragma unroll
    for (int j = 0; j < kCoverItems; j++) {
      const int i = base + tid * kCoverItems + j;
      const int q = i < M ? (back ? M - 1 - i : i) : 0;
      info[j] = minfo[q];
      rv[j] = reach[q];
    }
#pragma unroll
    for (int j = 0; j < kCoverItems; j++) {
      const int i = base + tid * kCoverItems + j;
      v[j] = kNone;
      if (i < M && static_cast<int>(info[j] >> kMarkClsShift) == c) v[j] = back ? -rv[j] - 1 : rv[j];
      mx = max(mx, v[j]);
      v[j] = mx;  // inclusive within the thread
    }
    const int32_t inc = wave_incl_max(mx);
    if (lane == kWave - 1) wm[w] = inc;
    __syncthreads();
    int32_t before = carry, all = carry;
#pragma unroll
    for (int q = 0; q < WAVES; q++) {
      if (q < w) before = max(before, wm[q]);
      all = max(all, wm[q]);
    }
    const int32_t up = __shfl_up(inc, 1, kWave);
    if (lane > 0) before = max(before, up);
#pragma unroll
    for (int j = 0; j < kCoverItems; j++) {
      const int i = base + tid * kCoverItems + j;
      if (i < M) {
        const int q = back ? M - 1 - i : i;
        const int32_t r = max(before, v[j]);
        if (back) cover_b[static_cast<size_t>(c) * M + q] = r == kNone ? 2147483647 : -(r + 1);
        else cover_f[static_cast<size_t>(c) * M + q] = r;
      }
    }
    carry = all;
    __syncthreads();
  }
}

// step starts: slot 0 and, per mark, {first covered slot, own slot, the slot after it, first slot
// past its reach}.  (For duplicate-free vocabularies the first and last alone would do; the own
// slot and its successor matter when two marks carry equal strings)
constexpr int kStepsPerMark = 4;
__global__ __launch_bounds__(kBlock) void piece_starts_kernel(MarkView mv, size_t n, uint32_t *__restrict__ pstart) {
  const int m = blockIdx.x * kBlock + threadIdx.x;
  if (m == 0) pstart[kStepsPerMark * static_cast<size_t>(mv.M)] = 0;
  if (m >= mv.M) return;
  const uint32_t last = static_cast<uint32_t>(n - 1);
  uint32_t *o = pstart + kStepsPerMark * static_cast<size_t>(m);
  // (marks of the text-only layout stand in front of slot `mslot`, which may be n: every start is clamped
  // to the last slot, so that the list sorts within bit_length(n) bits and every start is a real slot)
  o[0] = min(static_cast<uint32_t>(mv.reach_bwd[m] + 1), last);
  o[1] = min(mv.mslot[m], last);
  o[2] = min(mv.mslot[m] + 1u, last);
  o[3] = min(static_cast<uint32_t>(mv.reach_fwd[m]), last);
}

// One wave per step: evaluates the reference's merged answer at the step's first slot for both
// classes.  The stack top of the left->right scan is the nearest mark at or before the slot whose
// reach still covers it; the wave tests 64 marks per ballot walking away from the slot (and
// symmetrically for the right->left scan).
__global__ __launch_bounds__(kBlock) void piece_values_kernel(MarkView mv, const uint32_t *__restrict__ pstart,
                                                              int P, int32_t *__restrict__ pval_prefix,
                                                              int32_t *__restrict__ pval_suffix, int packed) {
  const int k = static_cast<int>((static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6);
  const int lane = lane_id();
  if (k >= P) return;
  const uint32_t slot = pstart[k];
  int lo = 0, hi = mv.M;  // first mark with slot > `slot`: 64 probes per step while the range is wide (decode.h)
  while (hi - lo > 64) {
    const int st = (hi - lo) / kWave + 1;
    const int idx = lo + lane * st;
    const bool gt = idx < hi ? mv.mslot[idx] > slot : true;
    const uint64_t m = __ballot(gt);
    if (!m) {
      lo += (kWave - 1) * st + 1;
      continue;
    }
    const int t = __ffsll(static_cast<long long>(m)) - 1;
    const int first_gt = lo + t * st;
    if (t) lo += (t - 1) * st + 1;
    hi = first_gt < hi ? first_gt : hi;
    if (!t) hi = lo;
  }
  while (lo < hi) {
    const int md = (lo + hi) >> 1;
    if (mv.mslot[md] <= slot) lo = md + 1; else hi = md;
  }
  const int ub = lo;
  int xq[2] = {-1, -1}, yq[2] = {-1, -1};  // mark index of the stack tops per class
  bool fdone[2] = {false, false}, bdone[2] = {false, false};  // class finished: found, or nothing can cover
  for (int top = ub - 1; top >= 0; top -= kWave) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (xq[c] >= 0 || mv.cover_f[static_cast<size_t>(c) * mv.M + top] <= static_cast<int32_t>(slot)) fdone[c] = true;
    }
    if (fdone[0] && fdone[1]) break;
    const int q = top - lane;
    bool cover = false;
    int cls = 0;
    if (q >= 0) {
      const uint32_t info = mv.minfo[q];
      cls = static_cast<int>(info >> kMarkClsShift);
      cover = mv.reach_fwd[q] > static_cast<int32_t>(slot);
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const uint64_t b = __ballot(cover && cls == c);
      if (xq[c] < 0 && b) xq[c] = top - (__ffsll(static_cast<long long>(b)) - 1);
    }
  }
  int lb = ub;
  if (ub > 0 && mv.mslot[ub - 1] == slot) lb = ub - 1;  // a mark on this very slot counts for both scans
  for (int base = lb; base < mv.M; base += kWave) {
#pragma unroll
    for (int c = 0; c < 2; c++) {
      if (yq[c] >= 0 || mv.cover_b[static_cast<size_t>(c) * mv.M + base] >= static_cast<int32_t>(slot)) bdone[c] = true;
    }
    if (bdone[0] && bdone[1]) break;
    const int q = base + lane;
    bool cover = false;
    int cls = 0;
    if (q < mv.M) {
      const uint32_t info = mv.minfo[q];
      cls = static_cast<int>(info >> kMarkClsShift);
      cover = mv.reach_bwd[q] < static_cast<int32_t>(slot);
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const uint64_t b = __ballot(cover && cls == c);
      if (yq[c] < 0 && b) yq[c] = base + __ffsll(static_cast<long long>(b)) - 1;
    }
  }
  if (lane < 2) {
    const int c = lane;
    const int x = c ? xq[1] : xq[0], y = c ? yq[1] : yq[0];
    int wq = -1;  // the winning mark
    if (x >= 0 && y >= 0) {  // linear.cpp:243-250: both -> x iff strictly longer, else y
      const int32_t xl = static_cast<int32_t>(mv.minfo[x] & kMarkLenMask), yl = static_cast<int32_t>(mv.minfo[y] & kMarkLenMask);
      wq = xl > yl ? x : y;
    } else if (x >= 0) {
      wq = x;
    } else if (y >= 0) {
      wq = y;
    }
    int32_t r = wq >= 0 ? mv.mid[wq] : -1;
    // packed: the token's length rides in the bits above its id (step_id / step_len below): the walk gets both
    // with one load instead of id -> tok_len[id], one link less in its chain of dependent loads
    if (packed && r >= 0) r |= static_cast<int32_t>(mv.minfo[wq] & kMarkLenMask) << kStepIdBits;
    (c ? pval_suffix : pval_prefix)[k] = r;
  }
}

// bidx[b] = first step whose start is >= b << shift (b = 0..nbuckets)
__global__ __launch_bounds__(kBlock) void piece_bucket_kernel(const uint32_t *__restrict__ pstart, int P, int shift,
                                                              unsigned nbuckets, uint32_t *__restrict__ bidx) {
  const unsigned b = blockIdx.x * kBlock + threadIdx.x;
  if (b > nbuckets) return;
  const uint64_t key = static_cast<uint64_t>(b) << shift;
  int lo = 0, hi = P;
  while (lo < hi) {
    const int md = (lo + hi) >> 1;
    if (pstart[md] < key) lo = md + 1; else hi = md;
  }
  bidx[b] = static_cast<uint32_t>(lo);
}

// bfast[b]: the two values of the one step that covers bucket b — or, when steps start inside the bucket,
// {kStepSlow | their number, index of the first of them}: the walk's lookup is rank -> this entry (one 8-byte load)
// for most slots, and a short binary search over those few starts plus one value load for the others, instead of
// rank -> bucket pair -> starts -> value.  (Measured and dropped: a 32-byte record per bucket with one or two inner
// starts, which makes the second case one more load as well — the walk was 4 % slower with it.)
constexpr int32_t kStepSlow = static_cast<int32_t>(0x80000000u);  // (values are -1 or >= 0; a slow entry is < -1)
__global__ __launch_bounds__(kBlock) void piece_bucket_fast_kernel(const uint32_t *__restrict__ bidx,
                                                                   const int32_t *__restrict__ pval_prefix,
                                                                   const int32_t *__restrict__ pval_suffix, unsigned nbuckets,
                                                                   int2 *__restrict__ bfast) {
  const unsigned b = blockIdx.x * kBlock + threadIdx.x;
  if (b >= nbuckets) return;
  const uint32_t lo = bidx[b], hi = bidx[b + 1];
  // (bucket 0 holds step 0, which starts at slot 0: lo == hi implies lo > 0)
  bfast[b] = lo == hi ? int2{pval_prefix[lo - 1], pval_suffix[lo - 1]}
                      : int2{kStepSlow | static_cast<int32_t>(hi - lo), static_cast<int32_t>(lo)};
}

struct StepTable {
  const uint32_t *pstart;
  const int32_t *pval_prefix, *pval_suffix;
  const uint32_t *bidx;
  int shift;
  int packed;  // values are (token length << kStepIdBits) | id (-1 stays -1): vocabularies below the two limits
  const int2 *bfast;
};
__device__ __forceinline__ int32_t step_id(const StepTable &st, int32_t raw) {
  return (st.packed && raw >= 0) ? (raw & ((1 << kStepIdBits) - 1)) : raw;
}
// (raw >= 0)
__device__ __forceinline__ int32_t step_len(const StepTable &st, int32_t raw, const int32_t *__restrict__ tok_len) {
  return st.packed ? (raw >> kStepIdBits) : tok_len[raw];
}

// index of the step containing SA slot r (pstart[0] == 0, so it always exists)
__device__ __forceinline__ int step_lookup(const StepTable &st, uint32_t r) {
  const uint32_t b = r >> st.shift;
  int lo = static_cast<int>(st.bidx[b]), hi = static_cast<int>(st.bidx[b + 1]);
  while (lo < hi) {  // first step of the bucket with start > r
    const int md = (lo + hi) >> 1;
    if (st.pstart[md] <= r) lo = md + 1; else hi = md;
  }
  return lo - 1;
}

// value of the step containing SA slot r, for a word-prefix position or not (linear.cpp:243-250's two arrays)
__device__ __forceinline__ int32_t step_raw(const StepTable &st, uint32_t r, bool prefix) {
  const int2 e = st.bfast[r >> st.shift];
  if (e.x >= -1) return prefix ? e.x : e.y;
  const int n = e.x & 0x7fffffff;
  int lo = e.y, hi = e.y + n;
  while (lo < hi) {  // first step of the bucket with start > r
    const int md = (lo + hi) >> 1;
    if (st.pstart[md] <= r) lo = md + 1; else hi = md;
  }
  return prefix ? st.pval_prefix[lo - 1] : st.pval_suffix[lo - 1];
}

// debug / parity: expand the step functions to the reference's per-slot arrays
__global__ __launch_bounds__(kBlock) void step_expand_kernel(StepTable st, size_t n, int32_t *__restrict__ best_prefix,
                                                             int32_t *__restrict__ best_suffix) {
  const size_t x = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (x >= n) return;
  const int k = step_lookup(st, static_cast<uint32_t>(x));
  best_prefix[x] = step_id(st, st.pval_prefix[k]);
  best_suffix[x] = step_id(st, st.pval_suffix[k]);
}

}  // namespace wp
