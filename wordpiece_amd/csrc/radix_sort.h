// radix_sort.h — stable LSD radix sort of (key, uint32 value) pairs, 8-bit digits (WP_RADIX_BITS).
//
// One pass = histogram kernel + column scan (spine, apply) + scatter kernel.  A tile is 256
// threads x ITEMS keys.  Inside a tile each wave ranks its keys with a wave64 match-any (one ballot
// per digit bit) against per-wave digit counters staged in LDS, the tile is reordered through LDS
// and written back so that every digit run is a coalesced segment.  HBM traffic per pass and
// element: sizeof(Key) (histogram) + 2*(sizeof(Key)+4) (scatter).
//
// A single-pass ("onesweep", chained look-back) variant was built and measured slower on MI355X
// (0.93 ms vs 0.77 ms per 1e8-element pass): every look-back step is a cross-XCD miss and ~770
// tiles are in flight, so the chain costs more than the separate 8-byte histogram read.  It was
// removed again; see DESIGN.md.  9-bit digits (7 instead of 8 passes over the 63-bit keys) were
// measured too: each pass gets 13 % slower (512 bins: shorter runs, twice the counter work), the
// whole encode only 1 % faster, so 8 bits stay the default.
#pragma once
#include <cstdlib>
#include <string>
#include <vector>

#include "primitives.h"

namespace wp {

#ifndef WP_RADIX_BITS
#define WP_RADIX_BITS 8
#endif
constexpr int kRadixBits = WP_RADIX_BITS;
constexpr int kRadixBins = 1 << kRadixBits;
constexpr int kBinsPerThread = kRadixBins / kBlock;

template <typename KeyT>
struct RadixCfg {
#ifndef WP_RADIX_ITEMS64
#define WP_RADIX_ITEMS64 24  // 6144-key tiles: 74 KB of LDS staging, two workgroups per CU (16 / 20 / 24: 13.78 / 13.73 / 13.69 ms)
#endif
#ifndef WP_RADIX_ITEMS32
#define WP_RADIX_ITEMS32 20  // 5120-key tiles: 46 KB of LDS, three workgroups per CU (12 / 16 / 20 / 24: 0.402 / 0.403 / 0.389 / 0.418 ms per pass)
#endif
  static constexpr int kItems = sizeof(KeyT) == 8 ? WP_RADIX_ITEMS64 : WP_RADIX_ITEMS32;
  static constexpr int kTile = kBlock * kItems;
  // Inputs of up to kSmallN elements (mark / step lists, the large groups of a doubling round) use
  // tiles of kSmallItems per thread: a full-size tile takes ~70 us from first load to last store, and
  // with less than one tile per CU that latency is the whole pass.
  static constexpr int kSmallItems = 4;
  static constexpr int kSmallTile = kBlock * kSmallItems;
};
constexpr size_t kRadixSmallN = size_t(1) << 21;

// Lanes of the wave holding the same digit, as (mismatch_lo, mismatch_hi) complemented.  Per digit
// bit: one ballot, and "my bit differs from lane l's bit" = ballot ^ sign-extended(my bit), OR-ed
// into a per-lane mismatch mask (6 vector ops per bit).
template <int BITS>
__device__ __forceinline__ void wave_match_any(uint32_t digit, uint32_t &peers_lo, uint32_t &peers_hi) {
  uint32_t mlo = 0, mhi = 0;
#pragma unroll
  for (int b = 0; b < BITS; b++) {
    const int32_t ext = static_cast<int32_t>(digit << (31 - b)) >> 31;  // -1 if bit b is set, else 0
    const uint64_t m = __ballot(ext != 0);
    mlo |= static_cast<uint32_t>(m) ^ static_cast<uint32_t>(ext);
    mhi |= static_cast<uint32_t>(m >> 32) ^ static_cast<uint32_t>(ext);
  }
  peers_lo = ~mlo;
  peers_hi = ~mhi;
}

// Stable rank of this lane's key among the keys of its wave round: `cnt` is the wave's LDS counter
// row.  Every peer reads the counter, the lowest peer lane adds the peer count (the row is private
// to the wave and LDS operations of a wave execute in order).
template <int BITS>
__device__ __forceinline__ uint32_t wave_rank_digit(volatile uint32_t *cnt, uint32_t digit, int lane) {
  uint32_t plo, phi;
  wave_match_any<BITS>(digit, plo, phi);
  const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));  // peers in lower lanes
  const uint32_t total = __popc(plo) + __popc(phi);
  const uint32_t old = cnt[digit];
  if (below == 0) cnt[digit] = old + total;  // the lowest peer lane
  return old + below;
}

// Offset table, tile-major: table[tile * kRadixBins + digit].  The histogram kernel fills it with
// the per-tile digit counts (one coalesced row per workgroup) and adds the row into the sums
// of its chunk of kColChunk tiles; a single-workgroup spine turns the chunk sums into exclusive
// prefixes in (digit, tile) order; the apply kernel rewrites every row as global offsets.
constexpr int kColChunk = 64;

// dig != nullptr: the digits of this pass were written as one byte per key by the previous pass's
// scatter (or by the key builder): the kernel then reads 1 byte per key instead of sizeof(KeyT).
// Counting does not care which thread sees which key of the tile, so a thread takes ITEMS consecutive
// bytes (8-byte loads) and "round j" of a wave is byte j of every lane.
template <typename KeyT, int ITEMS>
__global__ __launch_bounds__(kBlock) void radix_hist_kernel(const KeyT *__restrict__ keys,
                                                            const uint8_t *__restrict__ dig, size_t n,
                                                            int begin_bit, uint32_t mask,
                                                            uint32_t *__restrict__ table,
                                                            uint32_t *__restrict__ chunk_sums, int lds_atomics) {
  constexpr int WAVES = kBlock / kWave;
  // per-wave counters, bumped once per distinct digit of a wave round by its lowest lane (match-any):
  // no LDS atomics, no same-address serialisation on skewed digits.  lds_atomics: one LDS atomic per
  // key instead — faster when the digit is close to uniform (the two lowest digits of the round-0
  // keys: 0.18-0.19 vs 0.21-0.22 ms), slower on the skewed high digits (0.23 vs 0.21 ms).
  // lds_atomics == 2: one LDS atomic per key into one of kHistCopies interleaved copies of the histogram
  // (copy = lane % kHistCopies, counter of digit d in copy c at d * kHistCopies + c: the lanes of a wave that
  // share a digit spread over kHistCopies addresses in kHistCopies different banks) — for skewed digits, where
  // a fifth of a wave's lanes can hold the same digit.
  constexpr int kHistCopies = 8;
  static_assert(kHistCopies >= WAVES, "the per-wave modes use the first WAVES rows");
  __shared__ uint32_t sh[kHistCopies][kRadixBins];
  uint32_t *flat = &sh[0][0];
#pragma unroll
  for (int i = 0; i < kHistCopies; i++) {
#pragma unroll
    for (int q = 0; q < kBinsPerThread; q++) sh[i][q * kBlock + threadIdx.x] = 0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  volatile uint32_t *mycnt = sh[w];
  const size_t tile_base = static_cast<size_t>(blockIdx.x) * (kBlock * ITEMS);
  const size_t base = tile_base + static_cast<size_t>(w) * (kWave * ITEMS);
  uint32_t dg[ITEMS];
  size_t idx0, idx_step;  // element index of round j: idx0 + j * idx_step
  if (dig) {
    static_assert(ITEMS % 4 == 0, "digit bytes are loaded as words");
    idx0 = tile_base + static_cast<size_t>(threadIdx.x) * ITEMS;
    idx_step = 1;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(dig + idx0);  // tile bases and ITEMS are multiples of 4
#pragma unroll
    for (int q = 0; q < ITEMS / 4; q++) {
      const uint32_t wv = idx0 + 4 * q < n ? src[q] : 0u;  // (the digit arrays are padded past n)
#pragma unroll
      for (int b = 0; b < 4; b++) dg[4 * q + b] = (wv >> (8 * b)) & 0xffu;
    }
  } else {
    idx0 = base + lane;
    idx_step = kWave;
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
      const size_t i = idx0 + static_cast<size_t>(j) * kWave;
      dg[j] = i < n ? (static_cast<uint32_t>(keys[i] >> begin_bit) & mask) : 0u;
    }
  }
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const size_t i = idx0 + static_cast<size_t>(j) * idx_step;
    const uint32_t d = dg[j];
    if (lds_atomics == 2) {  // (kernel argument: wave-uniform)
      if (i < n) atomicAdd(&flat[d * kHistCopies + (lane & (kHistCopies - 1))], 1u);
      continue;
    }
    if (lds_atomics) {
      if (i < n) atomicAdd(&sh[w][d], 1u);
      continue;
    }
    uint32_t plo, phi;
    wave_match_any<kRadixBits>(d, plo, phi);
    const uint64_t valid = __ballot(i < n);  // lanes past the end (last tile only) are not counted
    plo &= static_cast<uint32_t>(valid);
    phi &= static_cast<uint32_t>(valid >> 32);
    const uint32_t below = __builtin_amdgcn_mbcnt_hi(phi, __builtin_amdgcn_mbcnt_lo(plo, 0u));
    if (below == 0 && i < n) mycnt[d] = mycnt[d] + __popc(plo) + __popc(phi);
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kBinsPerThread; q++) {
    const int d = q * kBlock + threadIdx.x;
    uint32_t c = 0;
    if (lds_atomics == 2) {
#pragma unroll
      for (int i = 0; i < kHistCopies; i++) c += flat[d * kHistCopies + i];
    } else {
#pragma unroll
      for (int i = 0; i < WAVES; i++) c += sh[i][d];
    }
    table[static_cast<size_t>(blockIdx.x) * kRadixBins + d] = c;
    if (c) atomicAdd(&chunk_sums[static_cast<size_t>(blockIdx.x / kColChunk) * kRadixBins + d], c);
  }
}

// single workgroup of 1024 threads = kSpineParts chunk ranges x kRadixBins digits:
// chunk_pre[chunk][d] = exclusive prefix of chunk_sums in (digit, chunk) order
constexpr int kSpineThreads = 1024;
constexpr int kSpineParts = kSpineThreads / kRadixBins;
constexpr int kSpineBatch = 16;
__global__ __launch_bounds__(kSpineThreads) void radix_spine_kernel(const uint32_t *__restrict__ chunk_sums,
                                                                    uint32_t *__restrict__ chunk_pre,
                                                                    unsigned nchunks) {
  __shared__ uint32_t part[kSpineParts][kRadixBins];
  __shared__ uint32_t wtot[kRadixBins / kWave];
  const int d = threadIdx.x & (kRadixBins - 1), q = threadIdx.x / kRadixBins;
  const unsigned per = (nchunks + kSpineParts - 1) / kSpineParts;
  const unsigned c0 = min(nchunks, q * per), c1 = min(nchunks, c0 + per);
  // (loads issued kSpineBatch at a time: the loops are latency bound, one row per iteration)
  uint32_t sum = 0;
  for (unsigned c = c0; c < c1; c += kSpineBatch) {
    uint32_t v[kSpineBatch];
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) v[j] = c + j < c1 ? chunk_sums[static_cast<size_t>(c + j) * kRadixBins + d] : 0u;
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) sum += v[j];
  }
  part[q][d] = sum;
  __syncthreads();
  uint32_t before = 0, total = 0;
#pragma unroll
  for (int i = 0; i < kSpineParts; i++) {
    const uint32_t v = part[i][d];
    if (i < q) before += v;
    total += v;
  }
  // exclusive scan of total[d] over the digits (threads 0..kRadixBins-1 hold q == 0)
  uint32_t inc = wave_incl_sum(q == 0 ? total : 0u);
  if (q == 0 && (threadIdx.x & 63) == 63) wtot[threadIdx.x >> 6] = inc;
  __syncthreads();
  if (q == 0) {
    uint32_t base_d = inc - total;
    for (int i = 0; i < (d >> 6); i++) base_d += wtot[i];
    part[0][d] = base_d;  // every thread has finished reading part[][] (barrier above)
  }
  __syncthreads();
  uint32_t run = part[0][d] + before;
  for (unsigned c = c0; c < c1; c += kSpineBatch) {
    uint32_t v[kSpineBatch];
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) v[j] = c + j < c1 ? chunk_sums[static_cast<size_t>(c + j) * kRadixBins + d] : 0u;
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) {
      if (c + j < c1) chunk_pre[static_cast<size_t>(c + j) * kRadixBins + d] = run;
      run += v[j];
    }
  }
}

__global__ __launch_bounds__(kRadixBins) void radix_apply_kernel(uint32_t *__restrict__ table,
                                                                 const uint32_t *__restrict__ chunk_pre,
                                                                 unsigned ntiles) {
  const int d = threadIdx.x;
  const unsigned t0 = blockIdx.x * kColChunk, t1 = min(ntiles, t0 + kColChunk);
  uint32_t run = chunk_pre[static_cast<size_t>(blockIdx.x) * kRadixBins + d];
  for (unsigned t = t0; t < t1; t += kSpineBatch) {
    uint32_t v[kSpineBatch];
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) v[j] = t + j < t1 ? table[static_cast<size_t>(t + j) * kRadixBins + d] : 0u;
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) {
      if (t + j < t1) table[static_cast<size_t>(t + j) * kRadixBins + d] = run;
      run += v[j];
    }
  }
}

// Small inputs (<= kRadixSmallN elements: at most 32 chunks): spine and apply in one launch — every
// workgroup sums the few chunk rows itself (digit totals and the part in front of its own chunk), scans the
// 256 totals, and rewrites its chunk of the table.  One launch less per pass, and these sorts are
// bound by launch latency, not by bytes.
__global__ __launch_bounds__(kRadixBins) void radix_apply_small_kernel(uint32_t *__restrict__ table,
                                                                       const uint32_t *__restrict__ chunk_sums,
                                                                       unsigned nchunks, unsigned ntiles) {
  // (one thread per digit and a block scan of kBlock values: used with 8-bit digits only)
  __shared__ uint32_t ssum[8];
  const int d = threadIdx.x;
  uint32_t total = 0, before = 0;
  for (unsigned c = 0; c < nchunks; c++) {
    const uint32_t v = chunk_sums[static_cast<size_t>(c) * kRadixBins + d];
    total += v;
    if (c < blockIdx.x) before += v;
  }
  uint32_t all;
  uint32_t run = block_excl_sum(total, ssum, all) + before;
  const unsigned t0 = blockIdx.x * kColChunk, t1 = min(ntiles, t0 + kColChunk);
  for (unsigned t = t0; t < t1; t += kSpineBatch) {
    uint32_t v[kSpineBatch];
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) v[j] = t + j < t1 ? table[static_cast<size_t>(t + j) * kRadixBins + d] : 0u;
#pragma unroll
    for (int j = 0; j < kSpineBatch; j++) {
      if (t + j < t1) table[static_cast<size_t>(t + j) * kRadixBins + d] = run;
      run += v[j];
    }
  }
}

// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Mapping workgroup b to
// tile (b % 8) * (ntiles / 8) + b / 8 gives every XCD a contiguous range of tiles: consecutive tiles
// append to the same digit runs, so the partial 128-byte lines at the run ends meet in one L2 and
// leave it as full lines.  Placement only affects speed, never the result.
__device__ __forceinline__ unsigned xcd_tile(unsigned b, unsigned ntiles) {
  const unsigned q = ntiles / 8, r = ntiles % 8, x = b % 8;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
}

// STABLE = false: the first pass of a sort has no earlier order to keep — equal digits may leave the tile in any
// order — so a key takes its rank inside the wave from ONE LDS atomic on the wave's digit counter (the old value)
// instead of the 8-ballot match-any (about 40 VALU instructions per key; DESIGN.md: the ranking is half of a pass).
// vin == nullptr: the values are the element indices (the identity), made up here instead of read.
// RBITS: bits of the digit that can be set (the ranking takes one ballot per bit: a 6-bit digit — the second
// destination-partition pass of the rank store — ranks with 6 ballots instead of 8).
template <typename KeyT, int ITEMS, bool STABLE = true, int RBITS = kRadixBits>
__global__ __launch_bounds__(kBlock) void radix_scatter_kernel(
    const KeyT *__restrict__ kin, const uint32_t *__restrict__ vin, KeyT *__restrict__ kout,
    uint32_t *__restrict__ vout, size_t n, int begin_bit, uint32_t mask,
    const uint32_t *__restrict__ goff, uint8_t *__restrict__ dout, int next_bit, uint32_t next_mask, int dig_from_val) {
  // dout != nullptr: also leave the next pass's digit of every key as one byte at its new position,
  // so that the next histogram reads 1 byte per key instead of the key (a pass is the
  // 12-byte record read and written; this adds 1 + 1).  dig_from_val: the digit is taken from the
  // value (the sort that follows is keyed by the values: the destination partition of the rank store).
  constexpr int TILE = kBlock * ITEMS;
  constexpr int WAVES = kBlock / kWave;
  constexpr uint32_t kOob = (1u << RBITS) - 1u;  // bin of the slots past the end: the last one that can hold keys
  static_assert(STABLE || RBITS == kRadixBits, "the first-pass form counts the slots past the end in the top bin");
  __shared__ uint32_t wcnt[WAVES][kRadixBins];
  __shared__ uint32_t dstart[kRadixBins];
  __shared__ uint32_t gbase[kRadixBins];
  __shared__ uint32_t ssum[8];
  __shared__ KeyT skeys[TILE];
  __shared__ uint32_t svals[TILE];

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned tile = xcd_tile(blockIdx.x, gridDim.x);
  const size_t tile_base = static_cast<size_t>(tile) * TILE;
  const size_t wave_base = tile_base + static_cast<size_t>(w) * (kWave * ITEMS);
  const uint32_t tile_count = static_cast<uint32_t>(min(static_cast<size_t>(TILE), n - tile_base));

#pragma unroll
  for (int i = 0; i < WAVES; i++) {
#pragma unroll
    for (int q = 0; q < kBinsPerThread; q++) wcnt[i][q * kBlock + tid] = 0;
  }
  __syncthreads();

  KeyT key[ITEMS];
  uint32_t val[ITEMS];
  uint32_t rnk[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    size_t i = wave_base + static_cast<size_t>(r) * kWave + lane;
    const bool valid = i < n;
    key[r] = valid ? kin[i] : static_cast<KeyT>(~static_cast<KeyT>(0));
  }
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const size_t i = wave_base + static_cast<size_t>(r) * kWave + lane;
    val[r] = i < n ? (vin ? vin[i] : static_cast<uint32_t>(i)) : 0u;
  }
  volatile uint32_t *mycnt = wcnt[w];
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    size_t i = wave_base + static_cast<size_t>(r) * kWave + lane;
    // out-of-range slots (only at the very end of the last tile) take the top bin: they are the
    // last keys in tile order, hence rank after every valid key and are never written back
    const uint32_t d = i < n ? (static_cast<uint32_t>(key[r] >> begin_bit) & mask) : kOob;
    if (STABLE) {
      rnk[r] = wave_rank_digit<RBITS>(mycnt, d, lane);
    } else {
      // (out-of-range slots of the last tile share the top bin with real keys here: they must still rank behind
      // them, so they are counted after the loop)
      rnk[r] = i < n ? atomicAdd(&wcnt[w][d], 1u) : 0u;
    }
  }
  if (!STABLE && wave_base + static_cast<size_t>(ITEMS) * kWave > n) {  // (wave-uniform: the wave that holds the end)
#pragma unroll
    for (int r = 0; r < ITEMS; r++) {
      size_t i = wave_base + static_cast<size_t>(r) * kWave + lane;
      if (i >= n) rnk[r] = atomicAdd(&wcnt[w][kRadixBins - 1], 1u);
    }
  }
  __syncthreads();
  // thread t owns bins [t*kBinsPerThread, +kBinsPerThread): exclusive scan across waves, then bins
  uint32_t tot = 0, binbase[kBinsPerThread];
#pragma unroll
  for (int q = 0; q < kBinsPerThread; q++) {
    const int bin = tid * kBinsPerThread + q;
    uint32_t t = 0;
#pragma unroll
    for (int i = 0; i < WAVES; i++) {
      uint32_t c = wcnt[i][bin];
      wcnt[i][bin] = t;
      t += c;
    }
    binbase[q] = tot;
    tot += t;
  }
  uint32_t all;
  const uint32_t ex = block_excl_sum(tot, ssum, all);
#pragma unroll
  for (int q = 0; q < kBinsPerThread; q++) {
    const int bin = tid * kBinsPerThread + q;
    const uint32_t ds = ex + binbase[q];
    dstart[bin] = ds;
    gbase[bin] = goff[static_cast<size_t>(tile) * kRadixBins + bin] - ds;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    size_t i = wave_base + static_cast<size_t>(r) * kWave + lane;
    const uint32_t d = i < n ? (static_cast<uint32_t>(key[r] >> begin_bit) & mask) : kOob;
    const uint32_t pos = dstart[d] + wcnt[w][d] + rnk[r];
    skeys[pos] = key[r];
    svals[pos] = val[r];
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const uint32_t k = static_cast<uint32_t>(j) * kBlock + tid;
    if (k < tile_count) {
      const KeyT kk = skeys[k];
      const uint32_t d = static_cast<uint32_t>(kk >> begin_bit) & mask;
      const size_t o = static_cast<size_t>(gbase[d]) + k;
      if (!wp_in_bounds(o < n, kSiteRadixScatter)) continue;
      const uint32_t vv = svals[k];
      kout[o] = kk;
      vout[o] = vv;
      if (dout) dout[o] = static_cast<uint8_t>((dig_from_val ? vv >> next_bit : static_cast<uint32_t>(kk >> next_bit)) & next_mask);
    }
  }
}

// Brackets kernels with HIP events on their own stream without synchronising; resolve() is
// called once after the whole encode has been synchronised.
struct EventSpans {
  bool on = false;
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  void begin(hipStream_t st) {
    if (!on) return;
    if (used + 2 > ev.size()) {
      for (int i = 0; i < 64; i++) {
        hipEvent_t e;
        WP_HIP(hipEventCreate(&e));
        ev.push_back(e);
      }
    }
    WP_HIP(hipEventRecord(ev[used], st));
  }
  void end(hipStream_t st) {
    if (!on) return;
    WP_HIP(hipEventRecord(ev[used + 1], st));
    used += 2;
  }
  double resolve() {  // total ms over all spans; resets
    double tot = 0;
    for (size_t i = 0; i + 1 < used; i += 2) {
      float ms = 0;
      WP_HIP(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
      tot += ms;
    }
    used = 0;
    return tot;
  }
  ~EventSpans() {
    for (auto e : ev) (void)hipEventDestroy(e);
  }
};

struct RadixStats {
  int passes = 0;
  long long elems = 0;
  long long digit_bytes = 0;  // digit bytes written by scatter launches (1 per element and launch)
  long long bytes = 0;        // algorithmic bytes of the counted scatter launches: record read (without the index column
                              // when the pass makes it up) + record written + digit byte written
  EventSpans spans;  // around every scatter launch
};

constexpr int kMaxZeroedPasses = 8;  // chunk-sum tables cleared by one memset per sort

template <typename KeyT>
size_t radix_tmp_words(size_t n) {
  // (monotone in n: covers the small-tile configuration of every input size up to n as well)
  const size_t ntiles = std::max<size_t>(cdiv(n, RadixCfg<KeyT>::kTile),
                                         cdiv(std::min(n, kRadixSmallN), RadixCfg<KeyT>::kSmallTile));
  size_t h = ntiles * kRadixBins;
  return h + (kMaxZeroedPasses + 1) * (cdiv(ntiles, kColChunk) + 1) * kRadixBins + 64;
}

struct BitRange {
  int begin, end;
};

// Sorts the given bit ranges of the keys, least significant range first (stable LSD).  Data
// ping-pongs between (k0,v0) and (k1,v1); returns 0 or 1 = which pair holds the result.
// tmp: tmp_words uint32, at least radix_tmp_words<KeyT>(n) (checked: the table | chunk_pre |
// chunk_sums x kMaxZeroedPasses layout below must fit).
// identity_vals: the input values are 0..n-1 and v0 need not hold them (the first pass makes them up)
// uniform_low_bits: digits below this bit are close to uniformly distributed (histogram by LDS atomics)
// dg0/dg1 (optional, n + 64 bytes each, pairs with k0/k1): digit bytes — every scatter leaves the next
// pass's digits there; dg0_ready: dg0 already holds the first pass's digits (written by the key builder)
// histogram of a digit that is not known to be uniform: LDS atomics into interleaved copies of the counters (mode 2 of
// radix_hist_kernel; the match-any form, mode 0, serves the small sorts)
constexpr int kHistSkewMode = 2;

struct DigitBytes {
  uint8_t *dg0 = nullptr, *dg1 = nullptr;
  bool dg0_ready = false;
  // tail: the LAST pass also leaves digit bytes, for a sort that follows this one — bits [tail_bit, ...) & tail_mask
  // of the key, or of the value (tail_from_val).  They end up in tail_out(returned cur).
  int tail_bit = -1;
  uint32_t tail_mask = 0;
  bool tail_from_val = false;
  uint8_t *tail_out(int cur) const { return cur ? dg1 : dg0; }
};

// Where a sort keeps its offset tables inside `tmp`.  radix_plan() checks the capacity and clears the chunk sums; a
// caller that already knows the digits of the FIRST pass per tile (the key builder: its tiles are the sort's tiles)
// fills table / chunk_sums0 itself and hands the plan to the sort, which then skips that histogram launch.
struct RadixPlan {
  uint32_t *table = nullptr, *chunk_pre = nullptr, *chunk_sums0 = nullptr;
  size_t cs_words = 0;
  unsigned ntiles = 0, nchunks = 0;
  bool small = false;
};
template <typename KeyT>
RadixPlan radix_plan(size_t n, uint32_t *tmp, size_t tmp_words, hipStream_t st) {
  RadixPlan p;
  p.small = n <= kRadixSmallN;
  p.ntiles = cdiv(n, p.small ? RadixCfg<KeyT>::kSmallTile : RadixCfg<KeyT>::kTile);
  p.nchunks = cdiv(p.ntiles, kColChunk);
  const size_t h = static_cast<size_t>(p.ntiles) * kRadixBins;
  p.cs_words = static_cast<size_t>(p.nchunks + 1) * kRadixBins;
  if (h + (kMaxZeroedPasses + 1) * p.cs_words > tmp_words) {
    throw std::logic_error("radix sort: temporary buffer too small (" + std::to_string(tmp_words) + " words for " +
                           std::to_string(n) + " elements)");
  }
  p.table = tmp;
  p.chunk_pre = tmp + h;
  p.chunk_sums0 = p.chunk_pre + p.cs_words;
  // every pass adds into its own chunk-sum table; the first kMaxZeroedPasses are cleared at once
  WP_HIP(hipMemsetAsync(p.chunk_sums0, 0, sizeof(uint32_t) * p.cs_words * kMaxZeroedPasses, st));
  return p;
}

// first_hist: the plan of this sort, made by the caller, with the first pass's histogram already in it
template <typename KeyT>
int radix_sort_ranges(KeyT *k0, uint32_t *v0, KeyT *k1, uint32_t *v1, size_t n, const BitRange *ranges,
                      int nranges, uint32_t *tmp, size_t tmp_words, hipStream_t st, RadixStats *stats,
                      bool identity_vals = false, int uniform_low_bits = 0, DigitBytes db = DigitBytes(),
                      bool input_order_free = false, const RadixPlan *first_hist = nullptr) {
  // input_order_free: nothing depends on the order the input is in (a sort from scratch, NOT one pass of a sort that a
  // caller runs as several calls, like the second partition pass of the rank store): the first pass may rank by atomics
  int cur = 0;
  if (n == 0) return cur;
  const RadixPlan plan = first_hist ? *first_hist : radix_plan<KeyT>(n, tmp, tmp_words, st);
  const bool small = plan.small;
  if (small) {
    stats = nullptr;  // the roofline statistics describe the full-size configuration only
    db = DigitBytes();
  }
  const unsigned ntiles = plan.ntiles, nchunks = plan.nchunks;
  const size_t cs_words = plan.cs_words;
  uint32_t *table = plan.table, *chunk_pre = plan.chunk_pre, *chunk_sums0 = plan.chunk_sums0;
  struct Pass {
    int bit;
    uint32_t mask;
  };
  std::vector<Pass> passes;
  for (int r = 0; r < nranges; r++) {
    for (int b = ranges[r].begin; b < ranges[r].end; b += kRadixBits) {
      passes.push_back({b, (1u << std::min(kRadixBits, ranges[r].end - b)) - 1u});
    }
  }
  for (size_t pi = 0; pi < passes.size(); pi++) {
    const int pass = static_cast<int>(pi), b = passes[pi].bit;
    const uint32_t mask = passes[pi].mask;
    KeyT *ki = cur ? k1 : k0, *ko = cur ? k0 : k1;
    uint32_t *vi = cur ? v1 : v0, *vo = cur ? v0 : v1;
    uint32_t *chunk_sums = chunk_sums0 + cs_words * static_cast<size_t>(pass % kMaxZeroedPasses);
    if (pass >= kMaxZeroedPasses) WP_HIP(hipMemsetAsync(chunk_sums, 0, sizeof(uint32_t) * cs_words, st));
    // digit bytes: read where the previous pass (or the key builder) left them, written for the next pass
    const uint8_t *dgi = nullptr;
    if (db.dg0 && (pi > 0 || db.dg0_ready)) dgi = cur ? db.dg1 : db.dg0;
    const bool last = pi + 1 == passes.size();
    uint8_t *dgo = (db.dg0 && (!last || db.tail_bit >= 0)) ? (cur ? db.dg0 : db.dg1) : nullptr;
    const int nbit = !last ? passes[pi + 1].bit : std::max(db.tail_bit, 0);
    const uint32_t nmask = !last ? passes[pi + 1].mask : db.tail_mask;
    const int from_val = last && db.tail_from_val ? 1 : 0;
    if (first_hist && pi == 0) {
      // (the caller has taken this histogram)
    } else if (small) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_hist_kernel<KeyT, RadixCfg<KeyT>::kSmallItems>), dim3(ntiles),
                         dim3(kBlock), 0, st, ki, dgi, n, b, mask, table, chunk_sums, 0);
    } else {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_hist_kernel<KeyT, RadixCfg<KeyT>::kItems>), dim3(ntiles),
                         dim3(kBlock), 0, st, ki, dgi, n, b, mask, table, chunk_sums, b < uniform_low_bits ? 1 : kHistSkewMode);
    }
    if (small && kRadixBins == kBlock) {
      hipLaunchKernelGGL(radix_apply_small_kernel, dim3(nchunks), dim3(kRadixBins), 0, st, table, chunk_sums, nchunks, ntiles);
    } else {
      hipLaunchKernelGGL(radix_spine_kernel, dim3(1), dim3(kSpineThreads), 0, st, chunk_sums, chunk_pre, nchunks);
      hipLaunchKernelGGL(radix_apply_kernel, dim3(nchunks), dim3(kRadixBins), 0, st, table, chunk_pre, ntiles);
    }
    if (stats) stats->spans.begin(st);
    const bool made_up_index = identity_vals;
    const uint32_t *vsrc = identity_vals ? static_cast<const uint32_t *>(nullptr) : vi;
    if (small) {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_scatter_kernel<KeyT, RadixCfg<KeyT>::kSmallItems, true>), dim3(ntiles),
                         dim3(kBlock), 0, st, ki, vsrc, ko, vo, n, b, mask, table, dgo, nbit, nmask, from_val);
    } else if (pi == 0 && input_order_free) {  // (no earlier order to keep: ranks by LDS atomics)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_scatter_kernel<KeyT, RadixCfg<KeyT>::kItems, false>), dim3(ntiles),
                         dim3(kBlock), 0, st, ki, vsrc, ko, vo, n, b, mask, table, dgo, nbit, nmask, from_val);
    } else if (sizeof(KeyT) == 4 && mask < 64u) {  // (a digit of at most 6 bits)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_scatter_kernel<KeyT, RadixCfg<KeyT>::kItems, true, 6>), dim3(ntiles),
                         dim3(kBlock), 0, st, ki, vsrc, ko, vo, n, b, mask, table, dgo, nbit, nmask, from_val);
    } else {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_scatter_kernel<KeyT, RadixCfg<KeyT>::kItems, true>), dim3(ntiles),
                         dim3(kBlock), 0, st, ki, vsrc, ko, vo, n, b, mask, table, dgo, nbit, nmask, from_val);
    }
    identity_vals = false;
    WP_LAUNCH_CHECK();
    if (stats) {
      stats->spans.end(st);
      stats->passes++;
      stats->elems += static_cast<long long>(n);
      if (dgo) stats->digit_bytes += static_cast<long long>(n);
      stats->bytes += static_cast<long long>(n) * static_cast<long long>(2 * sizeof(KeyT) + 4 + (made_up_index ? 0 : 4) + (dgo ? 1 : 0));
    }
    cur ^= 1;
  }
  return cur;
}

template <typename KeyT>
int radix_sort_pairs(KeyT *k0, uint32_t *v0, KeyT *k1, uint32_t *v1, size_t n, int begin_bit, int end_bit,
                     uint32_t *tmp, size_t tmp_words, hipStream_t st, RadixStats *stats, bool identity_vals = false,
                     int uniform_low_bits = 0, DigitBytes db = DigitBytes(), bool input_order_free = false,
                     const RadixPlan *first_hist = nullptr) {
  BitRange r{begin_bit, end_bit};
  return radix_sort_ranges<KeyT>(k0, v0, k1, v1, n, &r, 1, tmp, tmp_words, st, stats, identity_vals,
                                 uniform_low_bits, db, input_order_free, first_hist);
}

}  // namespace wp
