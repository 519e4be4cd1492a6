// format.h — the reference's id file format on the device: decimal id followed by one space, no
// newline (utils.cpp:30-35 writeToFile, linear.cpp:367-370).  The reference formats with
// `fout << id << ' '` on one core (≈ 21 M ids per 100 MB of text, i.e. seconds); here a tile of ids
// is converted in LDS and leaves as coalesced bytes: count -> scan -> write, like the id compaction.
#pragma once
#include "primitives.h"

namespace wp {

constexpr int kFmtItems = 8;
constexpr int kFmtTile = kBlock * kFmtItems;  // 2048 ids per workgroup
constexpr int kFmtMaxLen = 12;                // "-2147483648 "

__device__ __forceinline__ int fmt_len(int32_t v) {  // characters of "<v> "
  uint32_t a = v < 0 ? 0u - static_cast<uint32_t>(v) : static_cast<uint32_t>(v);
  int d = 1;
  d += a >= 10u;
  d += a >= 100u;
  d += a >= 1000u;
  d += a >= 10000u;
  d += a >= 100000u;
  d += a >= 1000000u;
  d += a >= 10000000u;
  d += a >= 100000000u;
  d += a >= 1000000000u;
  return d + (v < 0) + 1;
}

__global__ __launch_bounds__(kBlock) void fmt_count_kernel(const int32_t *__restrict__ ids, size_t n,
                                                           uint32_t *__restrict__ tile_bytes) {
  __shared__ uint32_t sm[8];
  const size_t base = static_cast<size_t>(blockIdx.x) * kFmtTile;
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < kFmtItems; j++) {
    const size_t i = base + static_cast<size_t>(j) * kBlock + threadIdx.x;
    if (i < n) c += fmt_len(ids[i]);
  }
  uint32_t tot;
  (void)block_excl_sum(c, sm, tot);
  if (threadIdx.x == 0) tile_bytes[blockIdx.x] = tot;
}

// tile_off: exclusive prefix of tile_bytes as 64-bit offsets (a 2 GB batch can exceed 4 GB of text)
__global__ __launch_bounds__(kBlock) void fmt_write_kernel(const int32_t *__restrict__ ids, size_t n,
                                                           const unsigned long long *__restrict__ tile_off,
                                                           char *__restrict__ out) {
  __shared__ uint32_t sm[8];
  __shared__ char sbuf[kFmtTile * kFmtMaxLen];
  const size_t base = static_cast<size_t>(blockIdx.x) * kFmtTile + static_cast<size_t>(threadIdx.x) * kFmtItems;
  int32_t v[kFmtItems];
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < kFmtItems; j++) {
    v[j] = base + j < n ? ids[base + j] : 0;
    if (base + j < n) c += fmt_len(v[j]);
  }
  uint32_t tot;
  uint32_t o = block_excl_sum(c, sm, tot);
#pragma unroll
  for (int j = 0; j < kFmtItems; j++) {
    if (base + j < n) {
      const int len = fmt_len(v[j]);
      uint32_t a = v[j] < 0 ? 0u - static_cast<uint32_t>(v[j]) : static_cast<uint32_t>(v[j]);
      int p = static_cast<int>(o) + len - 1;
      sbuf[p--] = ' ';
      do {
        sbuf[p--] = static_cast<char>('0' + a % 10u);
        a /= 10u;
      } while (a);
      if (v[j] < 0) sbuf[p] = '-';
      o += len;
    }
  }
  __syncthreads();
  char *dst = out + tile_off[blockIdx.x];
  for (uint32_t k = threadIdx.x; k < tot; k += kBlock) dst[k] = sbuf[k];
}

// 64-bit exclusive scan of the tile byte counts (single workgroup; tiles = n / 2048)
__global__ __launch_bounds__(1024) void fmt_offsets_kernel(const uint32_t *__restrict__ tile_bytes, size_t tiles,
                                                           unsigned long long *__restrict__ tile_off,
                                                           unsigned long long *__restrict__ total) {
  __shared__ unsigned long long ws[16];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  unsigned long long carry = 0;
  for (size_t base = 0; base < tiles; base += 1024) {
    const size_t i = base + threadIdx.x;
    const unsigned long long v = i < tiles ? tile_bytes[i] : 0ull;
    unsigned long long inc = v;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const unsigned long long t = __shfl_up(inc, d, kWave);
      if (lane >= d) inc += t;
    }
    if (lane == kWave - 1) ws[w] = inc;
    __syncthreads();
    unsigned long long before = carry, all = carry;
    for (int q = 0; q < 16; q++) {
      if (q < w) before += ws[q];
      all += ws[q];
    }
    if (i < tiles) tile_off[i] = before + inc - v;
    carry = all;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

}  // namespace wp
