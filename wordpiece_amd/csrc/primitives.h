// primitives.h — wave64 / block / device-wide scan building blocks.
#pragma once
#include "common.h"

namespace wp {

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

// ---- wave-level inclusive scans (64 lanes, shuffle based) ------------------------------
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    uint32_t t = __shfl_up(v, d, kWave);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ int32_t wave_incl_max(int32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    int32_t t = __shfl_up(v, d, kWave);
    if (lane >= d) v = max(v, t);
  }
  return v;
}
__device__ __forceinline__ int32_t wave_incl_min(int32_t v) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) {
    int32_t t = __shfl_up(v, d, kWave);
    if (lane >= d) v = min(v, t);
  }
  return v;
}
__device__ __forceinline__ int32_t wave_reduce_min(int32_t v) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) v = min(v, __shfl_xor(v, d, kWave));
  return v;
}
__device__ __forceinline__ uint32_t wave_reduce_sum(uint32_t v) {
#pragma unroll
  for (int d = kWave / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, kWave);
  return v;
}

// Workgroup barrier that only orders LDS traffic.  __syncthreads() also waits for the wave's
// outstanding global stores (vmcnt(0)), which costs a memory round trip per loop iteration in the
// single-workgroup "spine" loops that store a result and then reuse LDS scratch.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// ---- block-level (kBlock = 256 threads = 4 waves) -----------------------------------------
// Exclusive sum over the block; `total` = block sum.  smem: >= 8 uint32.
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t *smem, uint32_t &total) {
  const int lane = lane_id(), w = wave_id();
  uint32_t inc = wave_incl_sum(v);
  __syncthreads();  // protect smem reuse across calls
  if (lane == kWave - 1) smem[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kBlock / kWave; i++) {
    uint32_t s = smem[i];
    if (i < w) base += s;
    tot += s;
  }
  total = tot;
  return base + inc - v;
}
// Inclusive max over the block (running maximum in thread order).  smem: >= 8 int32.
__device__ __forceinline__ int32_t block_incl_max(int32_t v, int32_t *smem) {
  const int lane = lane_id(), w = wave_id();
  int32_t inc = wave_incl_max(v);
  __syncthreads();
  if (lane == kWave - 1) smem[w] = inc;
  __syncthreads();
  int32_t r = inc;
#pragma unroll
  for (int i = 0; i < kBlock / kWave; i++) {
    if (i < w) r = max(r, smem[i]);
  }
  return r;
}
__device__ __forceinline__ int32_t block_reduce_min(int32_t v, int32_t *smem) {
  v = wave_reduce_min(v);
  __syncthreads();
  if (lane_id() == 0) smem[wave_id()] = v;
  __syncthreads();
  int32_t r = smem[0];
#pragma unroll
  for (int i = 1; i < kBlock / kWave; i++) r = min(r, smem[i]);
  return r;
}

// ---- device-wide exclusive scan of uint32 (three launches, tiles of 2048) ------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;

// n_dev (optional): device-side element count <= n; tiles past it are skipped (their sum is 0)
__global__ __launch_bounds__(kBlock) void scan_reduce_kernel(const uint32_t *__restrict__ in, size_t n,
                                                             uint32_t *__restrict__ tile_sums,
                                                             const uint32_t *__restrict__ n_dev) {
  __shared__ uint32_t sm[8];
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile;
  if (n_dev) {
    n = min(n, static_cast<size_t>(*n_dev));
    if (base >= n) {
      if (threadIdx.x == 0) tile_sums[blockIdx.x] = 0;
      return;
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; j++) {
    size_t i = base + static_cast<size_t>(j) * kBlock + threadIdx.x;
    if (i < n) s += in[i];
  }
  s = wave_reduce_sum(s);
  if (lane_id() == 0) sm[wave_id()] = s;
  __syncthreads();
  if (threadIdx.x == 0) tile_sums[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

// single block: in-place exclusive scan of m values; writes the grand total to *total
// (n_dev, optional: only the tiles that hold elements below *n_dev are non-zero and visited)
// (total64, optional: the same sum without wrapping — the code point count of inputs beyond 4 GB)
__global__ __launch_bounds__(kBlock) void scan_spine_kernel(uint32_t *__restrict__ sums, size_t m,
                                                            uint32_t *__restrict__ total,
                                                            const uint32_t *__restrict__ n_dev,
                                                            unsigned long long *__restrict__ total64) {
  __shared__ uint32_t sm[8];
  if (n_dev) m = min(m, (static_cast<size_t>(*n_dev) + kScanTile - 1) / kScanTile);
  uint32_t carry = 0;
  unsigned long long carry64 = 0;
  for (size_t base = 0; base < m; base += kBlock) {
    size_t i = base + threadIdx.x;
    uint32_t v = i < m ? sums[i] : 0, tot;
    uint32_t ex = block_excl_sum(v, sm, tot);
    if (i < m) sums[i] = carry + ex;
    carry += tot;
    carry64 += tot;
  }
  if (threadIdx.x == 0 && total) *total = carry;
  if (threadIdx.x == 0 && total64) *total64 = carry64;
}

// (in == out is allowed — most callers scan in place — so the two carry no __restrict__)
__global__ __launch_bounds__(kBlock) void scan_apply_kernel(const uint32_t *in, uint32_t *out, size_t n,
                                                            const uint32_t *__restrict__ tile_prefix,
                                                            const uint32_t *__restrict__ n_dev) {
  __shared__ uint32_t sm[8];
  if (n_dev) {
    n = min(n, static_cast<size_t>(*n_dev));
    if (static_cast<size_t>(blockIdx.x) * kScanTile >= n) return;
  }
  const size_t base = static_cast<size_t>(blockIdx.x) * kScanTile + static_cast<size_t>(threadIdx.x) * kScanItems;
  uint32_t v[kScanItems], s = 0;
#pragma unroll
  for (int j = 0; j < kScanItems; j++) {
    size_t i = base + j;
    v[j] = i < n ? in[i] : 0;
    s += v[j];
  }
  uint32_t tot;
  uint32_t ex = block_excl_sum(s, sm, tot) + tile_prefix[blockIdx.x];
#pragma unroll
  for (int j = 0; j < kScanItems; j++) {
    size_t i = base + j;
    if (i < n) out[i] = ex;
    ex += v[j];
  }
}

// tmp must hold cdiv(n, kScanTile) + 1 uint32; `total` (device pointer, optional) gets the sum.
inline void device_exclusive_scan(const uint32_t *in, uint32_t *out, size_t n, uint32_t *tmp,
                                  uint32_t *total, hipStream_t st, const uint32_t *n_dev = nullptr,
                                  unsigned long long *total64 = nullptr) {
  if (n == 0) {
    if (total) WP_HIP(hipMemsetAsync(total, 0, sizeof(uint32_t), st));
    if (total64) WP_HIP(hipMemsetAsync(total64, 0, sizeof(unsigned long long), st));
    return;
  }
  unsigned tiles = cdiv(n, kScanTile);
  hipLaunchKernelGGL(scan_reduce_kernel, dim3(tiles), dim3(kBlock), 0, st, in, n, tmp, n_dev);
  hipLaunchKernelGGL(scan_spine_kernel, dim3(1), dim3(kBlock), 0, st, tmp, static_cast<size_t>(tiles), total, n_dev,
                     total64);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(tiles), dim3(kBlock), 0, st, in, out, n, tmp, n_dev);
  WP_LAUNCH_CHECK();
}

}  // namespace wp
