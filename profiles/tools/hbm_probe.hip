// hbm_probe — what plain streaming kernels reach on this GPU (read, write, copy), to put the
// roofline fractions of bench.py in context.  Build: hipcc -O3 --offload-arch=gfx950 -o hbm_probe hbm_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kItems = 8;
__global__ __launch_bounds__(256) void read_kernel(const uint4 *__restrict__ in, size_t n, unsigned *__restrict__ out) {
  size_t base = (size_t)blockIdx.x * 256 * kItems + threadIdx.x;
  uint4 v[kItems];
#pragma unroll
  for (int j = 0; j < kItems; j++) { size_t i = base + (size_t)j * 256; v[j] = i < n ? in[i] : make_uint4(0, 0, 0, 0); }
  unsigned s = 0;
#pragma unroll
  for (int j = 0; j < kItems; j++) s += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
  if (s == 0x12345678u) out[0] = s;  // never true for the test pattern; keeps the loads alive
}
__global__ __launch_bounds__(256) void write_kernel(uint4 *__restrict__ out, size_t n) {
  size_t base = (size_t)blockIdx.x * 256 * kItems + threadIdx.x;
#pragma unroll
  for (int j = 0; j < kItems; j++) { size_t i = base + (size_t)j * 256; if (i < n) out[i] = make_uint4(i, 1, 2, 3); }
}
__global__ __launch_bounds__(256) void copy_kernel(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n) {
  size_t base = (size_t)blockIdx.x * 256 * kItems + threadIdx.x;
  uint4 v[kItems];
#pragma unroll
  for (int j = 0; j < kItems; j++) { size_t i = base + (size_t)j * 256; v[j] = i < n ? in[i] : make_uint4(0, 0, 0, 0); }
#pragma unroll
  for (int j = 0; j < kItems; j++) { size_t i = base + (size_t)j * 256; if (i < n) out[i] = v[j]; }
}

// the access shape of the radix kernels: a wave reads rows of 64 x 8 B, 20 rows per lane, 4 waves per tile
__global__ __launch_bounds__(256) void read_rows_kernel(const unsigned long long *__restrict__ in, size_t n,
                                                        unsigned *__restrict__ out) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t base = (size_t)blockIdx.x * 5120 + (size_t)w * 1280 + lane;
  unsigned long long v[20];
#pragma unroll
  for (int j = 0; j < 20; j++) { size_t i = base + (size_t)j * 64; v[j] = i < n ? in[i] : 0ull; }
  unsigned long long s = 0;
#pragma unroll
  for (int j = 0; j < 20; j++) s ^= v[j];
  if (s == 0x123456789abcdefull) out[0] = 1;
}

int main() {
  const size_t bytes = 1200ull << 20;  // 1.2 GB per buffer, like one radix pass over 1e8 records
  const size_t n = bytes / 16;
  uint4 *a, *b; unsigned *o;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&o, 4));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  const unsigned grid = (unsigned)((n + 256 * kItems - 1) / (256 * kItems));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[4] = {"read", "write", "copy (read+write)", "read, 64 x 8 B rows"};
  for (int k = 0; k < 4; k++) {
    float best = 1e9f;
    for (int it = 0; it < 7; it++) {
      CK(hipEventRecord(e0));
      if (k == 0) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, a, n, o);
      if (k == 1) hipLaunchKernelGGL(write_kernel, dim3(grid), dim3(256), 0, 0, b, n);
      if (k == 2) hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, 0, a, b, n);
      if (k == 3) hipLaunchKernelGGL(read_rows_kernel, dim3((unsigned)((bytes / 8 + 5119) / 5120)), dim3(256), 0, 0,
                                     reinterpret_cast<const unsigned long long *>(a), bytes / 8, o);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (it > 1 && ms < best) best = ms;
    }
    const double moved = (k == 2 ? 2.0 : 1.0) * bytes;
    printf("%-18s %.3f ms  %.0f GB/s\n", names[k], best, moved / 1e9 / (best / 1e3));
  }
  return 0;
}
