// Yardstick only (never linked into the library): how fast does rocPRIM's radix_sort_pairs order 1e8
// (uint32 key, uint32 value) records on this GPU?  Compared in DESIGN.md with the 4 passes of radix_sort.h.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void fill(uint32_t* k, uint32_t* v, size_t n, uint32_t mask) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = i * 0x9E3779B97F4A7C15ull + 0x1234567; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
  k[i] = (uint32_t)x & mask; v[i] = (uint32_t)i;
}

int main(int argc, char** argv) {
  size_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 100000003ull;
  uint32_t *k0, *k1, *v0, *v1;
  CK(hipMalloc(&k0, n * 4)); CK(hipMalloc(&k1, n * 4)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
  for (uint32_t mask : {0xffffffffu, 0x3f1f0f07u}) {
    size_t tmp_bytes = 0;
    CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, 32));
    void* tmp; CK(hipMalloc(&tmp, tmp_bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    for (int it = 0; it < 7; ++it) {
      fill<<<(unsigned)((n + 255) / 256), 256>>>(k0, v0, n, mask);
      CK(hipEventRecord(a));
      CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, 32));
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (it >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("rocprim radix_sort_pairs n=%zu mask=%08x tmp=%zu B: best %.3f ms, mean %.3f ms (%.2f Gkeys/s)\n",
           n, mask, tmp_bytes, best, sum / 5, n / best / 1e6);
    CK(hipFree(tmp));
  }
  return 0;
}
