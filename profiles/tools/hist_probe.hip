// hist_probe — radix_hist_kernel / radix_scatter_kernel<uint64> in isolation on random keys.
// Build: hipcc -O3 --offload-arch=gfx950 -I wordpiece_amd/csrc -o hist_probe profiles/tools/hist_probe.hip
#include "radix_sort.h"
#include <cstdio>
#include <vector>
#include <random>
using namespace wp;
int main() {
  const size_t n = 100174216;
  std::vector<uint64_t> h(n);
  std::mt19937_64 rng(1);
  for (auto &x : h) x = rng() >> 1;
  uint64_t *k0, *k1; uint32_t *v0, *v1, *tmp;
  WP_HIP(hipMalloc(&k0, n * 8)); WP_HIP(hipMalloc(&k1, n * 8));
  WP_HIP(hipMalloc(&v0, n * 4)); WP_HIP(hipMalloc(&v1, n * 4));
  WP_HIP(hipMalloc(&tmp, radix_tmp_words<uint64_t>(n) * 4));
  WP_HIP(hipMemcpy(k0, h.data(), n * 8, hipMemcpyHostToDevice));
  WP_HIP(hipMemset(v0, 0, n * 4));
  const unsigned ntiles = cdiv(n, RadixCfg<uint64_t>::kTile);
  hipEvent_t e0, e1; WP_HIP(hipEventCreate(&e0)); WP_HIP(hipEventCreate(&e1));
  for (int bit = 0; bit < 64; bit += 24) {
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
      WP_HIP(hipMemsetAsync(tmp, 0, radix_tmp_words<uint64_t>(n) * 4, 0));
      WP_HIP(hipEventRecord(e0));
      hipLaunchKernelGGL(HIP_KERNEL_NAME(radix_hist_kernel<uint64_t, RadixCfg<uint64_t>::kItems>), dim3(ntiles), dim3(kBlock), 0, 0, k0,
                         static_cast<const uint8_t *>(nullptr), n, bit, 255u, tmp, tmp + (size_t)ntiles * kRadixBins, 0);
      WP_HIP(hipEventRecord(e1)); WP_HIP(hipEventSynchronize(e1));
      float ms; WP_HIP(hipEventElapsedTime(&ms, e0, e1));
      if (it > 0 && ms < best) best = ms;
    }
    printf("hist bit %2d: %.3f ms  %.0f GB/s\n", bit, best, n * 8.0 / 1e9 / (best / 1e3));
  }
#ifdef WP_HIST_NOCOUNT
  return 0;
#endif
  RadixStats st;
  st.spans.on = true;
  for (int it = 0; it < 3; it++) {
    WP_HIP(hipEventRecord(e0));
    radix_sort_pairs<uint64_t>(k0, v0, k1, v1, n, 0, 16, tmp, radix_tmp_words<uint64_t>(n), 0, &st);
    WP_HIP(hipEventRecord(e1)); WP_HIP(hipEventSynchronize(e1));
    float ms; WP_HIP(hipEventElapsedTime(&ms, e0, e1));
    printf("2 passes: %.3f ms total, scatter spans %.3f ms\n", ms, st.spans.resolve());
  }
  return 0;
}
