// Probe (never linked into the library): in which order does one wave's ds_add_rtn_u32 serve lanes that hit the SAME
// LDS address?  The first pass of a sort ranks by one LDS atomic per key (radix_sort.h) and is 15 % faster than the
// stable passes' 8-ballot match-any; if the hardware served equal addresses in ascending lane order the atomic form
// would be stable too.  The ISA leaves the order undefined, so the library does NOT rely on it — this only records
// what gfx950 does, for whoever weighs a guarded use of it (self-test at context creation, match-any as the fallback).
// Every wave owns a row of 256 counters (as in the scatter kernel); per round a lane draws a digit from a skewed
// distribution, adds 1 and compares the returned value with its stable rank (counter before the round + lanes below it
// with the same digit).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void probe(int rounds, int spread, unsigned long long *mismatch, unsigned long long *conflicts) {
  __shared__ uint32_t cnt[4][256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int q = lane; q < 256; q += 64) cnt[w][q] = 0;
  __syncthreads();
  const uint32_t gw = blockIdx.x * 4 + w;
  unsigned long long bad = 0, conf = 0;
  for (int r = 0; r < rounds; r++) {
    const uint32_t d = mix(gw * 7919u + r * 104729u + lane * 2654435761u) % static_cast<uint32_t>(spread);  // few values: many conflicts
    // stable rank by match-any
    uint64_t peers = ~0ull;
    for (int b = 0; b < 8; b++) {
      const uint64_t m = __ballot((d >> b) & 1u);
      peers &= ((d >> b) & 1u) ? m : ~m;
    }
    const uint32_t below = __popcll(peers & ((1ull << lane) - 1ull));
    const uint32_t before = cnt[w][d];  // (read by all peers before anyone adds: the loads of a wave precede its atomic)
    __builtin_amdgcn_wave_barrier();
    const uint32_t got = atomicAdd(&cnt[w][d], 1u);
    __builtin_amdgcn_wave_barrier();
    if (got != before + below) bad++;
    if (__popcll(peers) > 1) conf++;
  }
  if (bad) atomicAdd(mismatch, bad);
  atomicAdd(conflicts, conf);
}

int main() {
  unsigned long long *d, h[2];
  CK(hipMalloc(&d, 16));
  for (int spread : {1, 2, 5, 17, 64, 256}) {
    CK(hipMemset(d, 0, 16));
    hipLaunchKernelGGL(probe, dim3(256 * 12), dim3(256), 0, 0, 2000, spread, d, d + 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("digits drawn from %3d values: %llu lane-rounds with a conflicting lane, %llu not served in ascending lane order\n", spread, h[1], h[0]);
  }
  return 0;
}
