// Probe (never linked into the library): what does a wave-wide load cost on MI355X as a function of its width and of
// how its 64 addresses are spread?  The walk's token step is a handful of gathers (rank, class bytes, step table);
// its SQ counters say the waves wait 80 % of the time, and neither shorter chains of dependent loads nor more chains
// per lane moved its time — this measures whether the vector memory pipe (address processing / L1 tag lookups) is
// what saturates.  Every thread issues independent loads (UNROLL in flight), the result is XOR-folded so nothing is
// optimised away.  Reported: ns per wave-level load instruction per CU (all CUs busy, WAVES_PER_CU resident waves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Pattern { kCoalesced = 0, kStride24 = 1, kRandomSmall = 2, kRandomLarge = 3, kSame = 4, kStride8 = 5, kRandomMid = 6 };

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <typename T> __device__ __forceinline__ uint32_t fold(T v);
template <> __device__ __forceinline__ uint32_t fold<uint8_t>(uint8_t v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint32_t fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <> __device__ __forceinline__ uint32_t fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// bytes: size of the table in bytes (a power of two); every load is sizeof(T)-aligned
template <typename T, int PATTERN, int UNROLL>
__global__ __launch_bounds__(256) void gather(const uint8_t* __restrict__ table, size_t bytes, int iters, uint32_t* __restrict__ sink) {
  const uint32_t gt = blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wave = gt >> 6;
  const size_t mask = bytes - 1;
  uint32_t acc = 0;
  for (int it = 0; it < iters; it++) {
    T v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
      const uint32_t step = static_cast<uint32_t>(it) * UNROLL + u;
      size_t off;
      if (PATTERN == kCoalesced) off = (static_cast<size_t>(mix(wave * 977u + step)) * 4096 + lane * sizeof(T));
      else if (PATTERN == kStride24) off = (static_cast<size_t>(mix(wave * 977u + step)) * 4096 + lane * 24);
      else if (PATTERN == kStride8) off = (static_cast<size_t>(mix(wave * 977u + step)) * 4096 + lane * 8);
      else if (PATTERN == kSame) off = static_cast<size_t>(mix(wave * 977u + step)) * 4096;
      else off = static_cast<size_t>(mix(gt * 2654435761u + step * 40503u)) * 64;  // random line, per lane
      off &= mask & ~static_cast<size_t>(sizeof(T) - 1);
      v[u] = *reinterpret_cast<const T*>(table + off);
    }
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc ^= fold<T>(v[u]);
  }
  if (acc == 0x12345u) sink[0] = acc;  // (never true in practice: keeps the loads alive)
}

template <typename T, int PATTERN>
static int run(const char* name, const uint8_t* table, size_t bytes, uint32_t* sink, int cus, int wg_per_cu) {
  constexpr int UNROLL = 8;
  const int iters = 64;
  const dim3 grid(cus * wg_per_cu * 4);  // a few rounds of resident workgroups
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(gather<T, PATTERN, UNROLL>), grid, dim3(256), 0, 0, table, bytes, 4, sink);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(gather<T, PATTERN, UNROLL>), grid, dim3(256), 0, 0, table, bytes, iters, sink);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double wave_loads = static_cast<double>(grid.x) * 4 * iters * UNROLL;  // wave-level load instructions
  const double per_cu = wave_loads / cus;
  printf("%-34s %2zu B  %8.3f ms  %7.1f ns per wave-load per CU  (%5.2f lanes/clk/CU at 2.4 GHz, %6.1f GB/s useful)\n", name,
         sizeof(T), ms, ms * 1e6 / per_cu, 64.0 / (ms * 1e6 / per_cu * 2.4), wave_loads * 64 * sizeof(T) / (ms * 1e-3) / 1e9);
  return 0;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.name, cus);
  const size_t big = size_t(1) << 29;  // 512 MB
  uint8_t* table;
  uint32_t* sink;
  CK(hipMalloc(&table, big));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(table, 1, big));
  CK(hipMemset(sink, 0, 64));
  const int wg = 8;  // 32 waves per CU
#define RUN(T, P, NAME, BYTES) if (run<T, P>(NAME, table, BYTES, sink, cus, wg)) return 1
  RUN(uint32_t, kCoalesced, "coalesced (lane*4), 512 MB", big);
  RUN(uint4, kCoalesced, "coalesced (lane*16), 512 MB", big);
  RUN(uint32_t, kCoalesced, "coalesced (lane*4), 2 MB", size_t(1) << 21);
  RUN(uint32_t, kSame, "all lanes one address, 2 MB", size_t(1) << 21);
  RUN(uint8_t, kStride8, "stride 8 B, 2 MB", size_t(1) << 21);
  RUN(uint32_t, kStride8, "stride 8 B, 2 MB", size_t(1) << 21);
  RUN(uint32_t, kStride24, "stride 24 B, 2 MB", size_t(1) << 21);
  RUN(uint4, kStride24, "stride 24 B (unaligned 16), 2 MB", size_t(1) << 21);
  RUN(uint32_t, kStride24, "stride 24 B, 512 MB", big);
  RUN(uint8_t, kRandomSmall, "random line per lane, 2 MB", size_t(1) << 21);
  RUN(uint32_t, kRandomSmall, "random line per lane, 2 MB", size_t(1) << 21);
  RUN(uint2, kRandomSmall, "random line per lane, 2 MB", size_t(1) << 21);
  RUN(uint4, kRandomSmall, "random line per lane, 2 MB", size_t(1) << 21);
  RUN(uint32_t, kRandomMid, "random line per lane, 16 MB", size_t(1) << 24);
  RUN(uint2, kRandomMid, "random line per lane, 16 MB", size_t(1) << 24);
  RUN(uint32_t, kRandomMid, "random line per lane, 64 MB", size_t(1) << 26);
  RUN(uint32_t, kRandomLarge, "random line per lane, 512 MB", big);
  RUN(uint4, kRandomLarge, "random line per lane, 512 MB", big);
  return 0;
}
