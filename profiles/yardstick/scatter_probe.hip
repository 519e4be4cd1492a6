// Probe (never linked into the library): what does a 4-byte permutation scatter cost on MI355X once the list has
// been partitioned by the top bits of the destination, as a function of the window size, the entries in flight
// and the cache policy of the streamed loads?  Feeds the rank-store design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// entry k of the partitioned list: window k >> wbits, a bijective scramble of the low bits inside it
__global__ void make_list(uint32_t* dst, uint32_t* val, size_t n, int wbits) {
  size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k >= n) return;
  const uint32_t mask = (1u << wbits) - 1;
  uint32_t x = (uint32_t)k & mask;
  x = (x * 0x9E3779B1u + 0x7F4A7C15u) & mask; x ^= x >> (wbits / 2 + 1);
  x = (x * 0x85EBCA6Bu + 0xC2B2AE35u) & mask; x ^= x >> (wbits / 2);
  x = (x * 0x27D4EB2Fu) & mask;  // odd multipliers and xorshifts: bijections on wbits bits
  dst[k] = ((uint32_t)(k >> wbits) << wbits) | x;
  val[k] = (uint32_t)k;
}

__global__ void check(const uint32_t* dst, const uint32_t* out, size_t n, unsigned* bad) {
  size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k < n && out[dst[k]] != (uint32_t)k) atomicAdd(bad, 1u);
}

template <int ITEMS, int NTLOAD, int NTSTORE>
__global__ __launch_bounds__(256) void scatter(const uint32_t* __restrict__ dst, const uint32_t* __restrict__ val, size_t m,
                                               uint32_t* __restrict__ out, int xcd) {
  extern __shared__ uint32_t pad[];  // only to limit the workgroups per CU
  unsigned b = blockIdx.x;
  if (xcd) {
    const unsigned nb = gridDim.x, q = nb / 8, r = nb % 8, x = b % 8;
    b = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + b / 8;
  }
  const size_t base = (size_t)b * (256 * ITEMS) + threadIdx.x;
  uint32_t d[ITEMS], v[ITEMS];
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const size_t k = base + (size_t)j * 256;
    if (k < m) {
      d[j] = NTLOAD ? __builtin_nontemporal_load(dst + k) : dst[k];
      v[j] = NTLOAD ? __builtin_nontemporal_load(val + k) : val[k];
    }
  }
#pragma unroll
  for (int j = 0; j < ITEMS; j++) {
    const size_t k = base + (size_t)j * 256;
    if (k < m) {
      if (NTSTORE) __builtin_nontemporal_store(v[j], out + d[j]); else out[d[j]] = v[j];
    }
  }
}

// window-in-LDS variant: one workgroup owns one window of 2^wbits slots (wbits <= 15: 128 KB), streams the
// window's entries, places them in LDS and writes the window back with full-width stores
template <int WB>
__global__ __launch_bounds__(1024) void scatter_lds(const uint32_t* __restrict__ dst, const uint32_t* __restrict__ val,
                                                    size_t m, uint32_t* __restrict__ out) {
  extern __shared__ uint32_t win[];
  const size_t w0 = (size_t)blockIdx.x << WB;
  const size_t cnt = w0 + (1u << WB) <= m ? (1u << WB) : (m > w0 ? m - w0 : 0);
  for (size_t k = threadIdx.x * 4; k < cnt; k += 1024 * 4) {
    const uint4 d = *reinterpret_cast<const uint4*>(dst + w0 + k);
    const uint4 v = *reinterpret_cast<const uint4*>(val + w0 + k);
    win[d.x - w0] = v.x; win[d.y - w0] = v.y; win[d.z - w0] = v.z; win[d.w - w0] = v.w;
  }
  __syncthreads();
  for (size_t k = threadIdx.x * 4; k < cnt; k += 1024 * 4)
    *reinterpret_cast<uint4*>(out + w0 + k) = *reinterpret_cast<const uint4*>(win + k);
}


// partition windows of 2^pb slots, LDS windows of 2^WB: the 2^(pb-WB) workgroups of a partition window all stream
// its entries (from the L2 of their XCD, mostly) and keep what falls into their own LDS window
template <int WB>
__global__ __launch_bounds__(1024) void scatter_filter(const uint32_t* __restrict__ dst, const uint32_t* __restrict__ val,
                                                       size_t m, uint32_t* __restrict__ out, int pb) {
  extern __shared__ uint32_t win[];
  const unsigned r = 1u << (pb - WB), x = blockIdx.x % 8, t = blockIdx.x / 8;
  const size_t w = (size_t)(t / r) * 8 + x, s = t % r;
  const size_t e0 = w << pb, cnt = (size_t)1 << pb;
  const uint32_t base = (uint32_t)((w << pb) + (s << WB));
  for (size_t k = threadIdx.x * 4; k < cnt; k += 1024 * 4) {
    const uint4 d = *reinterpret_cast<const uint4*>(dst + e0 + k);
    const uint4 v = *reinterpret_cast<const uint4*>(val + e0 + k);
    if (d.x - base < (1u << WB)) win[d.x - base] = v.x;
    if (d.y - base < (1u << WB)) win[d.y - base] = v.y;
    if (d.z - base < (1u << WB)) win[d.z - base] = v.z;
    if (d.w - base < (1u << WB)) win[d.w - base] = v.w;
  }
  __syncthreads();
  for (size_t k = threadIdx.x * 4; k < ((size_t)1 << WB); k += 1024 * 4)
    *reinterpret_cast<uint4*>(out + base + k) = *reinterpret_cast<const uint4*>(win + k);
}

int main() {
  const size_t n = (size_t)1 << 27;
  uint32_t *dst, *val, *out; unsigned* bad;
  CK(hipMalloc(&dst, n * 4)); CK(hipMalloc(&val, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&bad, 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, int wbits, auto launch) -> int {
    make_list<<<(unsigned)(n / 256), 256>>>(dst, val, n, wbits);
    CK(hipMemset(out, 0xff, n * 4)); CK(hipMemset(bad, 0, 4));
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
    }
    check<<<(unsigned)(n / 256), 256>>>(dst, out, n, bad);
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("%-44s window 2^%-2d slots (%7.0f KB): %7.3f ms  %6.2f Gentry/s  bad %u\n", name, wbits, 4.0 * (1u << wbits) / 1024,
           best, n / best / 1e6, hb);
    fflush(stdout);
    return 0;
  };
#define V(ITEMS, NTL, NTS, LDSB, XCD, WB)                                                                         \
  if (run("items " #ITEMS " ntload " #NTL " ntstore " #NTS " lds " #LDSB " xcd " #XCD, WB, [&] {                 \
        scatter<ITEMS, NTL, NTS><<<(unsigned)(n / (256 * ITEMS)), 256, LDSB>>>(dst, val, n, out, XCD); })) return 1;
  for (int wb : {19}) { V(1, 0, 0, 0, 1, wb) }
  for (int wb : {19}) {
    V(1, 1, 0, 0, 1, wb) V(1, 0, 1, 0, 1, wb) V(1, 1, 1, 0, 1, wb)
    V(1, 0, 0, 0, 0, wb) V(2, 0, 0, 0, 1, wb) V(4, 1, 0, 0, 1, wb)
    V(1, 0, 0, 16384, 1, wb) V(1, 0, 0, 40000, 1, wb) V(1, 1, 0, 40000, 1, wb) V(1, 0, 0, 65536, 1, wb)
  }
  CK(hipFuncSetAttribute((const void*)scatter_lds<15>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 15));
  if (run("window in LDS, 1024 threads", 15, [&] { scatter_lds<15><<<(unsigned)(n >> 15), 1024, 4 << 15>>>(dst, val, n, out); })) return 1;
  if (run("window in LDS, 1024 threads", 14, [&] { scatter_lds<14><<<(unsigned)(n >> 14), 1024, 4 << 14>>>(dst, val, n, out); })) return 1;
  if (run("window in LDS, 1024 threads", 13, [&] { scatter_lds<13><<<(unsigned)(n >> 13), 1024, 4 << 13>>>(dst, val, n, out); })) return 1;
  CK(hipFuncSetAttribute((const void*)scatter_filter<15>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 15));
  for (int pb : {16, 17, 18, 19, 20})
    if (run("filter into LDS window 2^15, 1024 threads", pb, [&] { scatter_filter<15><<<(unsigned)(n >> 15), 1024, 4 << 15>>>(dst, val, n, out, pb); })) return 1;
  CK(hipFuncSetAttribute((const void*)scatter_filter<14>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 << 14));
  for (int pb : {17, 18, 19})
    if (run("filter into LDS window 2^14 (2 WG/CU), 1024 thr", pb, [&] { scatter_filter<14><<<(unsigned)(n >> 14), 1024, 4 << 14>>>(dst, val, n, out, pb); })) return 1;
  return 0;
}
