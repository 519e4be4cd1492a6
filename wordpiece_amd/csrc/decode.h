// decode.h — UTF-8 decode, character classes, alphabet compaction and the S = text·1·vocab build.
//
// Replaces utils::parseText / vkcom::decode_utf8 (utils.cpp:37-79, utf8.cpp:54-90,130-147) and the
// S build of linear.cpp:77-103.  Decoding is data-parallel: a continuation byte never yields a code
// point (it is consumed by a valid lead byte or dropped as invalid), and a non-continuation byte
// is always a decode point (a valid sequence only ever swallows continuation bytes), so every
// byte is classified independently and valid leads are stream-compacted.
#pragma once
#include "code.h"
#include "primitives.h"

namespace wp {

constexpr int kDecBytes = 16;
constexpr int kDecTile = kBlock * kDecBytes;  // 4096 input bytes per workgroup
constexpr uint32_t kCpTableSize = 0x110000;   // used[] / lut[] cover every code point

// loads this thread's 16 bytes + 4 halo bytes into b[20] (zero padded past nbytes)
__device__ __forceinline__ void load_bytes20(const uint8_t *__restrict__ text, size_t nbytes, size_t off,
                                             uint8_t (&b)[20]) {
  uint32_t w[5];
  const size_t nwords = (nbytes + 3) / 4;  // the buffer is allocated padded to 16 bytes
  const uint32_t *t32 = reinterpret_cast<const uint32_t *>(text);
#pragma unroll
  for (int k = 0; k < 5; k++) {
    size_t wi = off / 4 + k;
    w[k] = wi < nwords ? t32[wi] : 0u;
  }
#pragma unroll
  for (int k = 0; k < 20; k++) {
    uint32_t v = (w[k / 4] >> (8 * (k % 4))) & 0xffu;
    b[k] = (off + k < nbytes) ? static_cast<uint8_t>(v) : 0;
  }
}

// pass 1: number of valid code points per tile, and the number of input bytes no code point consumes
// (dropped != 0  <=>  the reference would print its invalid-unicode warning, utf8.cpp:143-145)
__global__ __launch_bounds__(kBlock) void decode_count_kernel(const uint8_t *__restrict__ text, size_t nbytes,
                                                              uint32_t *__restrict__ tile_counts,
                                                              unsigned long long *__restrict__ dropped,
                                                              uint32_t *__restrict__ used) {
  __shared__ uint32_t sm[8];
  __shared__ uint32_t low_used[8];  // bitmap of code points < 256 seen by this tile
  if (threadIdx.x < 8) low_used[threadIdx.x] = 0;
  __syncthreads();
  const size_t off = static_cast<size_t>(blockIdx.x) * kDecTile + static_cast<size_t>(threadIdx.x) * kDecBytes;
  uint32_t cnt = 0, used_bytes = 0;
  if (off < nbytes) {
    uint8_t b[20];
    load_bytes20(text, nbytes, off, b);
    bool ascii = off + kDecBytes <= nbytes;
#pragma unroll
    for (int j = 0; j < kDecBytes; j++) ascii = ascii && b[j] < 0x80;
    if (ascii) {  // common case: 16 one-byte code points
      cnt = kDecBytes;
      used_bytes = kDecBytes;
      uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;  // bits of the 128-entry ASCII part of the bitmap
#pragma unroll
      for (int j = 0; j < kDecBytes; j++) {
        const uint32_t bit = 1u << (b[j] & 31);
        const int wi = b[j] >> 5;
        m0 |= wi == 0 ? bit : 0u;
        m1 |= wi == 1 ? bit : 0u;
        m2 |= wi == 2 ? bit : 0u;
        m3 |= wi == 3 ? bit : 0u;
      }
      if (m0 & ~low_used[0]) atomicOr(&low_used[0], m0);
      if (m1 & ~low_used[1]) atomicOr(&low_used[1], m1);
      if (m2 & ~low_used[2]) atomicOr(&low_used[2], m2);
      if (m3 & ~low_used[3]) atomicOr(&low_used[3], m3);
    } else {
#pragma unroll
      for (int j = 0; j < kDecBytes; j++) {
        if (off + j < nbytes) {
          if ((b[j] & 0xc0u) != 0x80u) {
            uint32_t cp = decode_one(&b[j], static_cast<int64_t>(nbytes - (off + j)));
            if (cp != kInvalidUnicode) {
              cnt++;
              used_bytes += cp < 0x80 ? 1 : cp < 0x800 ? 2 : cp < 0x10000 ? 3 : 4;
              if (cp < 256) {
                atomicOr(&low_used[cp >> 5], 1u << (cp & 31));
              } else {
                used[cp] = 1u;
              }
            }
          }
        }
      }
    }
  }
  uint32_t tot, tot_bytes;
  (void)block_excl_sum(cnt, sm, tot);
  (void)block_excl_sum(used_bytes, sm, tot_bytes);
  if (threadIdx.x == 0) {
    tile_counts[blockIdx.x] = tot;
    // bytes of this tile that no code point consumed; clean input never touches the counter (one
    // same-address atomic per tile costs more than the whole pass)
    const size_t t0 = static_cast<size_t>(blockIdx.x) * kDecTile;
    const uint32_t tile_bytes = static_cast<uint32_t>(min(static_cast<size_t>(kDecTile), nbytes - t0));
    if (tot_bytes != tile_bytes) {
      atomicAdd(dropped, static_cast<unsigned long long>(tile_bytes) - static_cast<unsigned long long>(tot_bytes));
    }
  }
  if ((low_used[threadIdx.x >> 5] >> (threadIdx.x & 31)) & 1u) used[threadIdx.x] = 1u;
}

// pass 2 (after the alphabet is known): dense symbols, class bytes, symbol histogram; the raw code
// points are only kept when a debug copy is requested
template <typename SymT>
__global__ __launch_bounds__(kBlock) void decode_write_kernel(
    const uint8_t *__restrict__ text, size_t nbytes, const uint32_t *__restrict__ tile_prefix,
    const uint32_t *__restrict__ lut_excl, SymT *__restrict__ sym, uint8_t *__restrict__ cls,
    uint32_t *__restrict__ cps_dbg, const uint32_t *__restrict__ soft, int nsoft, uint32_t *__restrict__ sym_hist,
    int hist_shift) {
  __shared__ uint32_t sm[8];
  __shared__ uint32_t scp[kDecTile];
  __shared__ uint32_t shist[256];
  __shared__ uint16_t s_ascii[128];  // (class byte << 8) | dense symbol of the ASCII code points
  // The symbol histogram only steers the code lengths (any histogram gives a valid code), so it is
  // taken from every 16th tile of large inputs: the per-tile flush is ~50 same-address atomics.
  // (wide alphabets: the histogram is over symbol >> hist_shift, the coded part of the split code)
  const bool sampled = sym_hist != nullptr && (gridDim.x < 256 || (blockIdx.x & 15) == 0);
  shist[threadIdx.x] = 0;
  auto class_of = [&](uint32_t c) {
    uint8_t f = 0;
    if (is_space(c)) f |= kClsSpace;
    if (is_punctuation(c)) f |= kClsPunct;
    if (is_spacing_char(c)) {
      f |= kClsSpacing;
      int lo = 0, hi = nsoft;  // sorted list of "soft" spacing chars (usually empty)
      while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (soft[mid] < c) lo = mid + 1; else hi = mid;
      }
      if (lo < nsoft && soft[lo] == c) f |= kClsSoft;
    }
    return f;
  };
  // (lut_excl == nullptr / sym == nullptr: no dense symbols — the fast path works on the raw code points in cps_dbg)
  if (threadIdx.x < 128) {
    const uint32_t c = threadIdx.x;
    s_ascii[c] = static_cast<uint16_t>((static_cast<uint32_t>(class_of(c)) << 8) | (lut_excl ? ((lut_excl[c] + 1u) & 0xffu) : 0u));
  }
  const size_t off = static_cast<size_t>(blockIdx.x) * kDecTile + static_cast<size_t>(threadIdx.x) * kDecBytes;
  uint32_t cp[kDecBytes];
  uint32_t cnt = 0;
  if (off < nbytes) {
    uint8_t b[20];
    load_bytes20(text, nbytes, off, b);
    bool ascii = off + kDecBytes <= nbytes;
#pragma unroll
    for (int j = 0; j < kDecBytes; j++) ascii = ascii && b[j] < 0x80;
    if (ascii) {
#pragma unroll
      for (int j = 0; j < kDecBytes; j++) cp[j] = b[j];
      cnt = kDecBytes;
    } else {
#pragma unroll
      for (int j = 0; j < kDecBytes; j++) {
        cp[j] = kInvalidUnicode;
        if (off + j < nbytes && (b[j] & 0xc0u) != 0x80u) {
          cp[j] = decode_one(&b[j], static_cast<int64_t>(nbytes - (off + j)));
          if (cp[j] != kInvalidUnicode) cnt++;
        }
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < kDecBytes; j++) cp[j] = kInvalidUnicode;
  }
  uint32_t tot;
  uint32_t pos = block_excl_sum(cnt, sm, tot);
#pragma unroll
  for (int j = 0; j < kDecBytes; j++) {
    if (cp[j] != kInvalidUnicode) scp[pos++] = cp[j];
  }
  __syncthreads();
  const size_t out_base = tile_prefix[blockIdx.x];
  for (uint32_t k = threadIdx.x; k < tot; k += kBlock) {
    const uint32_t c = scp[k];
    uint32_t sv;
    uint8_t f;
    if (c < 128 && sizeof(SymT) == 1) {
      const uint32_t e = s_ascii[c];
      sv = e & 0xffu;
      f = static_cast<uint8_t>(e >> 8);
    } else {
      sv = lut_excl ? lut_excl[c] + 1u : 0u;
      f = class_of(c);
    }
    if (sym) sym[out_base + k] = static_cast<SymT>(sv);
    if (sampled) atomicAdd(&shist[(sv >> hist_shift) & 255u], 1u);
    if (cps_dbg) cps_dbg[out_base + k] = c;
    cls[out_base + k] = f;
  }
  if (sampled) {
    __syncthreads();
    if (shist[threadIdx.x]) atomicAdd(&sym_hist[threadIdx.x], shist[threadIdx.x]);
  }
}

__global__ __launch_bounds__(kBlock) void mark_used_kernel(const uint32_t *__restrict__ cps, size_t n,
                                                           uint32_t *__restrict__ used) {
  size_t i = static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) used[cps[i]] = 1u;
  if (i == 0) used[1] = 1u;  // the separator (linear.cpp:92,99)
}

// the separator and the vocab stream behind the text: sym[n_text + k] (dense, order-preserving
// symbol id >= 1; 0 is reserved for "past the end").  (The vocab stream is left out of the symbol
// histogram: it only steers the code lengths, and a few 1e5 same-address atomics are slow.)
template <typename SymT>
__global__ __launch_bounds__(kBlock) void map_vocab_symbols_kernel(const uint32_t *__restrict__ vocab_cps,
                                                                   size_t n_text, size_t n,
                                                                   const uint32_t *__restrict__ lut_excl,
                                                                   SymT *__restrict__ sym) {
  size_t i = n_text + static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= n) return;
  const uint32_t c = i == n_text ? 1u : vocab_cps[i - n_text - 1];
  const uint32_t sv = lut_excl[c] + 1u;
  sym[i] = static_cast<SymT>(sv);
}

// Device copy of the symbol code (code.h).  uniform_bits > 0: fixed width, tables unused.
struct DevCode {
  const uint16_t *cw;
  const uint8_t *len;
  const uint8_t *first_len;  // the kDecodeTableBytes of count_key_symbols' table (bmask u16[4096])
  int uniform_bits;  // > 0: fixed width; 0: variable-length code; < 0: split code with lo_bits = -uniform_bits
};
constexpr int kDecodeTableBytes = 2 * 4096;

// number of complete codewords inside the first t bits of a kKeyBits-bit key (t <= kKeyBits).
// tab = bmask u16[4096] (code.h): the codeword ends inside a 12-bit window as a bit mask, so one
// table step counts and skips all whole codewords of the window that still fit.
__device__ __forceinline__ int count_key_symbols(uint64_t key, int t, const uint8_t *tab, int uniform_bits) {
  if (uniform_bits > 0) return t / uniform_bits;
  const uint16_t *bmask = reinterpret_cast<const uint16_t *>(tab);
  int pos = 0, cnt = 0;
  if (uniform_bits < 0) {  // split code: codeword of the high part, then -uniform_bits verbatim bits
    const int lo = -uniform_bits;
    while (pos < t) {
      const int sh = kKeyBits - pos - 12;
      const uint32_t w = static_cast<uint32_t>(sh >= 0 ? (key >> sh) : (key << -sh)) & 0xfffu;
      const int l = __ffs(static_cast<int>(bmask[w])) + lo;  // first codeword of the window + low bits
      if (pos + l > t) break;
      pos += l;
      cnt++;
    }
    return cnt;
  }
  while (pos < t) {
    const int sh = kKeyBits - pos - 12;
    const uint32_t w = static_cast<uint32_t>(sh >= 0 ? (key >> sh) : (key << -sh)) & 0xfffu;
    const int r = t - pos;
    const uint32_t bm = bmask[w] & (r >= 12 ? 0xfffu : ((1u << r) - 1u));
    if (!bm) break;
    cnt += __popc(bm);
    pos += 32 - __clz(static_cast<int>(bm));
  }
  return cnt;
}

// First index in [lo, hi] whose key is >= key (hi: a position known to qualify, or the end of the array).  The
// searches of this path are latency chains, not bandwidth: while the range is wide the whole wave probes 64 evenly
// spaced positions with ONE load instruction and a ballot picks the stretch that holds the answer (3 steps from
// 1e8 keys down to a few hundred, and the probes of the first two steps are the same cache lines for every search:
// L2 hits); the last few hundred keys are finished by a binary search (a 64-way step there would pull in 64 lines
// of its own per search where the binary search touches 3).  12 dependent loads instead of 27.
// All 64 lanes must be active; every lane returns the same index.
#ifndef WP_WAVE_SEARCH_NARROW
#define WP_WAVE_SEARCH_NARROW 512
#endif
constexpr size_t kWaveSearchNarrow = WP_WAVE_SEARCH_NARROW;
__device__ __forceinline__ size_t wave_key_lower_bound(const Key0 *__restrict__ keys, size_t lo, size_t hi, uint64_t key) {
  const size_t lane = static_cast<size_t>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  while (hi - lo > kWaveSearchNarrow) {
    const size_t st = (hi - lo) / kWave + 1;
    const size_t idx = lo + lane * st;
    const bool ge = idx < hi ? static_cast<uint64_t>(keys[idx]) >= key : true;
    const uint64_t m = __ballot(ge);
    if (!m) {  // (all 64 probes in range and below the key)
      lo += (kWave - 1) * st + 1;
      continue;
    }
    const size_t t = static_cast<size_t>(__ffsll(static_cast<long long>(m)) - 1);
    const size_t first_ge = lo + t * st;
    if (t) lo += (t - 1) * st + 1;
    hi = first_ge < hi ? first_ge : hi;
    if (!t) hi = lo;
  }
  while (lo < hi) {  // (uniform: the loads broadcast)
    const size_t md = (lo + hi) >> 1;
    if (static_cast<uint64_t>(keys[md]) < key) lo = md + 1; else hi = md;
  }
  return lo;
}

// Round-0 keys: the first 63 bits of the codeword stream of every suffix (most significant bit
// first).  Positions past the end read symbol 0, whose codeword is the smallest, so a shorter
// suffix sorts first.
//
// stream(i) = codeword(i) ++ stream(i+1), hence key(i) = cw_i << (63 - l_i) | key(i+1) >> l_i: a lane
// owns 8 consecutive positions, builds the key of its last one symbol by symbol (≈14 table steps)
// and rolls the other 7 out of it (one step each).  LDS arrays are indexed p + (p >> 3) so that lanes
// striding by 8 hit distinct banks; keys leave through LDS as full coalesced rows.
constexpr int kKeyItems = 8;
constexpr int kKeyTile = kBlock * kKeyItems;
constexpr int kKeyHalo = 64;
__device__ __forceinline__ int key_pad(int p) { return p + (p >> 3); }
template <typename SymT>
__global__ __launch_bounds__(kBlock) void build_keys0_kernel(const SymT *__restrict__ sym, size_t n, DevCode code,
                                                             Key0 *__restrict__ keys,
                                                             uint8_t *__restrict__ dig0) {
  constexpr int kSymSlots = kKeyTile + kKeyHalo;
  __shared__ uint32_t ss[kSymSlots + kSymSlots / 8 + 1];
  __shared__ uint32_t stab[256];  // (len << 16) | codeword
  __shared__ Key0 skey[kKeyTile + kKeyTile / 8 + 1];  // (keys of up to 32 bits: half the LDS, 8 instead of 5 workgroups per CU)
  const size_t base = static_cast<size_t>(blockIdx.x) * kKeyTile;
  for (int k = threadIdx.x; k < kSymSlots; k += kBlock) {
    size_t i = base + k;
    ss[key_pad(k)] = i < n ? static_cast<uint32_t>(sym[i]) : 0u;
  }
  const int ub = code.uniform_bits > 0 ? code.uniform_bits : 0;
  const int lo = code.uniform_bits < 0 ? -code.uniform_bits : 0;  // split code: verbatim low bits
  const uint32_t lomask = (1u << lo) - 1u;
  if (!ub) stab[threadIdx.x] = (static_cast<uint32_t>(code.len[threadIdx.x]) << 16) | code.cw[threadIdx.x];
  __syncthreads();
  const int p0 = threadIdx.x * kKeyItems;
  if (base + p0 < n) {
    uint64_t key = 0;
    int used = 0, q = p0 + kKeyItems - 1;
    while (used < kKeyBits) {  // key of the lane's last position
      const uint32_t sv = ss[key_pad(q)];
      q++;
      const uint32_t e = ub ? 0u : stab[sv >> lo];
      const int l = ub ? ub : static_cast<int>(e >> 16) + lo;
      const uint32_t c = ub ? sv : (((e & 0xffffu) << lo) | (sv & lomask));
      const int take = min(l, kKeyBits - used);
      key = (key << take) | (c >> (l - take));
      used += take;
    }
    skey[key_pad(p0 + kKeyItems - 1)] = static_cast<Key0>(key);
#pragma unroll
    for (int j = kKeyItems - 2; j >= 0; j--) {
      const uint32_t sv = ss[key_pad(p0 + j)];
      const uint32_t e = ub ? 0u : stab[sv >> lo];
      const int l = ub ? ub : static_cast<int>(e >> 16) + lo;
      const uint64_t c = ub ? sv : (((e & 0xffffu) << lo) | (sv & lomask));
      key = (c << (kKeyBits - l)) | (key >> l);
      skey[key_pad(p0 + j)] = static_cast<Key0>(key);
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kKeyItems; j++) {
    const int li = j * kBlock + threadIdx.x;
    const size_t i = base + li;
    if (i < n) {
      const Key0 k = skey[key_pad(li)];
      keys[i] = k;  // (the values, 0..n-1, are made up by the first radix pass)
      if (dig0) dig0[i] = static_cast<uint8_t>(k);  // first radix digit: that pass's histogram reads 1 byte per key
    }
  }
}

}  // namespace wp
