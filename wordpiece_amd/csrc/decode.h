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
#include "radix_sort.h"
#include "utf8_swar.h"

namespace wp {

// ---- layout of the two UTF-8 passes -------------------------------------------------------------------------
// A wave owns kDecRows rows of 1 KB: lane l of row r holds the 16 bytes at row base + 16 l, so every row is one
// fully coalesced 16-byte-per-lane load and all rows of a wave are in flight before anything is used.  A
// workgroup (4 waves) covers 16 KB of input with ONE block-wide prefix step.  Bytes are tested as 32-bit words
// (SWAR): a word of pure ASCII is one mask test, and the structure of multi-byte sequences — lead classes,
// continuation bytes behind them, overlong forms, surrogates, > U+10FFFF — is evaluated for the four byte positions
// of a word at once from the word and its three byte-shifted successors (v_alignbyte), no per-byte unpacking.
constexpr int kDecChunk = 16;                               // bytes per lane and row
constexpr int kDecRows = 4;                                 // rows per wave
constexpr int kDecRowBytes = kWave * kDecChunk;             // 1 KB
constexpr int kDecWaveBytes = kDecRowBytes * kDecRows;      // 4 KB
constexpr int kDecTile = (kBlock / kWave) * kDecWaveBytes;  // 16 KB of input per workgroup
constexpr uint32_t kCpTableSize = 0x110000;                 // lut[] covers every code point
constexpr uint32_t kCpWords = kCpTableSize / 32;            // the used-code-point bitmap

typedef uint32_t dec_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
// The rows of a wave: w[r][0..3] = the lane's 16 bytes of row r, w[r][4] = the four bytes behind them.  Bytes at or
// behind nbytes read as zero (never a continuation byte: a sequence cut off by the end of the text is invalid, as
// decode_one's size test has it).  text: 4-byte aligned, readable up to the next multiple of 16 behind nbytes.
__device__ __forceinline__ void dec_load_rows(const uint8_t *__restrict__ text, size_t nbytes, size_t wave_base, int lane,
                                              uint32_t (&w)[kDecRows][5]) {
  const size_t padded = (nbytes + 15) & ~static_cast<size_t>(15);
#pragma unroll
  for (int r = 0; r < kDecRows; r++) {
    const size_t off = wave_base + static_cast<size_t>(r) * kDecRowBytes + static_cast<size_t>(lane) * kDecChunk;
    if (off < padded) {  // (one 16-byte load: global memory wants dword alignment for it, which the text has)
      const dec_u32x4 v = *reinterpret_cast<const dec_u32x4 *>(text + off);
      w[r][0] = v.x;
      w[r][1] = v.y;
      w[r][2] = v.z;
      w[r][3] = v.w;
    } else {
      w[r][0] = w[r][1] = w[r][2] = w[r][3] = 0u;
    }
  }
  const size_t tail_off = wave_base + kDecWaveBytes;  // the word behind the wave's last row (one broadcast load)
  const uint32_t tail = tail_off < padded ? *reinterpret_cast<const uint32_t *>(text + tail_off) : 0u;
#pragma unroll
  for (int r = 0; r < kDecRows; r++) {
    const uint32_t from_next_lane = __shfl_down(w[r][0], 1, kWave);
    const uint32_t next_row = r + 1 < kDecRows ? __shfl(w[r + 1 < kDecRows ? r + 1 : r][0], 0, kWave) : tail;
    w[r][4] = lane == kWave - 1 ? next_row : from_next_lane;
  }
  if (wave_base + kDecWaveBytes + 4 > nbytes) {  // (wave-uniform: only the wave that holds the end of the text)
#pragma unroll
    for (int r = 0; r < kDecRows; r++) {
      const size_t off = wave_base + static_cast<size_t>(r) * kDecRowBytes + static_cast<size_t>(lane) * kDecChunk;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const size_t pos = off + 4 * static_cast<size_t>(k);
        if (pos >= nbytes) {
          w[r][k] = 0u;
        } else if (pos + 4 > nbytes) {
          w[r][k] &= (1u << (8 * static_cast<unsigned>(nbytes - pos))) - 1u;
        }
      }
    }
  }
}
// positions of the lane's chunk that lie inside the text, as a 16-bit mask (0xffff except at the very end)
__device__ __forceinline__ uint32_t dec_inside16(size_t off, size_t nbytes) {
  if (off + kDecChunk <= nbytes) return 0xffffu;
  if (off >= nbytes) return 0u;
  return (1u << static_cast<unsigned>(nbytes - off)) - 1u;
}

// pass 1: number of valid code points per tile, the number of input bytes no code point consumes
// (dropped != 0  <=>  the reference would print its invalid-unicode warning, utf8.cpp:143-145), and — MARK — the
// set of code points that occur, as a bitmap (used_bits, kCpWords words): ASCII through 128 flag words in LDS
// (a plain store per byte), the rest of the BMP through an 8 KB LDS bitmap that the tile merges into the global
// one (read first: once a code point is known, a tile only reads), astral code points directly.
template <bool MARK>
__global__ __launch_bounds__(kBlock) void decode_count_kernel(const uint8_t *__restrict__ text, size_t nbytes,
                                                              uint32_t *__restrict__ tile_counts,
                                                              unsigned long long *__restrict__ dropped,
                                                              uint32_t *__restrict__ used_bits) {
  __shared__ uint32_t s_seen[128];   // ASCII code points seen by this tile (flag words: same-address stores merge)
  __shared__ uint32_t s_bmp[2048];   // code points 0x80..0xffff seen by this tile
  __shared__ uint32_t s_cnt[kBlock / kWave], s_use[kBlock / kWave], s_multi;
  const int lane = lane_id(), wv = wave_id();
  if (MARK) {
    if (threadIdx.x < 128) s_seen[threadIdx.x] = 0u;
#pragma unroll
    for (int q = 0; q < 2048 / kBlock; q++) s_bmp[q * kBlock + threadIdx.x] = 0u;
    if (threadIdx.x == 0) s_multi = 0u;
    __syncthreads();
  }
  const size_t tile_base = static_cast<size_t>(blockIdx.x) * kDecTile;
  const size_t wave_base = tile_base + static_cast<size_t>(wv) * kDecWaveBytes;
  uint32_t w[kDecRows][5];
  dec_load_rows(text, nbytes, wave_base, lane, w);
  uint32_t cnt = 0, used_bytes = 0;
  bool multi = false;
#pragma unroll
  for (int r = 0; r < kDecRows; r++) {
    const size_t off = wave_base + static_cast<size_t>(r) * kDecRowBytes + static_cast<size_t>(lane) * kDecChunk;
    const uint32_t inside = dec_inside16(off, nbytes);
    if (inside == 0u) continue;
    const bool ascii = ((w[r][0] | w[r][1] | w[r][2] | w[r][3]) & kHi) == 0u && inside == 0xffffu;
    if (ascii) {  // common case: 16 one-byte code points
      cnt += kDecChunk;
      used_bytes += kDecChunk;
      if (MARK) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
          for (int j = 0; j < 4; j++) s_seen[(w[r][k] >> (8 * j)) & 0xffu] = 1u;
        }
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t in4 = (inside >> (4 * k)) & 0xfu;
      if (in4 == 0u) continue;
      const uint32_t in_bytes = ((in4 & 1u) << 7) | ((in4 & 2u) << 14) | ((in4 & 4u) << 21) | ((in4 & 8u) << 28);
      Utf8Starts u = utf8_starts(w[r][k], w[r][k + 1]);
      u.v1 &= in_bytes;
      u.v2 &= in_bytes;
      u.v3 &= in_bytes;
      u.v4 &= in_bytes;
      const uint32_t n1 = __popc(u.v1), n2 = __popc(u.v2), n3 = __popc(u.v3), n4 = __popc(u.v4);
      cnt += n1 + n2 + n3 + n4;
      used_bytes += n1 + 2 * n2 + 3 * n3 + 4 * n4;
      if (MARK) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
          if ((u.v1 >> (8 * j + 7)) & 1u) s_seen[(w[r][k] >> (8 * j)) & 0xffu] = 1u;
        }
        uint32_t m = u.v2 | u.v3 | u.v4;
        while (m) {
          const int bit = __ffs(static_cast<int>(m)) - 1;  // 8 j + 7
          m &= m - 1u;
          const uint32_t cp = utf8_value(__builtin_amdgcn_alignbyte(w[r][k + 1], w[r][k], static_cast<uint32_t>(bit >> 3)));
          const uint32_t b = 1u << (cp & 31u);
          if (cp < 0x10000u) {
            if (!(s_bmp[cp >> 5] & b)) atomicOr(&s_bmp[cp >> 5], b);
            multi = true;
          } else if (!(used_bits[cp >> 5] & b)) {
            atomicOr(&used_bits[cp >> 5], b);
          }
        }
      }
    }
  }
  cnt = wave_reduce_sum(cnt);
  used_bytes = wave_reduce_sum(used_bytes);
  if (lane == 0) {
    s_cnt[wv] = cnt;
    s_use[wv] = used_bytes;
  }
  if (MARK && multi) s_multi = 1u;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0, tot_bytes = 0;
#pragma unroll
    for (int i = 0; i < kBlock / kWave; i++) {
      tot += s_cnt[i];
      tot_bytes += s_use[i];
    }
    tile_counts[blockIdx.x] = tot;
    // bytes of this tile that no code point consumed; clean input never touches the counter
    const uint32_t tile_bytes = static_cast<uint32_t>(min(static_cast<size_t>(kDecTile), nbytes - tile_base));
    if (tot_bytes != tile_bytes) {
      atomicAdd(dropped, static_cast<unsigned long long>(tile_bytes) - static_cast<unsigned long long>(tot_bytes));
    }
  }
  if (MARK) {
    if (threadIdx.x < 128) {  // waves 0 and 1: the ASCII flags as two bitmap words each
      const uint64_t m = __ballot(s_seen[threadIdx.x] != 0u);
      if (lane < 2) {
        const uint32_t bits = static_cast<uint32_t>(m >> (32 * lane));
        uint32_t *dst = used_bits + 2 * wv + lane;
        if (bits && (*dst & bits) != bits) atomicOr(dst, bits);
      }
    }
    if (s_multi) {
#pragma unroll
      for (int q = 0; q < 2048 / kBlock; q++) {
        const int i = q * kBlock + threadIdx.x;
        const uint32_t bits = s_bmp[i];
        if (bits && (used_bits[i] & bits) != bits) atomicOr(&used_bits[i], bits);
      }
    }
  }
}

// the code points of the vocabulary and the separator (linear.cpp:92,99) join the alphabet: the vocabulary's words
// of the bitmap come precomputed from the host (vocab.h: one entry per bitmap word, so no two threads meet)
__global__ __launch_bounds__(kBlock) void vocab_alphabet_kernel(const uint32_t *__restrict__ word_idx,
                                                                const uint32_t *__restrict__ word_bits, uint32_t n,
                                                                uint32_t *__restrict__ used_bits) {
  const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) used_bits[word_idx[i]] |= word_bits[i];
}

// Alphabet compaction: lut[c] = number of used code points below c (the dense symbol of a used c is lut[c] + 1).
// One workgroup turns the bitmap into per-word prefixes (and sigma), a second launch expands them.
constexpr int kAlphaThreads = 1024;
constexpr int kAlphaWords = kCpWords / kAlphaThreads;  // 34 bitmap words per thread
static_assert(kCpWords % kAlphaThreads == 0, "the bitmap divides evenly");
__global__ __launch_bounds__(kAlphaThreads) void alphabet_prefix_kernel(const uint32_t *__restrict__ used_bits,
                                                                       uint32_t *__restrict__ word_prefix,
                                                                       uint32_t *__restrict__ sigma) {
  __shared__ uint32_t s_w[kAlphaThreads / kWave];
  const uint32_t first = threadIdx.x * kAlphaWords;
  uint32_t bits[kAlphaWords], sum = 0;
#pragma unroll
  for (int i = 0; i < kAlphaWords; i++) {
    bits[i] = used_bits[first + i];
    sum += __popc(bits[i]);
  }
  const uint32_t inc = wave_incl_sum(sum);
  if (lane_id() == kWave - 1) s_w[wave_id()] = inc;
  __syncthreads();
  uint32_t run = inc - sum, all = 0;
  for (int i = 0; i < kAlphaThreads / kWave; i++) {
    const uint32_t v = s_w[i];
    if (i < wave_id()) run += v;
    all += v;
  }
#pragma unroll
  for (int i = 0; i < kAlphaWords; i++) {
    word_prefix[first + i] = run;
    run += __popc(bits[i]);
  }
  if (threadIdx.x == 0) *sigma = all;
}
__global__ __launch_bounds__(kBlock) void alphabet_lut_kernel(const uint32_t *__restrict__ used_bits,
                                                              const uint32_t *__restrict__ word_prefix,
                                                              uint32_t *__restrict__ lut_excl) {
  const uint32_t cp = blockIdx.x * kBlock + threadIdx.x;  // (grid = kCpTableSize / kBlock exactly)
  lut_excl[cp] = word_prefix[cp >> 5] + __popc(used_bits[cp >> 5] & ((1u << (cp & 31u)) - 1u));
}

// class byte of a code point (utf8.cpp:10-29 + the handle's list of "soft" spacing chars, usually empty): the BMP
// through the handle's table (vocab.h, cls_bmp), the rest computed
__device__ __forceinline__ uint8_t class_of_cp(uint32_t c, const uint8_t *__restrict__ cls_bmp, const uint32_t *__restrict__ soft,
                                               int nsoft) {
  if (c < 0x10000u) return cls_bmp[c];
  uint8_t f = 0;
  if (is_space(c)) f |= kClsSpace;
  if (is_punctuation(c)) f |= kClsPunct;
  if (is_spacing_char(c)) {
    f |= kClsSpacing;
    int lo = 0, hi = nsoft;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (soft[mid] < c) lo = mid + 1; else hi = mid;
    }
    if (lo < nsoft && soft[lo] == c) f |= kClsSoft;
  }
  return f;
}

// Writes 16 consecutive bytes that sit in the wave's staging row at [16 lane, 16 lane + 16) to global memory at
// dst + 16 lane, dst of ANY alignment: the staging row is re-read shifted by the (wave-uniform) misalignment so that
// every lane stores aligned dwords; the a = dst & 3 bytes in front of the first aligned dword and behind the last one
// are stored as bytes (they share their dword with the neighbouring row, which belongs to another wave).
// stage: the wave's row, kDecRowBytes bytes + 16 of slack in front (stage[-4..-1] readable); count = bytes of the
// row that are valid (a multiple of 16, or the ragged last row of the text)
__device__ __forceinline__ void dec_flush_row_bytes(const uint32_t *stage32, uint8_t *__restrict__ dst, uint32_t count, int lane) {
  const uint32_t a = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(dst)) & 3u;
  // aligned dword m of the row covers staged bytes [4 m - a, 4 m - a + 4)
  uint32_t *dst32 = reinterpret_cast<uint32_t *>(dst - a);
  const uint32_t ndw = (count + a) >> 2;  // whole dwords [1, ndw) when a != 0, [0, ndw) when a == 0; ragged bytes behind
  const uint32_t m0 = 4u * static_cast<uint32_t>(lane);
  uint32_t s[5];
#pragma unroll
  for (int q = 0; q < 5; q++) s[q] = stage32[static_cast<int>(m0) + q - 1];  // dwords m0 - 1 .. m0 + 3 of the staging row
  uint32_t o[4];
#pragma unroll
  for (int q = 0; q < 4; q++) o[q] = a ? __builtin_amdgcn_alignbyte(s[q + 1], s[q], 4u - a) : s[q + 1];
  const uint32_t first_whole = a ? 1u : 0u;
  if (m0 >= first_whole && m0 + 4u <= ndw) {
    *reinterpret_cast<uint4 *>(dst32 + m0) = make_uint4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t m = m0 + q;
      if (m >= first_whole && m < ndw) {
        dst32[m] = o[q];
      } else {  // a partial dword at either end of the row: byte stores of the staged bytes inside [0, count)
#pragma unroll
        for (int b = 0; b < 4; b++) {
          const int sb = static_cast<int>(4u * m + b) - static_cast<int>(a);
          if (sb >= 0 && static_cast<uint32_t>(sb) < count) dst[sb] = static_cast<uint8_t>(o[q] >> (8 * b));
        }
      }
    }
  }
  // the last a bytes of a full row spill into dword m = count / 4 (+ a / 4), which no lane owns
  if (a && lane == kWave - 1 && count == static_cast<uint32_t>(kDecRowBytes)) {
    const uint32_t last = stage32[kDecRowBytes / 4 - 1];
    for (uint32_t b = 4u - a; b < 4u; b++) dst[kDecRowBytes - 4 + b] = static_cast<uint8_t>(last >> (8 * b));
  }
}

// pass 2 (after the alphabet is known): dense symbols, class bytes, symbol histogram; the raw code
// points are only kept when a debug copy is requested (or for the fast path, which works on them).
// Per wave and row: a row of pure ASCII (the common case even in text with the odd multi-byte character) maps input
// byte p to output element p — symbols and classes come from a 128-entry LDS table, are staged in the wave's LDS
// row and leave as aligned 16-byte stores whatever the alignment of the output position; any other row compacts
// the positions of its valid leads into the wave's LDS list and is then handled one output element per lane.
template <typename SymT>
__global__ __launch_bounds__(kBlock) void decode_write_kernel(
    const uint8_t *__restrict__ text, size_t nbytes, const uint32_t *__restrict__ tile_prefix,
    const uint32_t *__restrict__ lut_excl, SymT *__restrict__ sym, uint8_t *__restrict__ cls,
    uint32_t *__restrict__ cps_dbg, const uint8_t *__restrict__ cls_bmp, const uint32_t *__restrict__ soft, int nsoft,
    uint32_t *__restrict__ sym_hist, int hist_shift) {
  constexpr int WAVES = kBlock / kWave;
  constexpr int kStageWords = kDecRowBytes / 4 + 8;  // 16 bytes of slack in front of a row, the four bytes behind it
  __shared__ uint32_t s_stage[WAVES][2][kStageWords];  // per wave: [0] symbols (ASCII row) / raw bytes (mixed row), [1] classes
  __shared__ uint16_t s_list[WAVES][kDecRowBytes];     // positions of the valid leads of a mixed row
  __shared__ uint32_t s_rows[WAVES][kDecRows];
  __shared__ uint32_t shist[256];
  __shared__ uint16_t s_ascii[128];  // (class byte << 8) | dense symbol of the ASCII code points
  // The symbol histogram only steers the code lengths (any histogram gives a valid code), so it is
  // taken from every 16th tile of large inputs.
  // (wide alphabets: the histogram is over symbol >> hist_shift, the coded part of the split code)
  const bool sampled = sym_hist != nullptr && (gridDim.x < 64 || (blockIdx.x & 15) == 0);
  const int lane = lane_id(), wv = wave_id();
  shist[threadIdx.x] = 0;
  // (lut_excl == nullptr / sym == nullptr: no dense symbols — the fast path works on the raw code points in cps_dbg)
  if (threadIdx.x < 128) {
    const uint32_t c = threadIdx.x;
    s_ascii[c] = static_cast<uint16_t>((static_cast<uint32_t>(cls_bmp[c]) << 8) |
                                       (lut_excl && sizeof(SymT) == 1 ? ((lut_excl[c] + 1u) & 0xffu) : 0u));
  }
  if (lane < 4) {
    s_stage[wv][0][lane] = 0u;  // the slack in front of the rows (read by the shifted flush, never stored)
    s_stage[wv][1][lane] = 0u;
  }
  const size_t tile_base = static_cast<size_t>(blockIdx.x) * kDecTile;
  const size_t wave_base = tile_base + static_cast<size_t>(wv) * kDecWaveBytes;
  uint32_t w[kDecRows][5];
  dec_load_rows(text, nbytes, wave_base, lane, w);
  // valid starts per lane and row (16-bit masks), row totals
  uint32_t starts[kDecRows], row_cnt[kDecRows];
  uint64_t row_ascii = 0;  // bit r: every lane's chunk of row r is pure ASCII and inside the text
#pragma unroll
  for (int r = 0; r < kDecRows; r++) {
    const size_t off = wave_base + static_cast<size_t>(r) * kDecRowBytes + static_cast<size_t>(lane) * kDecChunk;
    const uint32_t inside = dec_inside16(off, nbytes);
    const bool ascii = ((w[r][0] | w[r][1] | w[r][2] | w[r][3]) & kHi) == 0u;
    uint32_t m = 0;
    if (ascii) {
      m = inside;
    } else {
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const Utf8Starts u = utf8_starts(w[r][k], w[r][k + 1]);
        m |= byte_mask4(u.v1 | u.v2 | u.v3 | u.v4) << (4 * k);
      }
      m &= inside;
    }
    starts[r] = m;
    if (__ballot(ascii && inside == 0xffffu) == ~0ull) row_ascii |= 1ull << r;
    row_cnt[r] = wave_reduce_sum(__popc(m));
  }
  if (lane == 0) {
#pragma unroll
    for (int r = 0; r < kDecRows; r++) s_rows[wv][r] = row_cnt[r];
  }
  __syncthreads();
  size_t out = tile_prefix[blockIdx.x];  // output position of the wave's current row
  for (int i = 0; i < wv; i++) {
#pragma unroll
    for (int r = 0; r < kDecRows; r++) out += s_rows[i][r];
  }
  uint32_t *stage_a = &s_stage[wv][0][4], *stage_b = &s_stage[wv][1][4];  // (16 bytes of slack in front)
#pragma unroll
  for (int r = 0; r < kDecRows; r++) {
    if (row_cnt[r] == 0) continue;  // (wave-uniform)
    if ((row_ascii >> r) & 1ull) {
      // ---- a row of pure ASCII: output element = input byte
      uint32_t sy[4], cl[4];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sy[k] = cl[k] = 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const uint32_t e = s_ascii[(w[r][k] >> (8 * j)) & 0x7fu];
          sy[k] |= (e & 0xffu) << (8 * j);
          cl[k] |= (e >> 8) << (8 * j);
        }
      }
      if (sampled && sizeof(SymT) == 1) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
#pragma unroll
          for (int j = 0; j < 4; j++) atomicAdd(&shist[((sy[k] >> (8 * j)) & 0xffu) >> hist_shift], 1u);
        }
      }
      const size_t e0 = out + static_cast<size_t>(lane) * kDecChunk;
      if (sizeof(SymT) == 1 && sym) {
        reinterpret_cast<uint4 *>(stage_a)[lane] = make_uint4(sy[0], sy[1], sy[2], sy[3]);
      }
      reinterpret_cast<uint4 *>(stage_b)[lane] = make_uint4(cl[0], cl[1], cl[2], cl[3]);
      __builtin_amdgcn_wave_barrier();
      if (sizeof(SymT) == 1 && sym) dec_flush_row_bytes(stage_a, reinterpret_cast<uint8_t *>(sym) + out, kDecRowBytes, lane);
      dec_flush_row_bytes(stage_b, cls + out, kDecRowBytes, lane);
      __builtin_amdgcn_wave_barrier();
      if (sizeof(SymT) == 4 && sym) {  // 4-byte symbols: 16 consecutive elements per lane, always dword aligned
#pragma unroll
        for (int k = 0; k < 4; k++) {
          uint32_t v[4];
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const uint32_t c = (w[r][k] >> (8 * j)) & 0x7fu;
            v[j] = lut_excl[c] + 1u;
            if (sampled) atomicAdd(&shist[(v[j] >> hist_shift) & 255u], 1u);
          }
          *reinterpret_cast<uint4 *>(reinterpret_cast<uint32_t *>(sym) + e0 + 4 * k) = make_uint4(v[0], v[1], v[2], v[3]);
        }
      }
      if (cps_dbg) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
          *reinterpret_cast<uint4 *>(cps_dbg + e0 + 4 * k) =
              make_uint4(w[r][k] & 0xffu, (w[r][k] >> 8) & 0xffu, (w[r][k] >> 16) & 0xffu, w[r][k] >> 24);
        }
      }
    } else {
      // ---- a mixed row: positions of the valid leads -> the wave's list, raw bytes -> the wave's staging row
      const uint32_t c = __popc(starts[r]);
      uint32_t o = wave_incl_sum(c) - c;
      reinterpret_cast<uint4 *>(stage_a)[lane] = make_uint4(w[r][0], w[r][1], w[r][2], w[r][3]);
      if (lane == kWave - 1) stage_a[kDecRowBytes / 4] = w[r][4];  // the four bytes behind the row
      uint32_t m = starts[r];
      while (m) {
        const int j = __ffs(static_cast<int>(m)) - 1;
        m &= m - 1u;
        s_list[wv][o++] = static_cast<uint16_t>(lane * kDecChunk + j);
      }
      __builtin_amdgcn_wave_barrier();
      for (uint32_t e = lane; e < row_cnt[r]; e += kWave) {
        const uint32_t pos = s_list[wv][e];
        const uint32_t x = __builtin_amdgcn_alignbyte(stage_a[(pos >> 2) + 1], stage_a[pos >> 2], pos & 3u);
        const uint32_t cp = utf8_value(x);
        uint32_t sv;
        uint8_t f;
        if (cp < 128u && sizeof(SymT) == 1) {
          const uint32_t en = s_ascii[cp];
          sv = en & 0xffu;
          f = static_cast<uint8_t>(en >> 8);
        } else {
          sv = lut_excl ? lut_excl[cp] + 1u : 0u;
          f = cp < 128u ? static_cast<uint8_t>(s_ascii[cp] >> 8) : class_of_cp(cp, cls_bmp, soft, nsoft);
        }
        if (sym) sym[out + e] = static_cast<SymT>(sv);
        if (sampled) atomicAdd(&shist[(sv >> hist_shift) & 255u], 1u);
        if (cps_dbg) cps_dbg[out + e] = cp;
        cls[out + e] = f;
      }
      __builtin_amdgcn_wave_barrier();
    }
    out += row_cnt[r];
  }
  if (sampled) {
    __syncthreads();
    if (shist[threadIdx.x]) atomicAdd(&sym_hist[threadIdx.x], shist[threadIdx.x]);
  }
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

// First index in [lo, hi] whose key is >= key (hi: a position known to qualify, or the end of the array), by a whole
// wave.  What a search costs here is the number of distinct 128-byte lines it pulls in (gather probe,
// profiles/yardstick: ~1 ns of the CU's L1 fill per line from the L2, ~5 ns from HBM), not the length of its chain:
// a step probes WAYS evenly spaced positions with one load and a ballot picks the stretch that holds the answer.
// wide_steps 64-way steps first — worth their 64 lines only while all searches of a kernel probe the SAME lines (the
// first two steps over the whole array: L2 hits) — then 8-way steps (7 lines each, 3 bits), the last 8 keys by a
// binary search inside one or two lines.  All 64 lanes must be active; every lane returns the same index.
template <int WAYS>
__device__ __forceinline__ void wave_search_step(const Key0 *__restrict__ keys, size_t &lo, size_t &hi, uint64_t key, size_t lane) {
  const size_t st = (hi - lo) / WAYS + 1;
  const size_t idx = lo + lane * st;
  const bool ge = (lane < static_cast<size_t>(WAYS) && idx < hi) ? static_cast<uint64_t>(keys[idx]) >= key : true;
  const uint64_t m = __ballot(ge);
  if (!m) {  // (WAYS == 64: all probes in range and below the key)
    lo += (WAYS - 1) * st + 1;
    return;
  }
  const size_t t = static_cast<size_t>(__ffsll(static_cast<long long>(m)) - 1);
  const size_t first_ge = lo + t * st;
  if (t) lo += (t - 1) * st + 1;
  hi = first_ge < hi ? first_ge : hi;
  if (!t) hi = lo;
}
__device__ __forceinline__ size_t wave_key_lower_bound(const Key0 *__restrict__ keys, size_t lo, size_t hi, uint64_t key,
                                                       int wide_steps) {
  const size_t lane = static_cast<size_t>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  for (int s = 0; s < wide_steps && hi - lo > 512; s++) wave_search_step<kWave>(keys, lo, hi, key, lane);
  while (hi - lo > 8) wave_search_step<8>(keys, lo, hi, key, lane);
  while (lo < hi) {  // (uniform: the loads broadcast)
    const size_t md = (lo + hi) >> 1;
    if (static_cast<uint64_t>(keys[md]) < key) lo = md + 1; else hi = md;
  }
  return lo;
}
// First index in [from, n] whose key is >= key when the answer is expected near `from` (the end of an equal range that
// starts there): lane j probes from + 2^j - 1 (16 lanes, then the other 16: a range of up to 32 K keys costs ~11
// lines), the stretch between two probes is searched as above.
__device__ __forceinline__ size_t wave_key_gallop(const Key0 *__restrict__ keys, size_t from, size_t n, uint64_t key) {
  const size_t lane = static_cast<size_t>(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)));
  for (int phase = 0; phase < 2; phase++) {
    const size_t j = lane + 16u * static_cast<size_t>(phase);  // probe exponent of this lane (lanes >= 16: none)
    const size_t idx = from + ((size_t(1) << (j & 31)) - 1);
    const bool ge = (lane < 16 && idx < n) ? static_cast<uint64_t>(keys[idx]) >= key : true;
    const uint64_t m = __ballot(ge);
    const size_t t = static_cast<size_t>(__ffsll(static_cast<long long>(m)) - 1);  // (lanes >= 16 answer true: t <= 16)
    if (t == 16 && phase == 0) continue;  // not within 2^15 keys: the far probes
    const size_t e = t + 16u * static_cast<size_t>(phase);  // first exponent whose probe is >= key (or beyond n)
    if (e == 0) return from;
    const size_t lo = min(from + (size_t(1) << (e - 1)), n);  // (= the probe before it, + 1)
    const size_t hi = e >= 32 ? n : max(lo, min(from + ((size_t(1) << e) - 1), n));
    return wave_key_lower_bound(keys, lo, hi, key, 0);
  }
  return n;  // (not reached: the second phase always returns)
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

// The same keys for 8-bit symbols, out of registers: a lane owns 16 consecutive positions (one 16-byte load) and
// reads the 16 behind them (a second load that mostly hits L1), builds the key of its LAST position from the
// codewords of the following symbols and rolls the other 15 out of it, all with compile-time byte positions, and
// stores 16 keys (64 contiguous bytes) and their 16 first digits (one 16-byte store) straight from registers:
// no staging of symbols or keys in LDS, no barrier behind the table load.  The 17 symbols from the last position
// on must hold kKeyBits bits: the host takes this kernel only when every codeword has at least
// kKeys8MinLen bits (17 x 2 >= 32; a code with a 1-bit codeword, a text dominated by one character, keeps the
// generic kernel above).
constexpr int kKeys8Items = 16;
constexpr int kKeys8Tile = RadixCfg<Key0>::kTile;          // the key builder's tiles are the tiles of the round-0 sort,
constexpr int kKeys8Threads = kKeys8Tile / kKeys8Items;   // its workgroups as many lanes as that takes (20 keys per sort lane: 320)
static_assert(kKeys8Threads * kKeys8Items == kKeys8Tile && kKeys8Threads % kWave == 0 && kKeys8Threads >= 256 && kKeys8Threads <= 1024,
              "a whole number of waves, and at least the 256 lanes that load the code table and write the histogram row");
constexpr int kKeys8MinLen = (kKeyBits + kKeys8Items) / (kKeys8Items + 1);
// hist_table != nullptr: the tile of this workgroup is a tile of the round-0 sort: the histogram of the
// sort's first digit — the low byte of the keys — is taken right here (per-wave LDS counters: the tail of a compressed
// codeword stream is near-uniform) and written as that sort's tile row and chunk sums (radix_sort.h, RadixPlan): one
// launch and one byte written and read per key less than through the digit bytes (dig0 is nullptr then).
__global__ __launch_bounds__(kKeys8Threads) void build_keys0_u8_kernel(const uint8_t *__restrict__ sym, size_t n, DevCode code,
                                                                Key0 *__restrict__ keys, uint8_t *__restrict__ dig0,
                                                                uint32_t *__restrict__ hist_table,
                                                                uint32_t *__restrict__ hist_chunk_sums) {
  static_assert(sizeof(Key0) == 4, "register form of the key builder: 32-bit keys");
  __shared__ uint32_t stab[256];  // (len << 16) | codeword
  constexpr int WAVES = kKeys8Threads / kWave;
  __shared__ uint32_t shist[WAVES][kRadixBins];
  const int ub = code.uniform_bits > 0 ? code.uniform_bits : 0;
  if (threadIdx.x < 256) {
    if (hist_table) {
#pragma unroll
      for (int i = 0; i < WAVES; i++) shist[i][threadIdx.x] = 0u;
    }
    stab[threadIdx.x] = ub ? ((static_cast<uint32_t>(ub) << 16) | threadIdx.x)
                           : ((static_cast<uint32_t>(code.len[threadIdx.x]) << 16) | code.cw[threadIdx.x]);
  }
  __syncthreads();
  const size_t p0 = static_cast<size_t>(blockIdx.x) * kKeys8Tile + static_cast<size_t>(threadIdx.x) * kKeys8Items;
  // (the symbol buffer is 256-byte aligned and padded by 16 bytes behind n; positions at or behind n count as symbol 0)
  uint32_t w[8] = {};
  if (p0 < n) {
    const uint4 a = *reinterpret_cast<const uint4 *>(sym + p0);
    w[0] = a.x;
    w[1] = a.y;
    w[2] = a.z;
    w[3] = a.w;
    if (p0 + kKeys8Items <= n) {
      const uint4 b = *reinterpret_cast<const uint4 *>(sym + p0 + kKeys8Items);
      w[4] = b.x;
      w[5] = b.y;
      w[6] = b.z;
      w[7] = b.w;
    } else {
      w[4] = w[5] = w[6] = w[7] = 0u;
    }
    if (p0 + 2 * kKeys8Items > n) {
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const size_t pos = p0 + 4 * static_cast<size_t>(q);
        if (pos >= n) {
          w[q] = 0u;
        } else if (pos + 4 > n) {
          w[q] &= (1u << (8 * static_cast<unsigned>(n - pos))) - 1u;
        }
      }
    }
  }
  auto sym_at = [&](int t) { return (w[t >> 2] >> (8 * (t & 3))) & 0xffu; };  // (t: a compile-time constant everywhere)
  uint32_t key = 0;
  int used = 0;
#pragma unroll
  for (int t = kKeys8Items - 1; t < 2 * kKeys8Items; t++) {  // the key of the lane's last position
    if (used < kKeyBits) {
      const uint32_t e = stab[sym_at(t)];
      const int l = static_cast<int>(e >> 16);
      const int take = min(l, kKeyBits - used);
      key = (take < 32 ? key << take : 0u) | ((e & 0xffffu) >> (l - take));
      used += take;
    }
  }
  uint32_t k[kKeys8Items];
  k[kKeys8Items - 1] = key;
#pragma unroll
  for (int j = kKeys8Items - 2; j >= 0; j--) {  // key(i) = codeword(i) ++ key(i + 1)
    const uint32_t e = stab[sym_at(j)];
    const int l = static_cast<int>(e >> 16);
    key = ((e & 0xffffu) << (kKeyBits - l)) | (key >> l);
    k[j] = key;
  }
  if (hist_table) {
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < kKeys8Items; j++) {
      if (p0 + j < n) atomicAdd(&shist[wv][k[j] & 0xffu], 1u);
    }
    __syncthreads();
    if (threadIdx.x < kRadixBins) {
      uint32_t cnt = 0;
#pragma unroll
      for (int i = 0; i < WAVES; i++) cnt += shist[i][threadIdx.x];
      hist_table[static_cast<size_t>(blockIdx.x) * kRadixBins + threadIdx.x] = cnt;
      if (cnt) atomicAdd(&hist_chunk_sums[static_cast<size_t>(blockIdx.x / kColChunk) * kRadixBins + threadIdx.x], cnt);
    }
  }
  if (p0 >= n) return;
  if (p0 + kKeys8Items <= n) {
#pragma unroll
    for (int q = 0; q < kKeys8Items / 4; q++) {
      *reinterpret_cast<uint4 *>(keys + p0 + 4 * q) = make_uint4(k[4 * q], k[4 * q + 1], k[4 * q + 2], k[4 * q + 3]);
    }
    if (dig0) {
      uint32_t d[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        d[q] = (k[4 * q] & 0xffu) | ((k[4 * q + 1] & 0xffu) << 8) | ((k[4 * q + 2] & 0xffu) << 16) | (k[4 * q + 3] << 24);
      }
      *reinterpret_cast<uint4 *>(dig0 + p0) = make_uint4(d[0], d[1], d[2], d[3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < kKeys8Items; j++) {
      if (p0 + j < n) {
        keys[p0 + j] = k[j];
        if (dig0) dig0[p0 + j] = static_cast<uint8_t>(k[j]);
      }
    }
  }
}

}  // namespace wp
