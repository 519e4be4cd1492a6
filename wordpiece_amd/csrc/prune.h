// prune.h — which tied groups of round 0 have to be refined at all (depth-capped mode).
//
// After round 0 a tied group G shares d symbols (the complete codewords inside its 63-bit key).  The
// scanlines (linear.cpp:161-213) only ever compare an LCP with the length of an eligible vocab token,
// and a token t longer than d can only be a prefix of a member of G if t's own code stream starts
// with G's key — in other words if the key of t (its first 63 bits) equals G's key.  If no eligible
// token with a code stream of more than 63 bits carries G's key, every token that matches a member of
// G is at most d long, matches all members alike, and the order and LCPs inside G can stay
// undetermined ("LCP >= anything asked for") exactly as for the groups that retire at the depth cap:
// token ids are unchanged.  (Same-class duplicate tokens force the full suffix array, as before.)
// For a BERT-style vocabulary only a few hundred tokens are that long, and 29 % of all suffixes
// being tied after round 0 turns into a few thousand list entries; in the deep-prefix stress
// (config 5) only the word starts (1/512 of the positions) stay instead of everything.
//
// need_groups_kernel: one wave per eligible token: key of the token, equal range in the sorted keys
// (binary search), first claimant of a range fills need[lo..hi) with 1.
#pragma once
#include "decode.h"
#include "primitives.h"

namespace wp {

constexpr uint32_t kClaimEmpty = 0xffffffffu;

// first kKeyBits bits of the code stream of token symbols cps[0..len) (code points -> dense symbols
// through lut); returns false if the whole stream fits into the key (the token is not "long")
__device__ inline bool token_key(const uint32_t *__restrict__ cps, uint32_t len, const uint32_t *__restrict__ lut_excl,
                                 const DevCode &code, uint64_t &key_out) {
  const int ub = code.uniform_bits > 0 ? code.uniform_bits : 0;
  const int lo = code.uniform_bits < 0 ? -code.uniform_bits : 0;
  const uint32_t lomask = (1u << lo) - 1u;
  uint64_t key = 0;
  int used = 0;
  bool overflow = false;
  for (uint32_t j = 0; j < len; j++) {
    const uint32_t sv = lut_excl[cps[j]] + 1u;
    int l;
    uint32_t c;
    if (ub) {
      l = ub;
      c = sv;
    } else {
      const uint32_t hi = sv >> lo;
      l = static_cast<int>(code.len[hi]) + lo;
      c = (static_cast<uint32_t>(code.cw[hi]) << lo) | (sv & lomask);
    }
    if (used + l > kKeyBits) {
      const int take = kKeyBits - used;
      if (take > 0) key = (key << take) | (c >> (l - take));
      used = kKeyBits;
      overflow = true;
      break;
    }
    key = (key << l) | c;
    used += l;
  }
  key_out = key << (kKeyBits - used);
  return overflow;
}

__global__ __launch_bounds__(kBlock) void need_groups_kernel(const uint64_t *__restrict__ keys, size_t n,
                                                             const uint32_t *__restrict__ vocab_cps,
                                                             const uint32_t *__restrict__ tok_start,
                                                             const uint32_t *__restrict__ tok_info, int M,
                                                             const uint32_t *__restrict__ lut_excl, DevCode code,
                                                             uint32_t *__restrict__ claim, uint32_t claim_mask,
                                                             uint8_t *__restrict__ need,
                                                             unsigned long long *__restrict__ n_needed) {
  const int m = static_cast<int>((static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6);
  const int lane = lane_id();
  if (m >= M) return;
  uint64_t key = 0;
  const uint32_t len = tok_info[m] & 0x0fffffffu;
  if (!token_key(vocab_cps + tok_start[m], len, lut_excl, code, key)) return;  // wave-uniform
  // equal range of `key` in the sorted keys (every lane runs the same search: the loads broadcast)
  size_t lo = 0, hi = n;
  while (lo < hi) {
    const size_t md = (lo + hi) >> 1;
    if (keys[md] < key) lo = md + 1; else hi = md;
  }
  const size_t first = lo;
  hi = n;
  while (lo < hi) {
    const size_t md = (lo + hi) >> 1;
    if (keys[md] <= key) lo = md + 1; else hi = md;
  }
  const size_t last = lo;
  if (last - first < 2) return;  // no such suffix, or a singleton: nothing to refine
  // several tokens share a key (all long prefixes of one word): the first to claim the range fills it
  int won = 0;
  if (lane == 0) {
    uint32_t h = (static_cast<uint32_t>(first) * 2654435761u) & claim_mask;
    for (;;) {
      const uint32_t old = atomicCAS(&claim[h], kClaimEmpty, static_cast<uint32_t>(first));
      if (old == kClaimEmpty) {
        won = 1;
        break;
      }
      if (old == static_cast<uint32_t>(first)) break;
      h = (h + 1) & claim_mask;
    }
    if (won && n_needed) atomicAdd(n_needed, static_cast<unsigned long long>(last - first));
  }
  won = __shfl(won, 0, kWave);
  if (!won) return;
  for (size_t k = first + lane; k < last; k += kWave) need[k] = 1;
}

}  // namespace wp
