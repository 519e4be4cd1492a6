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
// (binary search); the first claimant of a range appends the group to the active list of round 1
// (list space and group number from one 64-bit atomic, so that both grow together) — round 0 needs no
// counting and compaction passes for it.
#pragma once
#include "decode.h"
#include "primitives.h"

namespace wp {

constexpr uint32_t kClaimEmpty = 0xffffffffu;

// First kKeyBits bits of the code stream of token symbols cps[0..len) (code points -> dense symbols through lut),
// left aligned; bits_out = bits of the stream inside the key (<= kKeyBits); returns true if the stream is longer
// than the key (a "long" token).  By a whole wave: lane j takes symbol j (a key holds at most kKeyBits symbols), the codeword ends come
// from one wave scan and the key from an OR across the lanes — three dependent loads instead of three per symbol
// (the per-token kernels are chains of dependent loads).  whole_out: codewords that lie in the key completely
// (= count_key_symbols of the key).  All 64 lanes must be active; every lane gets the same results.
__device__ inline bool wave_token_key(const uint32_t *__restrict__ cps, uint32_t len, const uint32_t *__restrict__ lut_excl,
                                      const DevCode &code, uint64_t &key_out, int &bits_out, uint32_t &whole_out) {
  const int ub = code.uniform_bits > 0 ? code.uniform_bits : 0;
  const int lo = code.uniform_bits < 0 ? -code.uniform_bits : 0;
  const uint32_t lomask = (1u << lo) - 1u;
  const uint32_t lane = static_cast<uint32_t>(lane_id());
  uint32_t l = 0;
  uint64_t c = 0;
  if (lane < len) {
    const uint32_t sv = lut_excl[cps[lane]] + 1u;
    if (ub) {
      l = static_cast<uint32_t>(ub);
      c = sv;
    } else {
      const uint32_t hi = sv >> lo;
      l = static_cast<uint32_t>(code.len[hi]) + static_cast<uint32_t>(lo);
      c = (static_cast<uint64_t>(code.cw[hi]) << lo) | (sv & lomask);
    }
  }
  const uint32_t end = wave_incl_sum(l), start = end - l;  // bit range of the lane's codeword in the stream
  uint64_t part = 0;
  if (l && start < static_cast<uint32_t>(kKeyBits)) {
    part = end <= static_cast<uint32_t>(kKeyBits) ? c << (kKeyBits - end) : c >> (end - kKeyBits);
  }
#pragma unroll
  for (int d = 1; d < kWave; d <<= 1) part |= __shfl_xor(part, d, kWave);
  const uint32_t total = __shfl(end, kWave - 1, kWave);  // bits of the first min(len, 64) symbols
  whole_out = static_cast<uint32_t>(__popcll(__ballot(l && end <= static_cast<uint32_t>(kKeyBits))));
  key_out = part;
  bits_out = static_cast<int>(total < static_cast<uint32_t>(kKeyBits) ? total : static_cast<uint32_t>(kKeyBits));
  return total > static_cast<uint32_t>(kKeyBits) || len > static_cast<uint32_t>(kWave);
}

// suffix at text position v against the token (dense symbols through lut): -1 / +1 = the suffix sorts
// before / behind every string that starts with the token, 0 = the token is a prefix of the suffix
template <typename SymT>
__device__ __forceinline__ int suffix_vs_token(const SymT *__restrict__ sym, size_t n, size_t v,
                                               const uint32_t *__restrict__ cps, uint32_t len,
                                               const uint32_t *__restrict__ lut_excl) {
  for (uint32_t j = 0; j < len; j++) {
    const uint32_t a = v + j < n ? static_cast<uint32_t>(sym[v + j]) : 0u;
    const uint32_t b = lut_excl[cps[j]] + 1u;
    if (a != b) return a < b ? -1 : 1;
  }
  return 0;
}

// Text-only layout (S = text . 1, the vocabulary kept out of the suffix sort): the reach of a token
// — the SA slots of the suffixes it is a prefix of, what the reference's stack pops delimit
// (linear.cpp:161-189) — is the equal range of its code stream in the sorted keys.  rng_lo/rng_hi
// (nullptr: not wanted) receive it for tokens whose stream fits the key; for long tokens they receive
// the group that carries the token's key (refined by the rounds, then narrowed by
// long_token_range_kernel) and rng_long[m] = 1.
// next active list (suffix_array.h): slots, suffixes, group numbers, depths, group table
struct NeededList {
  uint32_t *slots, *vals, *gid, *dep, *ghead;
  uint32_t *gfirst, *gdepth;   // per group: its first slot and its depth (the entries are written by needed_fill_kernel)
  unsigned long long *totals;  // low word: list entries, high word: groups (one atomic allocates both)
  uint32_t *sa;                // != nullptr: slot -> suffix for the slots of needed groups (text-only layout)
  uint32_t need_depth;         // a group whose depth reaches this needs no refinement (depth cap)
  // depth each needed group has to reach: 1 + the longest eligible token that carries its key (all of them meet
  // in the group's slot of the claim table: claim_need, by atomic max); gclaim: that slot, gneed: the result per
  // group (needed_need_kernel) — what DepthRule::gneed_in starts from
  uint32_t *claim_need, *gclaim, *gneed;
};

template <typename SymT>
__global__ __launch_bounds__(kBlock) void need_groups_kernel(const Key0 *__restrict__ keys,
                                                             const uint32_t *__restrict__ vals, size_t n,
                                                             const SymT *__restrict__ sym,
                                                             const uint32_t *__restrict__ vocab_cps,
                                                             const uint32_t *__restrict__ tok_start,
                                                             const uint32_t *__restrict__ tok_info, int M,
                                                             const uint32_t *__restrict__ lut_excl, DevCode code,
                                                             uint32_t *__restrict__ claim, uint32_t claim_mask,
                                                             NeededList out, uint32_t *__restrict__ rng_lo,
                                                             uint32_t *__restrict__ rng_hi,
                                                             uint8_t *__restrict__ rng_long) {
  const int m = static_cast<int>((static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x) >> 6);
  const int lane = lane_id();
  if (m >= M) return;
  uint64_t key = 0;
  int bits = 0;
  const uint32_t len = tok_info[m] & 0x0fffffffu;
  const uint32_t *cps = vocab_cps + tok_start[m];
  uint32_t whole = 0;
  const bool is_long = wave_token_key(cps, len, lut_excl, code, key, bits, whole);  // wave-uniform
  if (!is_long) {
    if (!rng_lo) return;
    // (the whole wave searches: wave_key_lower_bound)
    const size_t lb = wave_key_lower_bound(keys, 0, n, key, 2);
    const uint64_t step = 1ull << (kKeyBits - bits);
    const uint64_t above = key + step;  // first key that no longer starts with the stream
    const size_t ubd = (bits == 0 || (above >> kKeyBits) != 0) ? n : wave_key_gallop(keys, lb, n, above);
    if (lane == 0) {
      rng_lo[m] = static_cast<uint32_t>(lb);
      rng_hi[m] = static_cast<uint32_t>(ubd);
      rng_long[m] = 0;
    }
    return;
  }
  const size_t first = wave_key_lower_bound(keys, 0, n, key, 2);
  const size_t last = wave_key_gallop(keys, first, n, key + 1);
  if (last - first < 2) {  // no such suffix, or a single one: nothing to refine
    if (rng_lo && lane == 0) {
      size_t lb = first, ubd = first;
      if (last > first) {
        const int cmp = suffix_vs_token(sym, n, vals[first], cps, len, lut_excl);
        lb = cmp < 0 ? first + 1 : first;
        ubd = cmp <= 0 ? first + 1 : first;
      }
      rng_lo[m] = static_cast<uint32_t>(lb);
      rng_hi[m] = static_cast<uint32_t>(ubd);
      rng_long[m] = 0;
    }
    return;
  }
  if (rng_lo && lane == 0) {
    rng_lo[m] = static_cast<uint32_t>(first);
    rng_hi[m] = static_cast<uint32_t>(last);
    rng_long[m] = 1;
  }
  // several tokens share a key (all long prefixes of one word): the first to claim the range appends it
  unsigned long long got = ~0ull;
  uint32_t depth = 0, claim_slot = 0;
  if (lane == 0) {
    uint32_t h = (static_cast<uint32_t>(first) * 2654435761u) & claim_mask;
    bool won = false;
    for (;;) {
      const uint32_t old = atomicCAS(&claim[h], kClaimEmpty, static_cast<uint32_t>(first));
      if (old == kClaimEmpty) {
        won = true;
        break;
      }
      if (old == static_cast<uint32_t>(first)) break;
      h = (h + 1) & claim_mask;
    }
    atomicMax(&out.claim_need[h], min(len, out.need_depth - 1u) + 1u);
    claim_slot = h;
    if (won) {
      depth = whole;  // (= count_key_symbols(key, kKeyBits, ...): the codewords that lie in the key completely)
      if (depth < out.need_depth) {
        got = atomicAdd(out.totals, (1ull << 32) | static_cast<unsigned long long>(last - first));
      }
    }
  }
  if (lane == 0 && got != ~0ull) {  // the entries themselves: needed_fill_kernel, one thread per entry
    const uint32_t g = static_cast<uint32_t>(got >> 32);
    out.ghead[g] = static_cast<uint32_t>(got);
    out.gfirst[g] = static_cast<uint32_t>(first);
    out.gdepth[g] = depth;
    out.gclaim[g] = claim_slot;
  }
}

__global__ __launch_bounds__(kBlock) void needed_need_kernel(NeededList out) {
  const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
  if (g < reinterpret_cast<const uint32_t *>(out.totals)[1]) out.gneed[g] = out.claim_need[out.gclaim[g]];
}

// entries of the needed groups: list position p belongs to the group g with ghead[g] <= p < ghead[g+1]
// (ghead ascends with g: both come from one atomic) and is slot gfirst[g] + p - ghead[g]
__global__ __launch_bounds__(kBlock) void needed_fill_kernel(NeededList out, const uint32_t *__restrict__ sorted_vals,
                                                             size_t n) {
  const uint32_t *t32 = reinterpret_cast<const uint32_t *>(out.totals);
  const uint32_t n_act = t32[0], n_groups = t32[1];
  // a wave takes 64 consecutive list positions at a time: a 64-way search finds the group of the first one (3
  // dependent loads for 2e5 groups), the others lie at most 63 groups further on (a group has at least one entry)
  // — a binary search over all groups per entry made this kernel a chain of ~14 dependent loads
  const int lane = lane_id();
  for (size_t p0 = (static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x) & ~static_cast<size_t>(kWave - 1); p0 < n_act;
       p0 += static_cast<size_t>(gridDim.x) * kBlock) {
    const size_t p = p0 + lane;
    // first group whose successor starts behind p0, i.e. the last g with ghead[g] <= p0
    uint32_t a = 0, b = n_groups - 1;  // answer in [a, b]
    while (a < b) {
      const uint32_t st = (b - a) / kWave + 1;
      const uint32_t idx = a + static_cast<uint32_t>(lane) * st;
      const bool gt = idx < b ? out.ghead[idx + 1] > p0 : true;  // (group idx ends behind p0)
      const uint64_t mm = __ballot(gt);
      if (!mm) {
        a += (kWave - 1) * st + 1;
        continue;
      }
      const uint32_t t = static_cast<uint32_t>(__ffsll(static_cast<long long>(mm)) - 1);
      const uint32_t first = a + t * st;
      if (t) a += (t - 1) * st + 1;
      b = first < b ? first : b;
      if (!t) b = a;
    }
    if (p >= n_act) continue;
    uint32_t lo = a, hi = min(n_groups, a + static_cast<uint32_t>(kWave));  // last group in [lo, hi) with ghead[g] <= p
    while (hi - lo > 1) {
      const uint32_t md = (lo + hi) >> 1;
      if (out.ghead[md] <= p) lo = md; else hi = md;
    }
    const uint32_t k = out.gfirst[lo] + static_cast<uint32_t>(p - out.ghead[lo]);
    if (!wp_in_bounds(k < n, kSiteListSlot)) continue;
    const uint32_t v = sorted_vals[k];
    out.slots[p] = k;
    out.vals[p] = v;
    out.gid[p] = lo;
    out.dep[p] = out.gdepth[lo];
    if (out.sa) out.sa[k] = v;
  }
}

// group g of the list is [ghead[g], ghead[g+1]): the closing entry
__global__ void needed_list_close_kernel(const uint32_t *__restrict__ totals, uint32_t *__restrict__ ghead) {
  ghead[totals[1]] = totals[0];
}

// marks of the text-only layout for the step functions of scanline.h: a token stands in front of the
// first slot of its range and covers [lo, hi) in the left-to-right sense only
__global__ __launch_bounds__(kBlock) void virtual_marks_kernel(const uint32_t *__restrict__ rng_lo,
                                                               const uint32_t *__restrict__ rng_hi, int M,
                                                               const int32_t *__restrict__ tok_id,
                                                               const uint32_t *__restrict__ tok_info,
                                                               uint32_t *__restrict__ mslot, int32_t *__restrict__ mid,
                                                               uint32_t *__restrict__ minfo,
                                                               int32_t *__restrict__ reach_fwd,
                                                               int32_t *__restrict__ reach_bwd) {
  const int m = blockIdx.x * kBlock + threadIdx.x;
  if (m >= M) return;
  const uint32_t lo = rng_lo[m];
  mslot[m] = lo;
  mid[m] = tok_id[m];
  minfo[m] = tok_info[m];
  reach_fwd[m] = static_cast<int32_t>(max(rng_hi[m], lo));
  reach_bwd[m] = static_cast<int32_t>(lo);  // nothing in front of the mark is covered
}

}  // namespace wp
