// linear_path.h — one encode of the Linear WordPiece path on one device context, stage by stage.
//
// Stages (reference file:line in parentheses):
//   decode + classes + alphabet   utils.cpp:37-79, utf8.cpp:54-90, linear.cpp:83-103        decode.h      encode_on_device
//   dense symbols, code, keys     linear.cpp:77-103                                          decode.h      symbols_and_keys
//   round 0 of the suffix sort    linear.cpp:118-141 (libsais_int)                           radix_sort.h  sort_round0
//   ranks, needed groups          linear.cpp:144-147 (inverse SA), prune.h                   suffix_array.h ranks_round0
//   refinement                    trie.h (default layout) / doubling rounds + LCP            local_sort.h  refine
//   who marks + 4 scanlines       linear.cpp:153-213                                         scanline.h    scanlines
//   greedy walk + id stream       linear.cpp:215-316                                         walk.h        walk
#pragma once
#include <functional>

#include "context.h"
#include "local_sort.h"
#include "prune.h"
#include "trie.h"
#include "walk.h"

namespace wp {

// The needed list of round 0 (prune.h) is sized from a guess (the handle remembers what the last encode needed); when
// a text needs more, the device keeps the list empty, reports the length it wanted, and the encode runs again with room.
struct ListOverflow {
  size_t wanted;
};

// needed list longer than the room it was given: empty it (every later kernel reads its sizes on the device) and report
__global__ void needed_list_clamp_kernel(uint32_t *__restrict__ totals, uint32_t cap, uint32_t *__restrict__ wanted) {
  if (totals[0] > cap) {
    *wanted = totals[0];
    totals[0] = 0;
    totals[1] = 0;
  }
}

template <typename SymT>
struct LinearPath {
  // ---- what the decode phase hands over
  const wp_vocab *v;
  Context *c;
  wp_stats &S;
  Arena &ar, &aa;
  const uint8_t *d_text;
  size_t nbytes;
  const uint32_t *d_tile_prefix;
  size_t n_text, n;
  uint32_t *d_cps;
  uint8_t *d_cls;
  int bits;
  bool text_only;

  // ---- derived
  hipStream_t st, st2;
  const HostVocab &hv;
  bool full, prune, use_trie, window_store, use_digit_bytes, staged_possible, sparse_emit;
  uint32_t need_depth;
  int M, P, bucket_shift, bucket_shift_all, hb_n, win_mid;
  unsigned sl_tiles, sl_groups, nbuckets, nbuckets_all;
  size_t radix_words, emit_tiles, walk_blocks, claim_size, list_cap;

  // ---- device buffers.  n-sized "slabs" of 4 bytes per symbol change roles from stage to stage (see plan()).
  SymT *d_sym = nullptr, *d_vsym = nullptr;
  Key0 *KA = nullptr, *KB = nullptr;           // round-0 keys, ping-pong
  uint32_t *VA = nullptr, *VB = nullptr;       // round-0 values (suffix starts), ping-pong
  uint32_t *X0 = nullptr, *X1 = nullptr;       // scratch pair of the rank store; walk stage: see walk()
  uint8_t *DG0 = nullptr, *DG1 = nullptr;      // digit bytes
  RankEntry *d_rank = nullptr;
  uint32_t *d_node_of_slot = nullptr;          // trie refinement: end node of the suffix in a slot of a needed group
  int32_t *d_lcp = nullptr;                    // reference layout / debug
  uint32_t *d_sa = nullptr, *d_gdepth = nullptr;
  RankEntry *d_hd_n = nullptr;                 // reference layout: round 0's rank entries in suffix order
  int32_t *d_emit = nullptr, *d_dbg_best = nullptr;
  uint32_t *d_anchors = nullptr, *d_anchor_cnt = nullptr, *d_anchor_tmp = nullptr, *d_emit_cnt = nullptr, *d_emit_tmp = nullptr,
           *d_blk_cnt = nullptr, *d_blk_off = nullptr, *d_tile_scratch = nullptr;
  // list-sized (list_cap entries): the active list of the rounds and the large-group path
  uint32_t *AS0 = nullptr, *AS1 = nullptr, *AG = nullptr, *AD0 = nullptr, *AD1 = nullptr, *LA = nullptr, *LB = nullptr, *d_tdep = nullptr;
  uint64_t *LK0 = nullptr, *LK1 = nullptr, *LK2 = nullptr;  // sorted list keys | rank entries + scratch, large-list keys (and the widened round-0 keys), ping-pong
  uint32_t *LV0 = nullptr, *LV1 = nullptr, *LPOS = nullptr;
  uint32_t *d_ghead = nullptr, *d_gneed0 = nullptr, *d_gneed1 = nullptr, *d_lg_head = nullptr, *d_lg_off = nullptr;
  RerankAgg *d_agg = nullptr, *d_chunk_agg = nullptr;
  // vocabulary-sized
  uint32_t *d_claim = nullptr, *d_claim_need = nullptr, *d_gclaim = nullptr, *d_gfirst = nullptr, *d_gdep = nullptr, *d_gnode = nullptr,
           *d_gdone = nullptr, *d_rng_lo = nullptr, *d_rng_hi = nullptr, *d_child_sym = nullptr, *d_radix_tmp = nullptr,
           *d_radix_tmp2 = nullptr;  // (second: the large-group sort of the trie round runs on the side stream beside the rank store)
  uint8_t *d_rng_long = nullptr;
  uint32_t *d_mslot0 = nullptr, *d_mslot1 = nullptr, *d_midx0 = nullptr, *d_midx1 = nullptr, *d_minfo = nullptr, *d_tile_mlo = nullptr;
  int32_t *d_mid = nullptr, *d_rf = nullptr, *d_rb = nullptr, *d_cover_f = nullptr, *d_cover_b = nullptr;
  int32_t *d_tmin_f = nullptr, *d_tmin_b = nullptr, *d_gmin_f = nullptr, *d_gmin_b = nullptr, *d_pval_p = nullptr, *d_pval_s = nullptr;
  uint32_t *d_ps0 = nullptr, *d_ps1 = nullptr, *d_pv0 = nullptr, *d_pv1 = nullptr, *d_bidx = nullptr;
  int2 *d_bfast = nullptr, *d_bfast_all = nullptr;
  uint32_t *d_bidx_all = nullptr;

  // ---- state handed from stage to stage
  SymbolCode code;
  DevCode dcode{};
  DigitBytes db;
  RadixPlan sort_plan;        // round-0 sort: set up ahead of it when the key builder takes its first histogram
  bool hist_in_keys = false;
  int cur = 0, rounds = 1;
  Key0 *keys = nullptr, *other_keys = nullptr;
  uint32_t *vals = nullptr, *other_vals = nullptr, *slots = nullptr, *other_slots = nullptr, *adep = nullptr, *other_dep = nullptr;
  uint32_t *avals = nullptr, *spare_vals = nullptr;
  bool classified = false;
  size_t n_act = 0, n_large = 0, n_large_groups = 0;
  StepTable steps{}, steps_all{};  // (steps_all: for the kernels that look up EVERY position of a stretch, scanlines())

  LinearPath(const wp_vocab *v_, Context *c_, wp_stats &S_, Arena &ar_, Arena &aa_, const uint8_t *text, size_t nb,
             const uint32_t *tile_prefix, size_t n_text_, size_t n_, uint32_t *cps, uint8_t *cls, int bits_, bool text_only_)
      : v(v_), c(c_), S(S_), ar(ar_), aa(aa_), d_text(text), nbytes(nb), d_tile_prefix(tile_prefix), n_text(n_text_), n(n_),
        d_cps(cps), d_cls(cls), bits(bits_), text_only(text_only_), st(c_->stream), st2(c_->stream2), hv(v_->hv) {
    full = v->full_depth || hv.n_dup_eligible > 0 || v->lcp_kasai;
    need_depth = static_cast<uint32_t>(std::min<int64_t>(hv.longest + 1, 0x7fffffff));
    M = static_cast<int>(hv.elig_id.size());
    // round 0 only keeps the tied groups that carry the key of a long eligible token (prune.h)
    prune = !full && (M > 0 || text_only);
    // default layout: the needed groups are resolved along the token trie (trie.h) instead of by doubling rounds
    use_trie = text_only && !full && M > 0;
    sl_tiles = cdiv(n, kSlTile);
    sl_groups = cdiv(sl_tiles, kSlGroup);
    P = kStepsPerMark * M + 1;  // steps of the scanline result (scanline.h)
    // buckets of the step table's index: about four per step (most lookups then end at the bucket entry, scanline.h)
    const int bucket_bits = std::min(kStepBucketBitsMax, std::max(kStepBucketBits, bit_length(4 * static_cast<size_t>(P))));
    bucket_shift = std::max(0, bit_length(n) - bucket_bits);
    nbuckets = static_cast<unsigned>(((n - 1) >> bucket_shift) + 1);
    // ... and a second index of 2^18 buckets (2 MB: stays in the L2) for the kernels that look up every position of a
    // long word, on the chain or not — their ranks spread over all slots, and 16 MB of entries miss the L2 (config 5:
    // 13.0 against 11.0 ms; the coverage rule's reach kernel, whose positions nearly all ARE visited, is faster on the
    // fine index: config 3, 9.3 against 9.8 ms)
    bucket_shift_all = std::max(0, bit_length(n) - kStepBucketBits);
    nbuckets_all = static_cast<unsigned>(((n - 1) >> bucket_shift_all) + 1);
    radix_words = std::max(radix_tmp_words<uint64_t>(n), radix_tmp_words<uint32_t>(std::max<size_t>(n, kStepsPerMark * std::max(M, 1) + 1)));
    emit_tiles = cdiv(std::max<size_t>(n_text, 1), kScanTile);
    walk_blocks = cdiv(std::max<size_t>(n_text, 1), kBlock);  // (at most one anchor per position)
    sparse_emit = v->sparse_emit || EnvOptions::get().sparse_emit;
    // ids as per-workgroup lists (walk.h, StagedOut) unless several kernels contribute ids: decided here, except
    // for long words, which only the anchor gaps reveal
    staged_possible = !sparse_emit && !v->cover_anchors && hv.soft.empty();
    // digit bytes (radix_sort.h): the round-0 sort's histograms read 1 byte per key instead of the key
    use_digit_bytes = n > kRadixSmallN;
    claim_size = 1024;  // hash table of claimed key ranges (prune.h): a power of two >= 2 M
    while (claim_size < 2 * static_cast<size_t>(std::max(M, 1))) claim_size *= 2;
    // Round 0 stores a rank for every position: a permutation.  From 2^22 symbols on the list is partitioned by ALL
    // destination bits above kWinBits (one or two radix passes over 8-byte records) and every 2^kWinBits-slot window of
    // the rank table is assembled in LDS and written with full-width stores (window_store_kernel).
    hb_n = bit_length(n - 1);
    window_store = n >= (1u << 22);
    // (more than 8 bits above the window: two passes of about half the bits each — with uniform digits the runs a
    // tile appends to its bins are 4096 / bins entries, and 16-entry runs leave the workgroup as half lines)
    win_mid = hb_n - kWinBits <= kRadixBits ? hb_n : kWinBits + (hb_n - kWinBits + 1) / 2;
    // room for the active list: every suffix where every tied group goes on (full depth, no pruning), else the handle's
    // memory of the last encode or an eighth of the text
    if (!prune) {
      list_cap = n;
    } else {
      const size_t guess = c->list_hint ? c->list_hint : n / 8;
      list_cap = std::min(n, guess + guess / 4 + 65536);
    }
    S.symbol_bits = bits;
    S.full_depth = full;
    S.trie_refine = use_trie ? 1 : 0;
    S.key_bits = kKeyBits;
  }

  // st2 starts after everything queued on st so far / st continues after everything queued on st2
  void fork() {
    WP_HIP(hipEventRecord(c->evs[0], st));
    WP_HIP(hipStreamWaitEvent(st2, c->evs[0], 0));
  }
  void join() {
    WP_HIP(hipEventRecord(c->evs[1], st2));
    WP_HIP(hipStreamWaitEvent(st, c->evs[1], 0));
  }

  // ---- HBM layout (arena B) -------------------------------------------------------------------------------------
  // Per symbol, default layout with 8-bit symbols: sym 1, keys 4 + 4, values 4 + 4, digit bytes 1 + 1, rank-store
  // scratch 4 + 4, rank 4, trie nodes by slot 4, id scratch 4, anchors 4 = 43 B, + ~80 B per entry of the active list
  // (an eighth of the text unless the last encode needed more) + vocabulary-sized tables.  The reference layout and
  // the debug views add their own arrays.  Roles of the n-sized slabs by stage:
  //   sort        KA/KB keys, VA/VB values
  //   rank store  X0/X1 first partition pass, (other values, keys) second, d_rank <- windows
  //   walk        see walk(): ids, id lists, wide / long-word / coverage scratch in KA KB VA VB X0 X1
  void plan() {
    const bool ref = !text_only;                       // S = text . 1 . vocab: LCP-driven scanlines, doubling rounds
    const bool want_lcp = ref || v->keep_debug;        // (decided by flags, never by a pointer: the planning pass hands out nullptr)
    const size_t lc = list_cap, lg = lc / 2 + static_cast<size_t>(M) + 8;  // list entries, groups
    const size_t rr_tiles_l = cdiv(lc, kRrTile);
    const size_t rtiles = cdiv(std::max<size_t>(n_text, 1), kReachTile);
    for (int pass = 0; pass < 2; pass++) {
      d_sym = ar.take<SymT>(n + 16);
      KA = ar.take<Key0>(n + 16);
      KB = ar.take<Key0>(n + 16);
      VA = ar.take<uint32_t>(n + 16);
      VB = ar.take<uint32_t>(n + 16);
      X0 = ar.take<uint32_t>(n + 16);
      X1 = ar.take<uint32_t>(n + 16);
      DG0 = use_digit_bytes ? ar.take<uint8_t>(n + 64) : nullptr;
      DG1 = use_digit_bytes ? ar.take<uint8_t>(n + 64) : nullptr;
      d_rank = ar.take<RankEntry>(n);
      d_node_of_slot = use_trie ? ar.take<uint32_t>(n) : nullptr;
      d_lcp = want_lcp ? ar.take<int32_t>(n) : nullptr;  // (default layout: nothing reads LCPs)
      d_sa = (v->keep_debug || v->lcp_kasai) ? ar.take<uint32_t>(n) : nullptr;
      d_gdepth = (use_trie && !want_lcp) ? nullptr : ar.take<uint32_t>(n);  // group depths: the doubling rounds' (and the debug LCPs')
      d_hd_n = (want_lcp || !prune) ? ar.take<RankEntry>(n + 16) : nullptr;
      d_dbg_best = v->keep_debug ? ar.take<int32_t>(2 * n) : nullptr;
      d_emit = ar.take<int32_t>(n_text + 1);
      d_anchors = ar.take<uint32_t>(n_text + 1);
      d_anchor_cnt = ar.take<uint32_t>(emit_tiles + 1);
      d_anchor_tmp = ar.take<uint32_t>(cdiv(emit_tiles, kScanTile) + 8);
      d_emit_cnt = ar.take<uint32_t>(emit_tiles + 1);
      d_emit_tmp = ar.take<uint32_t>(cdiv(walk_blocks, kScanTile) + 8);  // (walk_blocks >= emit_tiles)
      d_blk_cnt = ar.take<uint32_t>(walk_blocks + 2);
      d_blk_off = ar.take<uint32_t>(walk_blocks + 2);
      d_tile_scratch = ar.take<uint32_t>(5 * (rtiles + 2) + 4 * (n_text / kMaxAnchorGap + 4));  // coverage tiles / long-word table
      // the active list
      AS0 = ar.take<uint32_t>(lc);
      AS1 = ar.take<uint32_t>(lc);
      AG = ar.take<uint32_t>(lc);
      AD0 = ar.take<uint32_t>(lc);
      AD1 = ar.take<uint32_t>(lc);
      LA = ar.take<uint32_t>(lc);
      LB = ar.take<uint32_t>(lc);
      d_tdep = ar.take<uint32_t>(lc + 2);
      LK0 = ar.take<uint64_t>(lc + 2);
      LK1 = ar.take<uint64_t>(lc + 2);
      LK2 = ar.take<uint64_t>(lc + 2);
      LV0 = ar.take<uint32_t>(lc);
      LV1 = ar.take<uint32_t>(lc);
      LPOS = ar.take<uint32_t>(lc + 16);
      d_ghead = ar.take<uint32_t>(lg);
      d_gneed0 = use_trie ? nullptr : ar.take<uint32_t>(lg);
      d_gneed1 = use_trie ? nullptr : ar.take<uint32_t>(lg);
      d_lg_head = ar.take<uint32_t>(lc / kLsMaxGroup + 4);
      d_lg_off = ar.take<uint32_t>(lc / kLsMaxGroup + 4);
      d_agg = ar.take<RerankAgg>(rr_tiles_l + 1);
      d_chunk_agg = ar.take<RerankAgg>(cdiv(rr_tiles_l, kRrChunk) + 1);
      d_radix_tmp = ar.take<uint32_t>(radix_words);
      d_radix_tmp2 = use_trie ? ar.take<uint32_t>(radix_tmp_words<uint64_t>(lc)) : nullptr;
      // vocabulary-sized
      d_claim = ar.take<uint32_t>(claim_size);
      d_claim_need = ar.take<uint32_t>(claim_size);
      d_gclaim = ar.take<uint32_t>(M + 4);
      d_gfirst = ar.take<uint32_t>(M + 4);  // per needed group (at most one per long token): first slot,
      d_gdep = ar.take<uint32_t>(M + 4);    // depth (whole codewords of the key),
      d_gnode = ar.take<uint32_t>(M + 4);   // trie node its members share and the symbols behind it (trie.h)
      d_gdone = ar.take<uint32_t>(M + 4);
      d_rng_lo = ar.take<uint32_t>(M + 1);
      d_rng_hi = ar.take<uint32_t>(M + 1);
      d_rng_long = ar.take<uint8_t>(M + 1);
      d_vsym = use_trie ? ar.take<SymT>(hv.stream.size() + 16) : nullptr;
      d_child_sym = use_trie ? ar.take<uint32_t>(hv.lt_child_cp.size() + 1) : nullptr;
      d_mslot0 = ar.take<uint32_t>(M + 1);
      d_mslot1 = ar.take<uint32_t>(M + 1);
      d_midx0 = ar.take<uint32_t>(M + 1);
      d_midx1 = ar.take<uint32_t>(M + 1);
      d_mid = ar.take<int32_t>(M + 1);
      d_minfo = ar.take<uint32_t>(M + 1);
      d_rf = ar.take<int32_t>(M + 1);
      d_cover_f = ar.take<int32_t>(2 * static_cast<size_t>(M) + 2);
      d_cover_b = ar.take<int32_t>(2 * static_cast<size_t>(M) + 2);
      d_rb = ar.take<int32_t>(M + 1);
      d_tile_mlo = ref ? ar.take<uint32_t>(sl_tiles + 2) : nullptr;
      d_tmin_f = ref ? ar.take<int32_t>(sl_tiles + 1) : nullptr;
      d_tmin_b = ref ? ar.take<int32_t>(sl_tiles + 1) : nullptr;
      d_gmin_f = ref ? ar.take<int32_t>(sl_groups + 1) : nullptr;
      d_gmin_b = ref ? ar.take<int32_t>(sl_groups + 1) : nullptr;
      d_ps0 = ar.take<uint32_t>(P + 1);
      d_ps1 = ar.take<uint32_t>(P + 1);
      d_pv0 = ar.take<uint32_t>(P + 1);
      d_pv1 = ar.take<uint32_t>(P + 1);
      d_pval_p = ar.take<int32_t>(P + 1);
      d_pval_s = ar.take<int32_t>(P + 1);
      d_bidx = ar.take<uint32_t>(static_cast<size_t>(nbuckets) + 2);
      d_bfast = ar.take<int2>(static_cast<size_t>(nbuckets) + 1);
      d_bidx_all = bucket_shift_all != bucket_shift ? ar.take<uint32_t>(static_cast<size_t>(nbuckets_all) + 2) : nullptr;
      d_bfast_all = bucket_shift_all != bucket_shift ? ar.take<int2>(static_cast<size_t>(nbuckets_all) + 1) : nullptr;
      if (pass == 0) ar.commit();
    }
    ar.arm(st);
  }

  // side stream: the anchor list (and the cleared emit array of the sparse id path) only need the class bytes; they
  // run next to the small latency-bound kernels of the scanline stage, not next to the radix passes
  void launch_anchors() {
    const unsigned atiles = cdiv(n_text, kAnchorTile);
    if (!staged_possible) WP_HIP(hipMemsetAsync(d_emit, 0x80, n_text * sizeof(int32_t), st2));
    hipLaunchKernelGGL(anchor_count_kernel, dim3(atiles), dim3(kBlock), 0, st2, d_cls, static_cast<const uint8_t *>(nullptr),
                       n_text, d_anchor_cnt);
    device_exclusive_scan(d_anchor_cnt, d_anchor_cnt, atiles, d_anchor_tmp, c->d_scalars + 10, st2);
    hipLaunchKernelGGL(anchor_write_kernel, dim3(atiles), dim3(kBlock), 0, st2, d_cls, static_cast<const uint8_t *>(nullptr),
                       n_text, d_anchor_cnt, d_anchors);
    hipLaunchKernelGGL(anchor_gap_kernel, dim3(std::min<size_t>(atiles, 1024)), dim3(kBlock), 0, st2, d_anchors,
                       c->d_scalars + 10, n_text, d_cls, hv.soft.empty() ? 1 : 0, c->d_scalars + 11);
  }

  // ---- S build: dense symbols, symbol code, round-0 keys (linear.cpp:77-103) -----------------------------------------
  void symbols_and_keys() {
    // alphabets > 255: the code covers symbol >> lo_bits (<= 256 values), the low bits follow verbatim
    const int lo_bits = sizeof(SymT) == 1 ? 0 : std::max(0, bits - 8);
    constexpr int kCodeReuse = 64;
    const bool reuse_code = c->code_cached && c->code_alphabet == static_cast<uint32_t>(S.alphabet) && c->code_bits == bits &&
                            c->code_lo == lo_bits && c->code_uses < kCodeReuse;
    if (!reuse_code) WP_HIP(hipMemsetAsync(c->d_symhist, 0, sizeof(uint32_t) * 256, st));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(decode_write_kernel<SymT>), dim3(cdiv(nbytes, kDecTile)), dim3(kBlock), 0, st, d_text,
                       nbytes, d_tile_prefix, c->d_lut, d_sym, d_cls, d_cps, c->d_cls_bmp, c->d_soft,
                       static_cast<int>(hv.soft.size()), reuse_code ? nullptr : c->d_symhist, lo_bits);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(map_vocab_symbols_kernel<SymT>), dim3(cdiv(n - n_text, kBlock)), dim3(kBlock), 0, st,
                       c->d_stream, n_text, n, c->d_lut, d_sym);
    if (reuse_code) {
      code = c->code_cache;  // (the device tables still hold it)
      c->code_uses++;
    } else {
      // frequencies of symbol >> lo_bits -> optimal order-preserving code (host, <= 256 items) -> device tables
      std::vector<uint32_t> h32(256);
      WP_HIP(hipMemcpyAsync(h32.data(), c->d_symhist, sizeof(uint32_t) * 256, hipMemcpyDeviceToHost, st));
      WP_HIP(hipStreamSynchronize(st));
      const size_t nitems = (static_cast<size_t>(S.alphabet) >> lo_bits) + 1;  // dense symbols 0..sigma
      std::vector<uint64_t> freq(nitems);
      for (size_t i = 0; i < nitems; i++) freq[i] = h32[i];
      code = build_symbol_code(freq, bits, true);  // (falls back to a fixed width where no code of <= 12 bits exists)
      if (!code.uniform_bits) {
        code.lo_bits = lo_bits;
        code.avg_bits += lo_bits;
      }
      c->code_cache = code;
      c->code_cached = true;
      c->code_alphabet = static_cast<uint32_t>(S.alphabet);
      c->code_bits = bits;
      c->code_lo = lo_bits;
      c->code_uses = 0;
      if (!code.uniform_bits) {
        const size_t blob_bytes = 512 + 256 + kDecodeTableBytes;
        std::memset(c->h_code, 0, blob_bytes);
        std::memcpy(c->h_code, code.cw.data(), code.cw.size() * sizeof(uint16_t));
        std::memcpy(c->h_code + 512, code.len.data(), code.len.size());
        std::memcpy(c->h_code + 768, code.bmask.data(), kDecodeTableBytes);
        WP_HIP(hipMemcpyAsync(c->d_code, c->h_code, blob_bytes, hipMemcpyHostToDevice, st));
      }
    }
    dcode = DevCode{reinterpret_cast<const uint16_t *>(c->d_code), c->d_code + 512, c->d_code + 768,
                    code.uniform_bits ? code.uniform_bits : -code.lo_bits};
    S.symbols_per_key = static_cast<int32_t>(kKeyBits / std::max(1.0, code.avg_bits));
    // 8-bit symbols with no codeword shorter than kKeys8MinLen bits (every ordinary text): the register form
    int min_len = code.uniform_bits ? code.uniform_bits : 99;
    for (uint8_t l : code.len) min_len = std::min<int>(min_len, l);
    if (sizeof(SymT) == 1 && min_len >= kKeys8MinLen) {
      if (n > kRadixSmallN) {  // its tiles are the tiles of the round-0 sort: the first digit's histogram comes along
        sort_plan = radix_plan<Key0>(n, d_radix_tmp, radix_words, st);
        hist_in_keys = true;
      }
      hipLaunchKernelGGL(build_keys0_u8_kernel, dim3(cdiv(n, kKeys8Tile)), dim3(kKeys8Threads), 0, st,
                         reinterpret_cast<const uint8_t *>(d_sym), n, dcode, KA, hist_in_keys ? nullptr : DG0,
                         hist_in_keys ? sort_plan.table : nullptr, hist_in_keys ? sort_plan.chunk_sums0 : nullptr);
    } else {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(build_keys0_kernel<SymT>), dim3(cdiv(n, kKeyTile)), dim3(kBlock), 0, st, d_sym, n, dcode,
                         KA, DG0);
    }
    WP_LAUNCH_CHECK();
  }

  // ---- round 0: 4 LSD passes over (key, suffix start) records (linear.cpp:121-141) ----------------------------------
  void sort_round0() {
    // (the low bits of a round-0 key are the tail of a compressed codeword stream: near-uniform digits)
    db.dg0 = DG0;
    db.dg1 = DG1;
    db.dg0_ready = DG0 != nullptr && !hist_in_keys;
    if (window_store) {  // the last pass leaves the first digit of the rank store's destination partition
      db.tail_bit = kWinBits;
      db.tail_mask = (1u << (win_mid - kWinBits)) - 1u;
      db.tail_from_val = true;
    }
    // histogram: one per-wave LDS counter per digit for the lowest digit (near-uniform), interleaved copies above it
    cur = radix_sort_pairs<Key0>(KA, VA, KB, VB, n, 0, kKeyBits, d_radix_tmp, radix_words, st, &c->rstats, true,
                                 code.uniform_bits ? 0 : 8, db, true, hist_in_keys ? &sort_plan : nullptr);
    S.hist_in_keys = hist_in_keys ? 1 : 0;
    keys = cur ? KB : KA;
    other_keys = cur ? KA : KB;
    vals = cur ? VB : VA;
    other_vals = cur ? VA : VB;
    slots = AS0;
    other_slots = AS1;
    adep = AD0;
    other_dep = AD1;
    avals = LA;
    spare_vals = LB;
  }

  // after every rerank: classify the new groups (large ones take the global path next round)
  // (runs on the side stream, next to the rank scatter)
  bool classify_groups(size_t list_len) {
    if (list_len <= static_cast<size_t>(kLsMaxGroup)) return false;  // no group can be large
    const size_t cap = list_len / 2 + 1;                              // a group has >= 2 entries
    WP_HIP(hipMemsetAsync(c->d_scalars + 6, 0, 2 * sizeof(uint32_t), st2));
    hipLaunchKernelGGL(large_groups_kernel, dim3(std::min<size_t>(cdiv(cap, kBlock), 2048)), dim3(kBlock), 0, st2, d_ghead,
                       c->d_scalars + 5, reinterpret_cast<unsigned long long *>(c->d_scalars + 6), d_lg_head, d_lg_off);
    hipLaunchKernelGGL(large_groups_close_kernel, dim3(1), dim3(1), 0, st2, c->d_scalars + 6, d_lg_off);
    return true;
  }

  // rank[dst[k]] = val[k] (val == nullptr: k) for a list of m entries.  Random 4-byte stores leave the L2s as partial
  // lines; from 4 M entries on, one radix pass over the top 8 bits of the destination first, and an XCD-aware scatter
  // after it, lets the stores of a workgroup fall into one ~1/256 window of the rank table.  t_dst / t_val: scratch.
  void store_ranks(uint32_t *dst, uint32_t *val, uint32_t *t_dst, uint32_t *t_val, size_t m) {
    if (val && m >= (1u << 22)) {
      const int hb = bit_length(n - 1);
      // (the top bits of a text position are uniformly distributed: histogram by LDS atomics)
      const int bc = radix_sort_pairs<uint32_t>(dst, val, t_dst, t_val, m, std::max(0, hb - 8), hb, d_radix_tmp, radix_words, st,
                                                nullptr, false, hb + 1, DigitBytes(), true);
      hipLaunchKernelGGL(scatter_pairs_kernel, dim3(cdiv(m, kSpTile)), dim3(kBlock), 0, st, bc ? t_dst : dst, bc ? t_val : val, m,
                         d_rank, n, 1);
    } else {
      hipLaunchKernelGGL(scatter_pairs_kernel, dim3(cdiv(m, kSpTile)), dim3(kBlock), 0, st, dst, val, m, d_rank, n, 0);
    }
  }

  // The rank store of round 0 at full size.  val == nullptr: the values are the slots themselves, made up by the first
  // pass.  The passes go through the scratch pairs a = (X0, X1) and b = (the value buffer the sort did not end in, the
  // sorted keys — free once the side stream's searches in them are over: before_second).
  // dig: digit bytes of dst bits [kWinBits, ...), left by the last pass of the sort; other: the second byte buffer.
  void store_ranks_round0(uint32_t *dst, uint32_t *val, uint8_t *dig, uint8_t *other, const std::function<void()> &before_second,
                          const std::function<void()> &before_window) {
    // (a per-device attribute: set on every call, the context may live on any device)
    WP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(window_store_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(kWinLdsBytes)));
    uint32_t *a_dst = X0, *a_val = X1, *b_dst = other_vals, *b_val = reinterpret_cast<uint32_t *>(keys);
    DigitBytes d1;
    d1.dg0 = dig;
    d1.dg1 = other;
    d1.dg0_ready = dig != nullptr;
    if (win_mid < hb_n) {  // (the first pass leaves the second pass's digits)
      d1.tail_bit = win_mid;
      d1.tail_mask = (1u << (hb_n - win_mid)) - 1u;
    }
    // (the digits of both passes are position bits, uniform over 64..128 values: LDS atomics into interleaved copies)
    radix_sort_pairs<uint32_t>(dst, val, a_dst, a_val, n, kWinBits, win_mid, d_radix_tmp, radix_words, st, &c->rstats,
                               val == nullptr, 0, d1, true);
    if (before_second) before_second();
    const uint32_t *f_dst = a_dst, *f_val = a_val;
    if (win_mid < hb_n) {
      DigitBytes d2;
      d2.dg0 = dig ? d1.tail_out(1) : nullptr;
      d2.dg1 = dig ? dig : nullptr;
      d2.dg0_ready = dig != nullptr;
      radix_sort_pairs<uint32_t>(a_dst, a_val, b_dst, b_val, n, win_mid, hb_n, d_radix_tmp, radix_words, st, &c->rstats, false, 0,
                                 d2);
      f_dst = b_dst;
      f_val = b_val;
    }
    if (before_window) before_window();
    hipLaunchKernelGGL(window_store_kernel, dim3(cdiv(n, size_t(1) << kWinBits)), dim3(kWinThreads), kWinLdsBytes, st, f_dst, f_val,
                       n, d_rank);
  }

  // ---- behind the sort: which tied groups go on (side stream), and the inverse suffix array (linear.cpp:144-147) ----
  void ranks_round0() {
    const DepthRule rule{need_depth, full ? 1 : 0, nullptr, nullptr, 0, nullptr};
    const unsigned tiles = cdiv(n, kRrTile);
    // Default layout: nobody asks for a group's head or depth — the step functions change at boundaries between
    // distinct keys only (short tokens) or inside needed groups (long tokens, resolved by the trie round, which stores
    // every rank it touches) — so a suffix's own slot serves as its rank and no kernel derives group heads.
    const bool slot_ranks = text_only && !d_lcp;
    uint8_t *dig = DG0 ? db.tail_out(cur) : nullptr, *dig_other = DG0 ? db.tail_out(cur ^ 1) : nullptr;
    // Depth-capped mode: the groups that have to go on are found from the vocabulary (prune.h) and appended to the
    // active list by the kernel that finds them — on the side stream (a few thousand waves of searches).
    // start_after != nullptr: the side stream starts there (an event of the main stream).
    auto enqueue_needed_groups = [&](hipEvent_t start_after) {
      if (start_after) WP_HIP(hipStreamWaitEvent(st2, start_after, 0));
      WP_HIP(hipMemsetAsync(d_claim, 0xff, claim_size * sizeof(uint32_t), st2));
      WP_HIP(hipMemsetAsync(d_claim_need, 0, claim_size * sizeof(uint32_t), st2));
      NeededList nl{slots, avals, AG, adep, d_ghead, d_gfirst, d_gdep, reinterpret_cast<unsigned long long *>(c->d_scalars + 4),
                    static_cast<uint32_t *>(nullptr), need_depth, d_claim_need, d_gclaim, d_gneed0};
      const TokenTrie trie{c->d_lt_chain_len, c->d_lt_chain_off, c->d_lt_child_begin, c->d_lt_child_node, d_child_sym};
      if (use_trie) {  // the vocabulary stream and the trie's child labels as dense symbols of this encode's alphabet
        const size_t ns = hv.stream.size(), nc = hv.lt_child_cp.size();
        hipLaunchKernelGGL(HIP_KERNEL_NAME(trie_map_symbols_kernel<SymT>), dim3(cdiv(std::max<size_t>(std::max(ns, nc), 1), kBlock)),
                           dim3(kBlock), 0, st2, c->d_stream, ns, c->d_lt_child_cp, nc, c->d_lut, d_vsym, d_child_sym);
      }
      if (M > 0) {  // (no eligible token at all: every tied group retires)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(need_groups_kernel<SymT>), dim3(cdiv(static_cast<size_t>(M) * kWave, kBlock)),
                           dim3(kBlock), 0, st2, keys, vals, n, d_sym, c->d_stream, c->d_elig_start, c->d_elig_info, M, c->d_lut, dcode,
                           d_claim, static_cast<uint32_t>(claim_size - 1), nl, text_only ? d_rng_lo : nullptr, d_rng_hi, d_rng_long);
      }
      hipLaunchKernelGGL(needed_list_clamp_kernel, dim3(1), dim3(1), 0, st2, c->d_scalars + 4, static_cast<uint32_t>(list_cap),
                         c->d_scalars + 8);
      hipLaunchKernelGGL(needed_list_close_kernel, dim3(1), dim3(1), 0, st2, c->d_scalars + 4, d_ghead);
      if (M > 0) {
        if (!use_trie) hipLaunchKernelGGL(needed_need_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st2, nl);
        hipLaunchKernelGGL(needed_fill_kernel, dim3(1024), dim3(kBlock), 0, st2, nl, vals, n);
        if (use_trie) {
          hipLaunchKernelGGL(HIP_KERNEL_NAME(trie_group_start_kernel<SymT>), dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st2, vals, d_gfirst,
                             d_gdep, c->d_scalars + 4, d_sym, n, d_vsym, trie, d_gnode, d_gdone);
        }
      }
      // (from here on the side stream no longer reads the sorted keys and suffixes: the rank store may reuse them)
      WP_HIP(hipEventRecord(c->evs[3], st2));
      if (use_trie) trie_round_sizes();
    };
    // (they start when the sort ends, beside the first partition pass's histogram: started beside its scatter instead,
    // which is bandwidth-bound, they cost the scatter more than they cost the histogram now — measured)
    if (prune) {
      fork();
      enqueue_needed_groups(nullptr);
      if (!slot_ranks) {  // reference layout: rank entries, LCPs and group depths from the sorted keys
        hipLaunchKernelGGL(round0_rank_kernel<true>, dim3(cdiv(n, kR0Tile)), dim3(kBlock), 0, st, keys, vals, n, dcode.first_len,
                           dcode.uniform_bits, d_sa, d_hd_n, d_lcp, d_gdepth);
      }
    } else {
      // every tied group goes on (true suffix array, or no pruning possible): the count / prefix / apply kernels take
      // 64-bit keys, the 32-bit round-0 keys are widened into the large-list key buffer (free during round 0)
      hipLaunchKernelGGL(widen_keys_kernel, dim3(std::min<size_t>(cdiv(n, kBlock), 8192)), dim3(kBlock), 0, st, keys, LK2, n);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_agg_kernel<true>), dim3(tiles), dim3(kBlock), 0, st, LK2, vals, n,
                         static_cast<const uint32_t *>(nullptr), static_cast<const RankEntry *>(nullptr),
                         static_cast<const uint32_t *>(nullptr), n, dcode.first_len, dcode.uniform_bits, rule, d_tdep, d_agg);
      hipLaunchKernelGGL(rerank_chunk_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, tiles, d_chunk_agg);
      hipLaunchKernelGGL(rerank_prefix_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, d_chunk_agg, tiles);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_apply_kernel<SymT, true>), dim3(tiles), dim3(kBlock), 0, st, LK2, vals,
                         static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr), d_tdep, n, d_agg, d_sym, n,
                         dcode.first_len, dcode.uniform_bits, rule, d_sa, d_hd_n, d_lcp, slots, avals, AG, adep, d_ghead, d_gdepth,
                         c->d_scalars + 4);
    }
    uint32_t *rank_vals = slot_ranks ? nullptr : reinterpret_cast<uint32_t *>(d_hd_n);
    auto keys_free = [&] {  // pair b of the rank store holds the sorted keys: the side stream's searches in them must be over
      if (prune) WP_HIP(hipStreamWaitEvent(st, c->evs[3], 0));
    };
    // The trie round (side stream) starts when the partition passes are through: its small latency-bound kernels run
    // beside the window store — beside the radix passes they took a quarter of the passes' bandwidth for the same gain.
    auto start_trie_round = [&] {
      if (!use_trie) return;
      WP_HIP(hipEventRecord(c->evs[4], st));
      WP_HIP(hipStreamWaitEvent(st2, c->evs[4], 0));
      trie_round_sort();
    };
    if (window_store) {
      store_ranks_round0(vals, rank_vals, dig, dig_other, keys_free, start_trie_round);
    } else {
      keys_free();
      start_trie_round();
      store_ranks(vals, rank_vals, X0, X1, n);
    }
    WP_LAUNCH_CHECK();
    if (use_trie) {
      trie_round_finish();
    } else {
      if (prune) join();
      fork();
      classified = classify_groups(std::min(n, list_cap));
      join();
    }
  }

  // ---- trie refinement of the needed groups (trie.h), on the side stream BESIDE the window store of the rank store ----
  // Nothing in it reads a rank: every entry of the needed list walks the token trie (its end node is its sort key),
  // the groups are sorted by node (LDS windows; large groups through a radix sort with its own temporary), one split
  // assigns the new slots.  Only the last step — the ranks of the entries that moved — has to wait for the rank table.
  // The list sizes travel to the host as soon as the list is built, so every launch is queued long before it can run.
  int trie_rb() const { return bit_length(hv.lt_chain_len.size() + 1); }  // second keys: 1 + trie node
  void trie_round_sizes() {
    classified = classify_groups(list_cap);  // (side stream: the table of large groups)
    WP_HIP(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(uint32_t) * 12, hipMemcpyDeviceToHost, st2));
    WP_HIP(hipEventRecord(c->evs[2], st2));
  }
  void trie_round_sort() {
    const TokenTrie trie{c->d_lt_chain_len, c->d_lt_chain_off, c->d_lt_child_begin, c->d_lt_child_node, d_child_sym};
    // (sizes are read on the device: the launches do not wait for the host to learn them)
    hipLaunchKernelGGL(HIP_KERNEL_NAME(trie_walk_kernel<SymT>), dim3(std::min<size_t>(cdiv(list_cap, kBlock), 16384)), dim3(kBlock), 0,
                       st2, avals, AG, d_gnode, d_gdone, c->d_scalars + 4, d_sym, n, d_vsym, trie, adep);
    WP_HIP(hipEventRecord(c->evs[5], st2));  // (the large-group path starts from here, beside the LDS sort: trie_round_finish)
    hipLaunchKernelGGL(local_sort_kernel, dim3(cdiv(list_cap, kLsT)), dim3(kBlock), 0, st2, avals, AG, adep, c->d_scalars + 4,
                       d_ghead, d_rank, n, trie_rb(), LK0, spare_vals, adep);
  }
  void trie_round_finish() {
    WP_HIP(hipEventSynchronize(c->evs[2]));
    n_act = c->h_scalars[4];
    n_large_groups = classified ? c->h_scalars[6] : 0;
    n_large = classified ? c->h_scalars[7] : 0;
    if (c->h_scalars[8] > list_cap) throw ListOverflow{c->h_scalars[8]};  // (the list was kept empty: nothing ran on it)
    S.active_per_round[0] = static_cast<int64_t>(n);
    uint64_t *skeys = LK0;
    RankEntry *hd = reinterpret_cast<RankEntry *>(LK1);
    if (n_act > 0) {
      S.active_per_round[1] = static_cast<int64_t>(n_act);
      const int rb = trie_rb();
      if (n_large > 0) {  // (second side stream: other list positions than the LDS sort's, scratch of its own)
        hipStream_t st3 = c->stream3;
        WP_HIP(hipStreamWaitEvent(st3, c->evs[5], 0));
        const int lgb = bit_length(n_large_groups > 0 ? n_large_groups - 1 : 0);
        hipLaunchKernelGGL(large_extract_kernel, dim3(cdiv(cdiv(n_large, kLxSpan), kBlock / kWave)), dim3(kBlock), 0, st3, avals, adep,
                           d_lg_head, d_lg_off, static_cast<uint32_t>(n_large_groups), n_large, d_rank, n, rb, LK1, LV0, LPOS, adep);
        const int lc = radix_sort_pairs<uint64_t>(LK1, LV0, LK2, LV1, n_large, 0, rb + lgb, d_radix_tmp2,
                                                  radix_tmp_words<uint64_t>(list_cap), st3, nullptr);
        hipLaunchKernelGGL(large_writeback_kernel, dim3(std::min<size_t>(cdiv(n_large, kBlock), 8192)), dim3(kBlock), 0, st3,
                           lc ? LK2 : LK1, lc ? LV1 : LV0, LPOS, n_large, AG, rb, skeys, spare_vals);
        WP_HIP(hipEventRecord(c->evs[6], st3));
        WP_HIP(hipStreamWaitEvent(st2, c->evs[6], 0));
      }
      DepthRule rrule{need_depth, 0, nullptr, nullptr, 1, d_node_of_slot};  // final round: every group retires
      const unsigned tiles = cdiv(n_act, kRrTile);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_agg_kernel<false>), dim3(tiles), dim3(kBlock), 0, st2, skeys, spare_vals, n_act, adep,
                         d_rank, d_gdepth, n, dcode.first_len, dcode.uniform_bits, rrule, d_tdep, d_agg);
      hipLaunchKernelGGL(rerank_chunk_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st2, d_agg, tiles, d_chunk_agg);
      hipLaunchKernelGGL(rerank_prefix_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st2, d_agg, d_chunk_agg, tiles);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_apply_kernel<SymT, false>), dim3(tiles), dim3(kBlock), 0, st2, skeys, spare_vals, slots,
                         adep, d_tdep, n_act, d_agg, d_sym, n, dcode.first_len, dcode.uniform_bits, rrule, d_sa, hd, d_lcp,
                         other_slots, avals, AG, other_dep, d_ghead, d_gdepth, c->d_scalars + 4);
    }
    join();  // the rank table is complete (main stream) and the new ranks are known (side stream)
    if (n_act > 0) {
      store_ranks(spare_vals, hd, reinterpret_cast<uint32_t *>(hd) + list_cap + 2, reinterpret_cast<uint32_t *>(skeys), n_act);
      rounds = 2;
    }
    WP_LAUNCH_CHECK();
  }

  // Between two rounds the host needs the new list sizes (grids, large-group path).  The copy of the scalars and the
  // LDS segmented sort of the next round are queued first — the sort reads its sizes on the device and gets a grid for
  // the largest possible list — and only then does the host wait for the copy: the round trip hides behind the sort.
  void next_round_begin(size_t upper, int rb) {
    WP_HIP(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(uint32_t) * 12, hipMemcpyDeviceToHost, st));
    WP_HIP(hipEventRecord(c->evs[2], st));
    fork();  // the large-group path of the next round (side stream) may start from here
    if (upper > 0) {
      hipLaunchKernelGGL(local_sort_kernel, dim3(cdiv(upper, kLsT)), dim3(kBlock), 0, st, avals, AG, adep, c->d_scalars + 4, d_ghead,
                         d_rank, n, rb, LK0, spare_vals, static_cast<const uint32_t *>(nullptr));
    }
    WP_HIP(hipEventSynchronize(c->evs[2]));
    n_act = c->h_scalars[4];
    n_large_groups = classified ? c->h_scalars[6] : 0;
    n_large = classified ? c->h_scalars[7] : 0;
    if (c->h_scalars[8] > list_cap) throw ListOverflow{c->h_scalars[8]};  // (the list was kept empty: nothing ran on it)
  }

  // ---- rounds >= 1 over the active list --------------------------------------------------------------------------
  // Default layout: ONE split — every entry's second key is the trie node its suffix ends in (trie.h), all groups
  // retire — run beside the rank store (trie_round_begin / _finish).  Reference layout / full depth: prefix
  // doubling, a group sorted by rank[i + depth(group)] per round (doubling_rounds).
  void refine() {
    if (!use_trie) doubling_rounds();  // (trie refinement has run beside the rank store: trie_round_begin / _finish)
    S.rounds = rounds;
    // every tie that is left shares at least this many symbols: need_depth for the groups that went through the
    // rounds, the shortest possible key (whole codewords in kKeyBits bits) for the groups round 0 let go
    const int max_len = code.uniform_bits ? code.uniform_bits : kMaxCodeLen + code.lo_bits;
    const int32_t key_syms = std::max(1, kKeyBits / max_len);
    S.sorted_depth = full ? 0x7fffffff
                          : (prune ? std::min<int32_t>(static_cast<int32_t>(need_depth), key_syms) : static_cast<int32_t>(need_depth));
    S.needed_after_round0 = prune ? (rounds > 1 ? S.active_per_round[1] : 0) : -1;
    if (prune && rounds > 1) c->list_hint = static_cast<size_t>(S.active_per_round[1]);
  }

  void doubling_rounds() {
    const int rb = bit_length(n);  // second keys of a round: 1 + rank (<= n)
    const DepthRule rule{need_depth, full ? 1 : 0, nullptr, nullptr, 0, nullptr};
    next_round_begin(std::min(n, list_cap), rb);
    S.active_per_round[0] = static_cast<int64_t>(n);
    // behind a pruned round 0 every group carries the depth its own tokens need (DepthRule, prune.h)
    uint32_t *gneed_cur = d_gneed0, *gneed_nxt = d_gneed1;
    const bool group_need = prune && M > 0;
    while (n_act > 0) {
      DepthRule rrule = rule;
      if (group_need) {
        rrule.gneed_in = gneed_cur;
        rrule.gneed_out = gneed_nxt;
        std::swap(gneed_cur, gneed_nxt);
      }
      // A doubling round adds to a group's depth the depth of the group its second keys point into: that doubles the
      // depth while those groups are refined too (full depth: 31 rounds for 2^31 symbols), and adds at least the depth
      // of a round-0 group — one symbol or more — when they retired in round 0.  More rounds than that can only mean
      // corrupted ranks: stop instead of spinning.
      if (static_cast<uint64_t>(rounds) > 80 + (full ? 0ull : static_cast<uint64_t>(need_depth))) {
        throw HipError("prefix doubling did not converge (internal error)");
      }
      if (rounds < 40) S.active_per_round[rounds] = static_cast<int64_t>(n_act);
      // small groups: one LDS-resident segmented sort per window of the list (already queued by next_round_begin:
      // avals -> (LK0, spare_vals))
      uint64_t *skeys = LK0, *kfree = LK1;
      uint32_t *svals = spare_vals, *nvals = avals;  // avals is free again once the sorts have consumed it
      if (n_large > 0) {  // large groups (side stream, disjoint list positions): extract, global radix sort on
                          // (dense large id, second key), write back
        const int lgb = bit_length(n_large_groups > 0 ? n_large_groups - 1 : 0);
        hipLaunchKernelGGL(large_extract_kernel, dim3(cdiv(cdiv(n_large, kLxSpan), kBlock / kWave)), dim3(kBlock), 0, st2, avals, adep,
                           d_lg_head, d_lg_off, static_cast<uint32_t>(n_large_groups), n_large, d_rank, n, rb, LK1, LV0, LPOS,
                           static_cast<const uint32_t *>(nullptr));
        const int lc = radix_sort_pairs<uint64_t>(LK1, LV0, LK2, LV1, n_large, 0, rb + lgb, d_radix_tmp, radix_words, st2, nullptr);
        hipLaunchKernelGGL(large_writeback_kernel, dim3(std::min<size_t>(cdiv(n_large, kBlock), 8192)), dim3(kBlock), 0, st2,
                           lc ? LK2 : LK1, lc ? LV1 : LV0, LPOS, n_large, AG, rb, skeys, svals);
        join();
      }
      const unsigned tiles = cdiv(n_act, kRrTile);
      RankEntry *hd = reinterpret_cast<RankEntry *>(kfree);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_agg_kernel<false>), dim3(tiles), dim3(kBlock), 0, st, skeys, svals, n_act, adep,
                         d_rank, d_gdepth, n, dcode.first_len, dcode.uniform_bits, rrule, d_tdep, d_agg);
      hipLaunchKernelGGL(rerank_chunk_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, tiles, d_chunk_agg);
      hipLaunchKernelGGL(rerank_prefix_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, d_chunk_agg, tiles);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_apply_kernel<SymT, false>), dim3(tiles), dim3(kBlock), 0, st, skeys, svals, slots,
                         adep, d_tdep, n_act, d_agg, d_sym, n, dcode.first_len, dcode.uniform_bits, rrule, d_sa, hd, d_lcp,
                         other_slots, nvals, AG, other_dep, d_ghead, d_gdepth, c->d_scalars + 4);
      fork();
      // (scratch of the partitioned store: behind the rank entries in their buffer, and the sorted keys)
      store_ranks(svals, hd, reinterpret_cast<uint32_t *>(hd) + list_cap + 2, reinterpret_cast<uint32_t *>(skeys), n_act);
      WP_LAUNCH_CHECK();
      rounds++;
      classified = classify_groups(n_act);
      join();
      std::swap(slots, other_slots);
      std::swap(adep, other_dep);
      avals = nvals;
      spare_vals = svals;
      next_round_begin(n_act, rb);  // (the next list is at most as long as this one)
    }
  }

  // ---- who marks + the four scanlines as step functions (linear.cpp:153-213) -----------------------------------------
  void scanlines() {
    // (the side stream may start now, but its launches are issued behind the first scanline kernels so that the host
    // does not keep the main stream waiting)
    if (n_text > 0) fork();
    MarkView mv{};
    // (step values carry the token length above the id where both fit: scanline.h)
    const int pack_steps = (hv.longest < kStepMaxLen && hv.tokens.size() < (size_t(1) << kStepIdBits)) ? 1 : 0;
    const size_t vocab_base = n_text + 1;
    uint32_t *mslot = d_mslot0, *midx = d_midx0;
    if (text_only) {
      // S = text . 1: the reach of every token is its range in the sorted keys (prune.h); a long token's is the run of
      // the trie nodes below its own inside its group (trie.h).  The marks arrive sorted (tokens in lexicographic order).
      if (M > 0) {
        hipLaunchKernelGGL(trie_token_range_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, d_node_of_slot, c->d_elig_node,
                           c->d_elig_subtree, M, d_rng_lo, d_rng_hi, d_rng_long);
        hipLaunchKernelGGL(virtual_marks_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, d_rng_lo, d_rng_hi, M, c->d_elig_id,
                           c->d_elig_info, d_mslot0, d_mid, d_minfo, d_rf, d_rb);
      }
      mv = MarkView{mslot, d_mid, d_minfo, d_rf, d_rb, M, d_cover_f, d_cover_b};
    } else {
      if (M > 0) {
        hipLaunchKernelGGL(mark_slots_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, c->d_elig_start, M, vocab_base, d_rank,
                           d_mslot0, d_midx0);
        int mc = radix_sort_pairs<uint32_t>(d_mslot0, d_midx0, d_mslot1, d_midx1, M, 0, bit_length(n), d_radix_tmp, radix_words, st,
                                            nullptr);
        mslot = mc ? d_mslot1 : d_mslot0;
        midx = mc ? d_midx1 : d_midx0;
        hipLaunchKernelGGL(mark_gather_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, midx, M, c->d_elig_id, c->d_elig_info,
                           d_mid, d_minfo);
      }
      hipLaunchKernelGGL(tile_mlo_kernel, dim3(cdiv(sl_tiles + 1, kBlock)), dim3(kBlock), 0, st, mslot, M, n, sl_tiles, d_tile_mlo);
      hipLaunchKernelGGL(sl_summary_kernel, dim3(sl_tiles), dim3(kBlock), 0, st, d_lcp, n, mslot, d_minfo, d_tile_mlo, d_tmin_f,
                         d_tmin_b, d_rf, d_rb);
      hipLaunchKernelGGL(sl_group_min_kernel, dim3(cdiv(static_cast<size_t>(sl_groups) * kWave, kBlock)), dim3(kBlock), 0, st,
                         d_tmin_f, d_tmin_b, sl_tiles, sl_groups, d_gmin_f, d_gmin_b);
      mv = MarkView{mslot, d_mid, d_minfo, d_rf, d_rb, M, d_cover_f, d_cover_b};
      if (M > 0) {
        hipLaunchKernelGGL(sl_reach_global_kernel, dim3(cdiv(static_cast<size_t>(M) * kWave, kBlock)), dim3(kBlock), 0, st, d_lcp, n,
                           sl_tiles, sl_groups, mslot, d_minfo, M, d_tmin_f, d_tmin_b, d_gmin_f, d_gmin_b, d_rf, d_rb);
      }
    }
    if (M > 0) {
      hipLaunchKernelGGL(mark_cover_kernel, dim3(4), dim3(kCoverThreads), 0, st, d_minfo, d_rf, d_rb, M, d_cover_f, d_cover_b);
    }
    if (n_text > 0) launch_anchors();
    hipLaunchKernelGGL(piece_starts_kernel, dim3(cdiv(std::max(M, 1), kBlock)), dim3(kBlock), 0, st, mv, n, d_ps0);
    const int pc = radix_sort_pairs<uint32_t>(d_ps0, d_pv0, d_ps1, d_pv1, P, 0, bit_length(n), d_radix_tmp, radix_words, st, nullptr);
    uint32_t *pstart = pc ? d_ps1 : d_ps0;
    hipLaunchKernelGGL(piece_values_kernel, dim3(cdiv(static_cast<size_t>(P) * kWave, kBlock)), dim3(kBlock), 0, st, mv, pstart, P,
                       d_pval_p, d_pval_s, pack_steps);
    hipLaunchKernelGGL(piece_bucket_kernel, dim3(cdiv(nbuckets + 1, kBlock)), dim3(kBlock), 0, st, pstart, P, bucket_shift, nbuckets,
                       d_bidx);
    hipLaunchKernelGGL(piece_bucket_fast_kernel, dim3(cdiv(nbuckets, kBlock)), dim3(kBlock), 0, st, d_bidx, d_pval_p, d_pval_s, nbuckets,
                       d_bfast);
    WP_LAUNCH_CHECK();
    steps = StepTable{pstart, d_pval_p, d_pval_s, d_bidx, bucket_shift, pack_steps, d_bfast};
    steps_all = steps;
    if (bucket_shift_all != bucket_shift) {
      hipLaunchKernelGGL(piece_bucket_kernel, dim3(cdiv(nbuckets_all + 1, kBlock)), dim3(kBlock), 0, st, pstart, P, bucket_shift_all,
                         nbuckets_all, d_bidx_all);
      hipLaunchKernelGGL(piece_bucket_fast_kernel, dim3(cdiv(nbuckets_all, kBlock)), dim3(kBlock), 0, st, d_bidx_all, d_pval_p, d_pval_s,
                         nbuckets_all, d_bfast_all);
      WP_LAUNCH_CHECK();
      steps_all = StepTable{pstart, d_pval_p, d_pval_s, d_bidx_all, bucket_shift_all, pack_steps, d_bfast_all};
    }
  }

  // words longer than a lane should walk (walk.h, "long words"): pointer doubling instead.  Scratch: the slabs of the sort.
  void walk_long_words(const WalkArgs &wa, size_t n_anchors) {
    const uint32_t lw_cap = static_cast<uint32_t>(n_text / kMaxAnchorGap + 2);
    LongWord *d_lw = reinterpret_cast<LongWord *>(d_tile_scratch);
    uint32_t *d_lw_off = d_tile_scratch + 2 * static_cast<size_t>(lw_cap);
    uint32_t *d_lw_fail = d_lw_off + lw_cap + 1;
    hipLaunchKernelGGL(long_word_collect_kernel, dim3(std::min<size_t>(cdiv(std::max<size_t>(n_anchors, 1), kBlock), 2048)),
                       dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, n_text, d_cls, d_lw, lw_cap, c->d_scalars + 12);
    WP_LAUNCH_CHECK();
    fetch_scalars(c, 13);
    const uint32_t nw = std::min(c->h_scalars[12], lw_cap);
    if (nw == 0) return;
    std::vector<LongWord> h_lw(nw);
    WP_HIP(hipMemcpyAsync(h_lw.data(), d_lw, sizeof(LongWord) * nw, hipMemcpyDeviceToHost, st));
    WP_HIP(hipStreamSynchronize(st));
    std::vector<uint32_t> h_off(nw + 1);
    uint64_t total64 = 0;
    uint32_t longest = 0;
    for (uint32_t i = 0; i < nw; i++) {
      h_off[i] = static_cast<uint32_t>(total64);
      total64 += h_lw[i].end - h_lw[i].begin;
      longest = std::max(longest, h_lw[i].end - h_lw[i].begin);
    }
    h_off[nw] = static_cast<uint32_t>(total64);
    const uint32_t total = static_cast<uint32_t>(total64);  // <= n_text < 2^31
    WP_HIP(hipMemcpyAsync(d_lw_off, h_off.data(), sizeof(uint32_t) * (nw + 1), hipMemcpyHostToDevice, st));
    WP_HIP(hipMemsetAsync(d_lw_fail, 0, sizeof(uint32_t) * nw, st));
    int32_t *d_lid = reinterpret_cast<int32_t *>(X0);
    uint32_t *jump_a = X1, *jump_b = reinterpret_cast<uint32_t *>(KA);
    uint8_t *d_mark = reinterpret_cast<uint8_t *>(KB);
    const dim3 grid(cdiv(total, kBlock));
    WalkArgs wa_all = wa;  // (every position of the long words is looked up: the small index)
    wa_all.steps = steps_all;
    hipLaunchKernelGGL(long_word_next_kernel, grid, dim3(kBlock), 0, st, wa_all, d_lw, d_lw_off, nw, total, d_lid, jump_a, d_mark);
    WP_HIP(hipStreamSynchronize(st));  // h_off is a stack-owned upload source
    for (uint32_t reach = 1; reach < longest; reach *= 2) {  // after r rounds: chain prefixes of length 2^r
      hipLaunchKernelGGL(long_word_mark_kernel, grid, dim3(kBlock), 0, st, jump_a, total, d_mark);
      hipLaunchKernelGGL(long_word_double_kernel, grid, dim3(kBlock), 0, st, jump_a, total, jump_b);
      std::swap(jump_a, jump_b);
    }
    hipLaunchKernelGGL(long_word_mark_kernel, grid, dim3(kBlock), 0, st, jump_a, total, d_mark);
    hipLaunchKernelGGL(long_word_fail_kernel, grid, dim3(kBlock), 0, st, d_lid, d_mark, d_lw_off, nw, total, d_lw_fail);
    hipLaunchKernelGGL(long_word_emit_kernel, grid, dim3(kBlock), 0, st, wa, d_lw, d_lw_off, nw, total, d_lid, d_mark, d_lw_fail);
    WP_LAUNCH_CHECK();
    S.anchor_mode = 2;
  }

  // long stretches without class-rule anchors ("soft" spacing chars): anchors from the matches themselves, inside the
  // long gaps of the class rule only (walk.h).  Returns the number of anchors.
  size_t cover_anchors(WalkArgs &wa) {
    uint32_t *d_reach = X0, *d_reach_tiles = d_tile_scratch;
    uint8_t *d_aflags = reinterpret_cast<uint8_t *>(X1);
    const unsigned rtiles = cdiv(n_text, kReachTile), atiles = cdiv(n_text, kAnchorTile);
    uint32_t *d_wp_tiles = d_reach_tiles + rtiles + 1;  // first word-prefix position at or behind each tile
    uint32_t *d_ns_tiles = d_wp_tiles + rtiles + 1;     // same for non-space positions
    uint32_t *d_gap_a = d_ns_tiles + rtiles + 1, *d_gap_b = d_gap_a + rtiles + 1;  // where the coverage rule applies
    // (the class-rule anchor list is still in d_anchors: the coverage rule is only needed inside its long gaps)
    hipLaunchKernelGGL(gap_tiles_kernel, dim3(cdiv(rtiles, kBlock)), dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, n_text, rtiles,
                       v->cover_anchors ? 1 : 0, d_gap_a, d_gap_b);
    hipLaunchKernelGGL(reach_kernel, dim3(rtiles), dim3(kBlock), 0, st, wa, d_reach, d_reach_tiles, d_gap_a, d_gap_b);
    hipLaunchKernelGGL(reach_spine_kernel, dim3(1), dim3(1024), 0, st, d_reach_tiles, static_cast<size_t>(rtiles));
    WP_HIP(hipMemsetAsync(d_anchor_cnt, 0, sizeof(uint32_t) * atiles, st));
    hipLaunchKernelGGL(cover_flags_kernel, dim3(rtiles), dim3(kBlock), 0, st, d_cls, d_reach, d_reach_tiles, n_text, d_aflags,
                       d_wp_tiles, d_ns_tiles, d_gap_a, d_gap_b, d_anchor_cnt);
    hipLaunchKernelGGL(suffix_min_kernel, dim3(2), dim3(1024), 0, st, d_wp_tiles, d_ns_tiles, static_cast<size_t>(rtiles));
    device_exclusive_scan(d_anchor_cnt, d_anchor_cnt, atiles, d_anchor_tmp, c->d_scalars + 10, st);  // (counted by cover_flags_kernel)
    hipLaunchKernelGGL(anchor_write_kernel, dim3(atiles), dim3(kBlock), 0, st, d_cls, d_aflags, n_text, d_anchor_cnt, d_anchors);
    WP_LAUNCH_CHECK();
    fetch_scalars(c, 11);
    wa.aflags = d_aflags;
    wa.wp_from_tile = d_wp_tiles;
    wa.ns_from_tile = d_ns_tiles;
    S.anchor_mode = 1;
    return c->h_scalars[10];
  }

  // ---- greedy walk + id stream (linear.cpp:215-316).  Slabs: ids VB, id lists KA, wide list / counts KB VA (X0 / X1
  // hold the coverage rule's reach / flags or the long words' scratch) --------------------------------------------------
  int32_t *walk(size_t *n_ids_out) {
    int32_t *d_ids = reinterpret_cast<int32_t *>(VB);
    *n_ids_out = 0;
    if (n_text == 0) return d_ids;
    WalkArgs wa{d_cls, n_text, d_rank, steps, c->d_tok_len, hv.unk_id, d_emit, nullptr, nullptr, nullptr,
                hv.soft.empty() ? 1 : 0, static_cast<int32_t>(hv.tokens.size())};
    S.anchor_mode = 0;
    join();  // (the anchor list of the side stream)
    fetch_scalars(c, 12);
    size_t n_anchors = c->h_scalars[10];
    const size_t max_anchor_gap = c->h_scalars[11];
    const bool all_hard = hv.soft.empty();
    bool staged = staged_possible && max_anchor_gap <= kMaxAnchorGap;
    if (staged_possible && !staged) WP_HIP(hipMemsetAsync(d_emit, 0x80, n_text * sizeof(int32_t), st));  // long words after all
    if (!v->cover_anchors && all_hard && max_anchor_gap > kMaxAnchorGap) {
      walk_long_words(wa, n_anchors);
    } else if (v->cover_anchors || max_anchor_gap > kMaxAnchorGap) {
      n_anchors = cover_anchors(wa);
      // coverage anchors: every id still comes from the lanes of the walk kernel, each inside its own stretch
      // [anchor, next anchor) — the id lists work as they do for the class rule (the cleared emit array is not used)
      staged = !sparse_emit;
    }
    S.n_anchors = static_cast<int64_t>(n_anchors);
    // one lane per anchor (a grid sized for the worst case, every position an anchor, costs 0.35 ms of empty workgroups)
    const size_t acap = std::max<size_t>(n_anchors, 1);
    if (staged) {
      int32_t *d_ctmp = reinterpret_cast<int32_t *>(KA);
      const unsigned sblocks = cdiv(acap, static_cast<size_t>(kWbWords));
      // stretches of more than kWideMin positions (class rule, hard spacing chars only: one word each) go to a whole
      // wave each first (walk.h, wide walk)
      if (S.anchor_mode == 0 && all_hard && max_anchor_gap > kWideMin) {
        uint32_t *d_wide_list = reinterpret_cast<uint32_t *>(KB), *d_wide_cnt = VA;
        WP_HIP(hipMemsetAsync(c->d_scalars + 13, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(wide_collect_kernel, dim3(std::min<size_t>(cdiv(acap, kBlock), 2048)), dim3(kBlock), 0, st, d_anchors,
                           c->d_scalars + 10, n_text, d_wide_list, c->d_scalars + 13);
        WalkArgs wa_all = wa;  // (every position of the wide words is looked up: the small index)
        wa_all.steps = steps_all;
        hipLaunchKernelGGL(walk_wide_kernel, dim3(std::min<size_t>(cdiv(acap, kBlock / kWave), 8192)), dim3(kBlock), 0, st, wa_all, d_anchors,
                           c->d_scalars + 10, d_wide_list, c->d_scalars + 13, d_wide_cnt);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(walk_lean_kernel<true>), dim3(sblocks), dim3(kBlock), 0, st, wa, d_anchors,
                           c->d_scalars + 10, acap, d_ctmp, d_blk_cnt, d_wide_cnt);
      } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(walk_lean_kernel<false>), dim3(sblocks), dim3(kBlock), 0, st, wa, d_anchors,
                           c->d_scalars + 10, acap, d_ctmp, d_blk_cnt, static_cast<const uint32_t *>(nullptr));
      }
      device_exclusive_scan(d_blk_cnt, d_blk_off, sblocks, d_emit_tmp, c->d_scalars + 9, st);
      hipLaunchKernelGGL(emit_gather_kernel, dim3(sblocks), dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, acap, d_ctmp, d_blk_cnt,
                         d_blk_off, d_ids, kWbWords);
    } else {
      hipLaunchKernelGGL(walk_kernel, dim3(cdiv(acap, kBlock)), dim3(kBlock), 0, st, wa, d_anchors, c->d_scalars + 10, acap);
      const unsigned tiles = cdiv(n_text, kScanTile);
      hipLaunchKernelGGL(emit_count_kernel, dim3(tiles), dim3(kBlock), 0, st, d_emit, n_text, d_emit_cnt);
      device_exclusive_scan(d_emit_cnt, d_emit_cnt, tiles, d_emit_tmp, c->d_scalars + 9, st);
      hipLaunchKernelGGL(emit_write_kernel, dim3(tiles), dim3(kBlock), 0, st, d_emit, n_text, d_emit_cnt, d_ids);
    }
    S.staged_emit = staged ? 1 : 0;
    WP_LAUNCH_CHECK();
    return d_ids;
  }

  // guard zones, bounds counters, the id count, statistics, debug views
  void finish(int32_t *d_ids, size_t *n_ids_out) {
    if (ar.guard) {  // debugging aid: no kernel may have written outside the buffer it was given
      static const uint32_t init[2] = {0u, 0xffffffffu};
      WP_HIP(hipMemcpyAsync(c->d_scalars + 16, init, sizeof(init), hipMemcpyHostToDevice, st));
      ar.check(st, c->d_scalars + 16);
      aa.check(st, c->d_scalars + 16);
      fetch_scalars(c, 18);
      if (c->h_scalars[16] != 0) {
        throw HipError("arena guard: " + std::to_string(c->h_scalars[16]) + " guard zone(s) overwritten, first behind allocation #" +
                       std::to_string(c->h_scalars[17] - 1));
      }
      S.guard_zones = static_cast<int32_t>(ar.zones.size() + aa.zones.size());
    }
#ifdef WP_DEBUG_BOUNDS
    {
      unsigned int oob[kBoundSites] = {};
      WP_HIP(hipStreamSynchronize(st));
      WP_HIP(hipMemcpyFromSymbol(oob, HIP_SYMBOL(g_wp_oob), sizeof(oob)));
      const unsigned int zero[kBoundSites] = {};
      WP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_wp_oob), zero, sizeof(zero)));
      if (oob[0] | oob[1] | oob[2] | oob[3]) {
        throw HipError("debug bounds: out-of-range addresses skipped: radix scatter " + std::to_string(oob[0]) + ", rank store " +
                       std::to_string(oob[1]) + ", token id " + std::to_string(oob[2]) + ", list slot " + std::to_string(oob[3]));
      }
      S.reserved0 = 1;  // this is the bounds-checking build
    }
#endif
    fetch_scalars(c, 20);
    const size_t n_ids = n_text > 0 ? c->h_scalars[9] : 0;
    S.n_ids = static_cast<int64_t>(n_ids);
    S.radix_passes = c->rstats.passes;
    S.radix_pass_elems = c->rstats.elems;
    S.radix_digit_bytes = c->rstats.digit_bytes;
    S.radix_pass_bytes = c->rstats.bytes;
    S.arena_bytes = static_cast<int64_t>(c->a_buf.cap + c->b_buf.cap);
    if (v->stage_timing) {
      auto span = [&](int a, int b) {
        float ms = 0;
        WP_HIP(hipEventElapsedTime(&ms, c->ev[a], c->ev[b]));
        return static_cast<double>(ms);
      };
      S.ms_decode = span(0, 2);
      S.ms_sa = span(2, 3);
      S.ms_lcp = span(3, 4);
      S.ms_scan = span(4, 5);
      S.ms_walk = span(5, 6);
      S.ms_total = span(0, 6);
      S.ms_radix_scatter = c->rstats.spans.resolve();
    }
    c->d_ids = d_ids;
    c->dbg.sym = d_sym;
    c->dbg.sym_bytes = sizeof(SymT);
    c->dbg.sa = d_sa;
    c->dbg.rank = d_rank;
    c->dbg.lcp = d_lcp;
    c->dbg.steps = steps;
    c->dbg.best_scratch = d_dbg_best;
    c->dbg.cps = d_cps;
    c->dbg.n = n;
    c->dbg.n_text = n_text;
    *n_ids_out = n_ids;
  }

  void run(size_t *n_ids_out) {
    plan();
    symbols_and_keys();
    if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[2], st));
    sort_round0();
    ranks_round0();
    refine();
    if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[3], st));
    if (v->lcp_kasai) {  // alternative LCP builder: chunked Kasai exactly as linear.cpp:18-70
      const size_t chunk = 64;
      hipLaunchKernelGGL(HIP_KERNEL_NAME(kasai_kernel<SymT>), dim3(cdiv(cdiv(n, chunk), kBlock)), dim3(kBlock), 0, st, d_sym, d_sa,
                         d_rank, n, chunk, d_lcp);
      WP_LAUNCH_CHECK();
    }
    if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[4], st));
    scanlines();
    if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[5], st));
    size_t n_ids = 0;
    int32_t *d_ids = walk(&n_ids);
    if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[6], st));
    finish(d_ids, n_ids_out);
  }
};

// The whole device path on context c (the calling thread has c's device current).  d_text must be
// 4-byte aligned and readable up to the next multiple of 16.  S: statistics of this call.
static void encode_on_device(const wp_vocab *v, Context *c, const uint8_t *d_text, size_t nbytes, size_t *n_ids_out,
                             wp_stats &S) {
  hipStream_t st = c->stream;
  const HostVocab &hv = v->hv;
  std::memset(&S, 0, sizeof(S));
  S.n_bytes = static_cast<int64_t>(nbytes);
  S.longest_token = hv.longest;
  c->d_ids = nullptr;
  c->dbg = {};
  *n_ids_out = 0;
  if (nbytes == 0) return;  // linear.cpp:323-325
  // (no limit on the byte length: the reference limits total_length = code points + vocab symbols,
  // linear.cpp:104-106, checked below once the code points are counted — in 64 bits, since the tile
  // prefix itself is 32-bit and wraps for inputs beyond 4 G code points)
  const bool guard = v->arena_guard || EnvOptions::get().arena_guard;

  c->rstats.passes = 0;
  c->rstats.elems = 0;
  c->rstats.digit_bytes = 0;
  c->rstats.bytes = 0;
  c->rstats.spans.on = v->stage_timing;
  c->rstats.spans.used = 0;
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[0], st));

  // ---------------- phase A: decode ----------------
  const unsigned dec_tiles = cdiv(nbytes, kDecTile);
  Arena aa(&c->a_buf, guard);
  uint32_t *d_tile_cnt = nullptr, *d_cnt_tmp = nullptr, *d_cps = nullptr;
  uint8_t *d_cls = nullptr;
  for (int pass = 0; pass < 2; pass++) {
    d_tile_cnt = aa.take<uint32_t>(dec_tiles + 1);
    d_cnt_tmp = aa.take<uint32_t>(cdiv(dec_tiles, kScanTile) + 8);
    d_cps = v->keep_debug ? aa.take<uint32_t>(nbytes + 1) : nullptr;  // raw code points: debug copy only
    d_cls = aa.take<uint8_t>(nbytes + 32);  // (the walk reads 16 class bytes from any position on)
    if (pass == 0) aa.commit();
  }
  aa.arm(st);
  WP_HIP(hipMemsetAsync(c->d_scalars, 0, sizeof(uint32_t) * kScalars, st));
  WP_HIP(hipMemsetAsync(c->d_used, 0, sizeof(uint32_t) * kCpWords, st));
  hipLaunchKernelGGL(decode_count_kernel<true>, dim3(dec_tiles), dim3(kBlock), 0, st, d_text, nbytes, d_tile_cnt,
                     reinterpret_cast<unsigned long long *>(c->d_scalars + 2), c->d_used);
  device_exclusive_scan(d_tile_cnt, d_tile_cnt, dec_tiles, d_cnt_tmp, c->d_scalars + 0, st, nullptr,
                        reinterpret_cast<unsigned long long *>(c->d_scalars + 14));
  // does the text itself hold code point 0 or 1 (the separator)?  (read before the vocab marks its symbols)
  // (bits 0 and 1 of the first bitmap word)
  WP_HIP(hipMemcpyAsync(c->d_scalars + 20, c->d_used, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(vocab_alphabet_kernel, dim3(cdiv(hv.used_word_idx.size(), kBlock)), dim3(kBlock), 0, st,
                     c->d_vocab_word_idx, c->d_vocab_word_bits, static_cast<uint32_t>(hv.used_word_idx.size()), c->d_used);
  // alphabet: bitmap -> per-word prefixes + sigma -> lut (dense symbol of a used code point c = lut[c] + 1)
  hipLaunchKernelGGL(alphabet_prefix_kernel, dim3(1), dim3(kAlphaThreads), 0, st, c->d_used, c->d_scan_tmp, c->d_scalars + 1);
  hipLaunchKernelGGL(alphabet_lut_kernel, dim3(kCpTableSize / kBlock), dim3(kBlock), 0, st, c->d_used, c->d_scan_tmp, c->d_lut);
  WP_LAUNCH_CHECK();
  fetch_scalars(c, 22);
  unsigned long long n_text64;
  std::memcpy(&n_text64, c->h_scalars + 14, sizeof(n_text64));
  if (n_text64 + 1 + hv.stream.size() > 2000000000ull) throw std::length_error("64bit not implemented");  // linear.cpp:104-106
  const size_t n_text = c->h_scalars[0];
  // Layout of S.  The reference concatenates the whole vocabulary behind the text in every call and batch
  // (linear.cpp:77-101, 333, 347, 367).  Here the vocabulary normally stays out of the suffix sort: S = text . 1,
  // and the tokens come in through their code streams (prune.h).  The reference's layout is kept for the true
  // suffix array (full depth, duplicate vocab lines), for texts or tokens that hold the code points 0 / 1
  // (they sort around the separator), and on request (WP_OPT_VOCAB_IN_S).
  const bool full_sa = v->full_depth || hv.n_dup_eligible > 0 || v->lcp_kasai;
  const bool text_only = !full_sa && !v->vocab_in_s && !EnvOptions::get().vocab_in_s && !hv.low_cp && (c->h_scalars[20] & 3u) == 0;
  S.vocab_in_s = text_only ? 0 : 1;
  const uint32_t sigma = c->h_scalars[1];
  unsigned long long dropped;
  std::memcpy(&dropped, c->h_scalars + 2, sizeof(dropped));
  if (dropped != 0) std::cerr << "WARNING Input contains invalid unicode characters." << std::endl;

  const size_t n = n_text + 1 + (text_only ? 0 : hv.stream.size());  // total_length, linear.cpp:77-82
  S.n_text = static_cast<int64_t>(n_text);
  S.n_total = static_cast<int64_t>(n);
  S.alphabet = sigma;
  if (n > 2000000000ull) throw std::length_error("64bit not implemented");  // linear.cpp:104-106
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[1], st));

  const int bits = std::max(1, bit_length(sigma));  // symbols are 1..sigma, 0 = past the end
  for (int attempt = 0;; attempt++) {
    Arena ab(&c->b_buf, guard);
    try {
      if (sigma <= 255) {
        LinearPath<uint8_t>(v, c, S, ab, aa, d_text, nbytes, d_tile_cnt, n_text, n, d_cps, d_cls, bits, text_only).run(n_ids_out);
      } else {
        LinearPath<uint32_t>(v, c, S, ab, aa, d_text, nbytes, d_tile_cnt, n_text, n, d_cps, d_cls, bits, text_only).run(n_ids_out);
      }
      S.list_retries = attempt;
      return;
    } catch (const ListOverflow &o) {
      // the needed list of round 0 did not fit the room it was given (an eighth of the text, or what the last encode on
      // this context needed): nothing ran on the list — once more from the symbols on, with room for what it asked for
      if (attempt >= 2) throw HipError("needed list overflow after a retry (internal error)");
      WP_HIP(hipStreamSynchronize(c->stream));
      WP_HIP(hipStreamSynchronize(c->stream2));
      c->list_hint = o.wanted;
      c->rstats.passes = 0;
      c->rstats.elems = 0;
      c->rstats.digit_bytes = 0;
      c->rstats.bytes = 0;
      c->rstats.spans.used = 0;
      WP_HIP(hipMemsetAsync(c->d_scalars + 4, 0, sizeof(uint32_t) * 10, st));  // (list sizes, overflow flag, walk counters)
    }
  }
}

}  // namespace wp


