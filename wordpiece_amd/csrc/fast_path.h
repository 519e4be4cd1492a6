// fast_path.h — word_piece::fast on the device (fast.cpp:19-150; kernels in fast.h): decode -> code points + class
// bytes -> anchors -> trie walk per word -> id stream.
#pragma once
#include "context.h"
#include "fast.h"
#include "walk.h"

namespace wp {

static void encode_fast_on_device(const wp_vocab *v, Context *c, const uint8_t *d_text, size_t nbytes, size_t *n_ids_out,
                                  wp_stats &S) {
  hipStream_t st = c->stream;
  const HostVocab &hv = v->hv;
  std::memset(&S, 0, sizeof(S));
  S.n_bytes = static_cast<int64_t>(nbytes);
  S.longest_token = hv.fast_max_len;
  S.n_devices = 1;
  c->d_ids = nullptr;
  c->dbg = {};
  *n_ids_out = 0;
  if (nbytes == 0) return;  // fast.cpp:154-156
  Arena aa(&c->a_buf, v->arena_guard || EnvOptions::get().arena_guard);
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[0], st));
  const unsigned dec_tiles = cdiv(nbytes, kDecTile);
  uint32_t *d_tile_cnt = nullptr, *d_cnt_tmp = nullptr, *d_cps = nullptr;
  uint8_t *d_cls = nullptr;
  for (int pass = 0; pass < 2; pass++) {
    d_tile_cnt = aa.take<uint32_t>(dec_tiles + 1);
    d_cnt_tmp = aa.take<uint32_t>(cdiv(dec_tiles, kScanTile) + 8);
    d_cps = aa.take<uint32_t>(nbytes + 1);
    d_cls = aa.take<uint8_t>(nbytes + 16);
    if (pass == 0) aa.commit();
  }
  aa.arm(st);
  WP_HIP(hipMemsetAsync(c->d_scalars, 0, sizeof(uint32_t) * kScalars, st));
  hipLaunchKernelGGL(decode_count_kernel<false>, dim3(dec_tiles), dim3(kBlock), 0, st, d_text, nbytes, d_tile_cnt,
                     reinterpret_cast<unsigned long long *>(c->d_scalars + 2), static_cast<uint32_t *>(nullptr));
  device_exclusive_scan(d_tile_cnt, d_tile_cnt, dec_tiles, d_cnt_tmp, c->d_scalars + 0, st, nullptr,
                        reinterpret_cast<unsigned long long *>(c->d_scalars + 14));
  WP_LAUNCH_CHECK();
  fetch_scalars(c, 16);
  unsigned long long n_text64;
  std::memcpy(&n_text64, c->h_scalars + 14, sizeof(n_text64));
  // positions are 32-bit and bit 31 of an anchor entry is the skip flag of the sparse / long-word walk (walk.h,
  // kAnchorSkip): the same kind of limit as linear.cpp:104-106, never silent truncation
  if (n_text64 >= (1ull << 31)) throw std::length_error("64bit not implemented (fast path: text of 2^31 or more code points)");
  const size_t n_text = c->h_scalars[0];
  unsigned long long dropped;
  std::memcpy(&dropped, c->h_scalars + 2, sizeof(dropped));
  if (dropped != 0) std::cerr << "WARNING Input contains invalid unicode characters." << std::endl;
  S.n_text = static_cast<int64_t>(n_text);
  S.n_total = static_cast<int64_t>(n_text);
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[1], st));
  if (n_text == 0) return;

  Arena ar(&c->b_buf, aa.guard);
  const size_t tiles = cdiv(n_text, kScanTile), atiles = cdiv(n_text, kAnchorTile), walk_blocks = cdiv(n_text, kBlock);
  const bool staged_possible = !EnvOptions::get().sparse_emit && !v->sparse_emit;
  const uint32_t lw_cap = static_cast<uint32_t>(n_text / kMaxAnchorGap + 2);
  int32_t *d_emit = nullptr, *d_ids = nullptr, *d_lid = nullptr;
  uint32_t *d_anchors = nullptr, *d_anchor_cnt = nullptr, *d_anchor_tmp = nullptr, *d_emit_cnt = nullptr, *d_emit_tmp = nullptr,
           *d_blk_cnt = nullptr, *d_blk_off = nullptr, *jump_a = nullptr, *jump_b = nullptr, *d_lw_off = nullptr, *d_lw_fail = nullptr;
  uint8_t *d_mark = nullptr;
  LongWord *d_lw = nullptr;
  for (int pass = 0; pass < 2; pass++) {
    d_emit = ar.take<int32_t>(n_text + 1);
    d_ids = ar.take<int32_t>(n_text + 1);
    d_anchors = ar.take<uint32_t>(n_text + 1);
    d_anchor_cnt = ar.take<uint32_t>(atiles + 1);
    d_anchor_tmp = ar.take<uint32_t>(cdiv(atiles, kScanTile) + 8);
    d_emit_cnt = ar.take<uint32_t>(tiles + 1);
    d_emit_tmp = ar.take<uint32_t>(cdiv(walk_blocks, kScanTile) + 8);
    d_blk_cnt = ar.take<uint32_t>(walk_blocks + 2);
    d_blk_off = ar.take<uint32_t>(walk_blocks + 2);
    d_lid = ar.take<int32_t>(n_text + 1);
    jump_a = ar.take<uint32_t>(n_text + 1);
    jump_b = ar.take<uint32_t>(n_text + 1);
    d_mark = ar.take<uint8_t>(n_text + 1);
    d_lw = ar.take<LongWord>(lw_cap);
    d_lw_off = ar.take<uint32_t>(lw_cap + 1);
    d_lw_fail = ar.take<uint32_t>(lw_cap + 1);
    if (pass == 0) ar.commit();
  }
  ar.arm(st);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(decode_write_kernel<uint32_t>), dim3(dec_tiles), dim3(kBlock), 0, st, d_text, nbytes,
                     d_tile_cnt, static_cast<const uint32_t *>(nullptr), static_cast<uint32_t *>(nullptr), d_cls, d_cps,
                     c->d_cls_bmp, static_cast<const uint32_t *>(nullptr), 0, static_cast<uint32_t *>(nullptr), 0);
  hipLaunchKernelGGL(fast_anchor_count_kernel, dim3(atiles), dim3(kBlock), 0, st, d_cls, n_text, d_anchor_cnt);
  device_exclusive_scan(d_anchor_cnt, d_anchor_cnt, atiles, d_anchor_tmp, c->d_scalars + 10, st);
  hipLaunchKernelGGL(fast_anchor_write_kernel, dim3(atiles), dim3(kBlock), 0, st, d_cls, n_text, d_anchor_cnt, d_anchors);
  hipLaunchKernelGGL(fast_anchor_gap_kernel, dim3(std::min<size_t>(atiles, 1024)), dim3(kBlock), 0, st, d_anchors,
                     c->d_scalars + 10, n_text, d_cls, c->d_scalars + 11);
  WP_LAUNCH_CHECK();
  fetch_scalars(c, 12);
  const size_t n_anchors = c->h_scalars[10], max_gap = c->h_scalars[11];
  S.n_anchors = static_cast<int64_t>(n_anchors);
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[2], st));
  FastArgs fa{d_cps, d_cls, n_text,
              TrieView{c->d_trie_key, c->d_trie_child, c->d_trie_id, static_cast<uint32_t>(hv.trie_key.size() - 1)},
              c->d_tok_len, hv.unk_id, static_cast<uint32_t>(std::min<uint64_t>(static_cast<uint64_t>(hv.fast_max_len), n_text)),
              d_emit};
  // ids as per-workgroup lists (walk.h, StagedOut) unless the long-word kernels contribute ids of their own
  const bool staged = staged_possible && max_gap <= kMaxAnchorGap;
  if (!staged) WP_HIP(hipMemsetAsync(d_emit, 0x80, n_text * sizeof(int32_t), st));
  if (max_gap > kMaxAnchorGap) {  // long words: pointer doubling instead of one lane per word (walk.h)
    hipLaunchKernelGGL(fast_long_word_collect_kernel, dim3(std::min<size_t>(cdiv(std::max<size_t>(n_anchors, 1), kBlock), 2048)),
                       dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, n_text, d_cls, d_lw, lw_cap, c->d_scalars + 12);
    WP_LAUNCH_CHECK();
    fetch_scalars(c, 13);
    const uint32_t nw = std::min(c->h_scalars[12], lw_cap);
    if (nw > 0) {
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
      const uint32_t total = static_cast<uint32_t>(total64);  // <= n_text
      WP_HIP(hipMemcpyAsync(d_lw_off, h_off.data(), sizeof(uint32_t) * (nw + 1), hipMemcpyHostToDevice, st));
      WP_HIP(hipMemsetAsync(d_lw_fail, 0, sizeof(uint32_t) * nw, st));
      const dim3 grid(cdiv(total, kBlock));
      hipLaunchKernelGGL(fast_long_word_next_kernel, grid, dim3(kBlock), 0, st, fa, d_lw, d_lw_off, nw, total, d_lid, jump_a,
                         d_mark);
      WP_HIP(hipStreamSynchronize(st));  // h_off is a stack-owned upload source
      uint32_t *ja = jump_a, *jb = jump_b;
      for (uint32_t reach = 1; reach < longest; reach *= 2) {
        hipLaunchKernelGGL(long_word_mark_kernel, grid, dim3(kBlock), 0, st, ja, total, d_mark);
        hipLaunchKernelGGL(long_word_double_kernel, grid, dim3(kBlock), 0, st, ja, total, jb);
        std::swap(ja, jb);
      }
      hipLaunchKernelGGL(long_word_mark_kernel, grid, dim3(kBlock), 0, st, ja, total, d_mark);
      hipLaunchKernelGGL(long_word_fail_kernel, grid, dim3(kBlock), 0, st, d_lid, d_mark, d_lw_off, nw, total, d_lw_fail);
      hipLaunchKernelGGL(fast_long_word_emit_kernel, grid, dim3(kBlock), 0, st, fa, d_lw, d_lw_off, nw, total, d_lid, d_mark,
                         d_lw_fail);
      WP_LAUNCH_CHECK();
      S.anchor_mode = 2;
    }
  }
  const size_t acap = std::max<size_t>(n_anchors, 1);
  const unsigned wblocks = cdiv(acap, kBlock);
  if (staged) {
    const int words = kWbWords;
    const unsigned sblocks = cdiv(acap, static_cast<size_t>(words));
    hipLaunchKernelGGL(HIP_KERNEL_NAME(walk_balanced_kernel<FastArgs, FastStep, false>), dim3(sblocks), dim3(kBlock), 0, st, fa,
                       d_anchors, c->d_scalars + 10, acap, d_lid, d_blk_cnt,  // (d_lid: the long-word id buffer, idle here)
                       static_cast<const uint32_t *>(nullptr));
    device_exclusive_scan(d_blk_cnt, d_blk_off, sblocks, d_emit_tmp, c->d_scalars + 9, st);
    hipLaunchKernelGGL(emit_gather_kernel, dim3(sblocks), dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, acap, d_lid,
                       d_blk_cnt, d_blk_off, d_ids, words);
  } else {
    hipLaunchKernelGGL(fast_walk_kernel, dim3(wblocks), dim3(kBlock), 0, st, fa, d_anchors, c->d_scalars + 10, acap);
    hipLaunchKernelGGL(emit_count_kernel, dim3(tiles), dim3(kBlock), 0, st, d_emit, n_text, d_emit_cnt);
    device_exclusive_scan(d_emit_cnt, d_emit_cnt, tiles, d_emit_tmp, c->d_scalars + 9, st);
    hipLaunchKernelGGL(emit_write_kernel, dim3(tiles), dim3(kBlock), 0, st, d_emit, n_text, d_emit_cnt, d_ids);
  }
  S.staged_emit = staged ? 1 : 0;
  WP_LAUNCH_CHECK();
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[3], st));
  if (ar.guard) {
    static const uint32_t init[2] = {0u, 0xffffffffu};
    WP_HIP(hipMemcpyAsync(c->d_scalars + 16, init, sizeof(init), hipMemcpyHostToDevice, st));
    ar.check(st, c->d_scalars + 16);
    aa.check(st, c->d_scalars + 16);
    fetch_scalars(c, 18);
    if (c->h_scalars[16] != 0) throw HipError("arena guard: guard zone overwritten in the fast path");
    S.guard_zones = static_cast<int32_t>(ar.zones.size() + aa.zones.size());
  }
  fetch_scalars(c, 10);
  const size_t n_ids = c->h_scalars[9];
  S.n_ids = static_cast<int64_t>(n_ids);
  if (v->stage_timing) {
    auto span = [&](int a, int b) {
      float ms = 0;
      WP_HIP(hipEventElapsedTime(&ms, c->ev[a], c->ev[b]));
      return static_cast<double>(ms);
    };
    S.ms_decode = span(0, 2);
    S.ms_walk = span(2, 3);
    S.ms_total = span(0, 3);
  }
  c->d_ids = d_ids;
  *n_ids_out = n_ids;
}


}  // namespace wp
