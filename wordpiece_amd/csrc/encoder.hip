// encoder.hip — orchestration of the HIP Linear-WordPiece path and the C ABI (include/wordpiece_amd.h).
//
// Stages (reference file:line in parentheses):
//   decode + classes + S build   utils.cpp:37-79, utf8.cpp:54-90, linear.cpp:77-103     decode.h
//   suffix array / rank / LCP    linear.cpp:118-149 (libsais_int, inverse SA, calcLcp)   suffix_array.h, radix_sort.h
//   who marks + 4 scanlines      linear.cpp:153-213                                        scanline.h
//   greedy walk + id stream      linear.cpp:215-316                                        walk.h
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstring>
#include <fstream>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/wordpiece_amd.h"
#include "code.h"
#include "decode.h"
#include "fast.h"
#include "format.h"
#include "local_sort.h"
#include "prune.h"
#include "radix_sort.h"
#include "scanline.h"
#include "suffix_array.h"
#include "trie.h"
#include "vocab.h"
#include "walk.h"

namespace wp {

static thread_local std::string g_last_error;

struct DeviceBuffer {
  void *p = nullptr;
  size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (p) WP_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + (1 << 20);
    WP_HIP(hipMalloc(&p, want));
    cap = want;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Guard zones (WP_OPT_ARENA_GUARD / env WP_ARENA_GUARD=1, a debugging aid): every arena allocation is
// followed by kGuardBytes of a fixed pattern; after the encode a kernel checks that every zone is
// intact, i.e. that no kernel wrote past the end (or before the start) of the buffer it was given.
constexpr size_t kGuardBytes = 256;
constexpr uint32_t kGuardWord = 0xA5C3F00Du;

__global__ __launch_bounds__(kBlock) void guard_fill_kernel(char *base, const unsigned long long *offs, int count) {
  const int z = blockIdx.x;
  if (z >= count) return;
  uint32_t *g = reinterpret_cast<uint32_t *>(base + offs[z]);
  if (threadIdx.x < kGuardBytes / 4) g[threadIdx.x] = kGuardWord;
}
// bad[0] = number of damaged zones, bad[1] = 1 + index of the first one
__global__ __launch_bounds__(kBlock) void guard_check_kernel(const char *base, const unsigned long long *offs, int count,
                                                             uint32_t *bad) {
  const int z = blockIdx.x;
  if (z >= count) return;
  const uint32_t *g = reinterpret_cast<const uint32_t *>(base + offs[z]);
  const bool broken = threadIdx.x < kGuardBytes / 4 && g[threadIdx.x] != kGuardWord;
  if (__syncthreads_or(broken) && threadIdx.x == 0) {
    atomicAdd(&bad[0], 1u);
    atomicMin(&bad[1], static_cast<uint32_t>(z) + 1u);
  }
}

// bump allocator over a DeviceBuffer: plan() first with the same sequence of take() calls
struct Arena {
  DeviceBuffer *buf;
  size_t off = 0;
  bool planning = true, guard = false;
  std::vector<unsigned long long> zones;  // byte offsets of the guard zones (guard mode)
  explicit Arena(DeviceBuffer *b, bool g = false) : buf(b), guard(g) {}
  template <typename T>
  T *take(size_t count) {
    size_t bytes = (count * sizeof(T) + 255) & ~static_cast<size_t>(255);
    size_t o = off;
    off += bytes;
    if (guard) {
      if (!planning) zones.push_back(off);
      off += kGuardBytes;
    }
    if (planning) return nullptr;
    return reinterpret_cast<T *>(static_cast<char *>(buf->p) + o);
  }
  void commit() {
    buf->ensure(off + (guard ? 8 * 512 : 0));  // (guard mode: room for the zone table behind the arena)
    off = 0;
    planning = false;
  }
  // the zone table lives behind the last allocation; call after the second (real) round of take()s
  unsigned long long *zone_table() const {
    return reinterpret_cast<unsigned long long *>(static_cast<char *>(buf->p) + ((off + 255) & ~static_cast<size_t>(255)));
  }
  void arm(hipStream_t st) {
    if (!guard || zones.empty()) return;
    if (zones.size() > 500) throw std::logic_error("arena guard: too many allocations");
    WP_HIP(hipMemcpyAsync(zone_table(), zones.data(), zones.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
    WP_HIP(hipStreamSynchronize(st));
    hipLaunchKernelGGL(guard_fill_kernel, dim3(zones.size()), dim3(kBlock), 0, st, static_cast<char *>(buf->p), zone_table(),
                       static_cast<int>(zones.size()));
  }
  // bad: 2 device words, cleared by the caller to {0, 0xffffffff}
  void check(hipStream_t st, uint32_t *bad) const {
    if (!guard || zones.empty()) return;
    hipLaunchKernelGGL(guard_check_kernel, dim3(zones.size()), dim3(kBlock), 0, st, static_cast<const char *>(buf->p),
                       zone_table(), static_cast<int>(zones.size()), bad);
  }
};

static int bit_length(uint64_t v) {
  int b = 0;
  while (v) {
    b++;
    v >>= 1;
  }
  return b;
}

constexpr int kScalars = 32;

struct Context {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // side stream: latency-bound helpers overlap the bandwidth-bound kernels
  hipEvent_t evs[3] = {};         // fork / join / scalars fetched
  // vocab tables on the device
  uint32_t *d_stream = nullptr, *d_elig_start = nullptr, *d_elig_info = nullptr, *d_soft = nullptr;
  uint32_t *d_vocab_word_idx = nullptr, *d_vocab_word_bits = nullptr;  // the vocabulary's words of the alphabet bitmap
  uint8_t *d_cls_bmp = nullptr;                                         // class byte of every BMP code point
  uint32_t *d_lt_chain_len = nullptr, *d_lt_chain_off = nullptr, *d_lt_child_begin = nullptr, *d_lt_child_cp = nullptr,
           *d_lt_child_node = nullptr, *d_elig_node = nullptr, *d_elig_subtree = nullptr;  // the token trie (vocab.h, trie.h)
  int32_t *d_elig_id = nullptr, *d_tok_len = nullptr;
  unsigned long long *d_trie_key = nullptr;  // the fast path's token trie (vocab.h)
  uint32_t *d_trie_child = nullptr;
  int32_t *d_trie_id = nullptr;
  DeviceBuffer text_buf, a_buf, b_buf, fmt_buf;  // fmt_buf: id text of encodeExternal
  // wp_linear_encode_batch: second text buffer, two id staging buffers and the copy streams of the shard pipeline
  DeviceBuffer text_buf2, ids_stage[2];
  hipStream_t up_stream = nullptr, down_stream = nullptr;
  hipEvent_t pipe_ev[4] = {};  // ids staged [2], ids downloaded [2]
  uint32_t *d_used = nullptr, *d_lut = nullptr, *d_scan_tmp = nullptr;  // bitmap of the code points in use (kCpWords), lut (kCpTableSize), per-word prefixes (kCpWords)
  uint32_t *d_scalars = nullptr;                                         // kScalars words of device scalars
  uint8_t *d_code = nullptr;     // symbol code tables: cw u16[256] | len u8[256] | bmask u16[4096]
  // The symbol code of the last encode, kept while the alphabet size stays the same: ANY order-preserving code over
  // the dense symbol ids 0..sigma is correct (the histogram only steers the codeword lengths), so consecutive
  // shards / batches of one corpus reuse it and skip the histogram download, the host-side construction and the
  // table upload — one host round trip less per encode.  Rebuilt every kCodeReuse encodes to follow the text.
  SymbolCode code_cache;
  bool code_cached = false;
  uint32_t code_alphabet = 0;
  int code_bits = 0, code_lo = 0, code_uses = 0;
  uint8_t *h_code = nullptr;     // pinned staging of the same (the upload needs no host wait: every encode ends with one)
  uint32_t *d_symhist = nullptr;  // 256 counters
  uint32_t *h_scalars = nullptr;                                         // pinned mirror
  RadixStats rstats;
  hipEvent_t ev[8] = {};
  // results / debug views of the last call (device pointers into the arenas)
  const int32_t *d_ids = nullptr;
  struct {
    const void *sym = nullptr;
    int sym_bytes = 0;
    const uint32_t *sa = nullptr, *cps = nullptr;
    const RankEntry *rank = nullptr;
    const int32_t *lcp = nullptr;
    StepTable steps{};
    int32_t *best_scratch = nullptr;  // room for 2n int32 (debug expansion of the step functions)
    size_t n = 0, n_text = 0;
  } dbg;
  Context() = default;
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  ~Context();  // releases whatever was built (a half-built context of a failed make_context included)
};

// The calling thread's current HIP device, put back when the scope ends: no entry point of the C ABI leaves the
// caller on another device than it came in with (a host process — PyTorch, say — keeps allocating on "its" GPU).
struct DeviceGuard {
  int prev = -1;
  DeviceGuard() {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};

}  // namespace wp

using namespace wp;

struct wp_vocab {
  HostVocab hv;
  std::unique_ptr<Context> ctx;                  // the handle's own device context
  std::vector<std::unique_ptr<Context>> multi;  // one per entry of the device list of wp_linear_encode_multi
  int device = -1;
  bool full_depth = false, keep_debug = false, stage_timing = false, lcp_kasai = false, fused_rerank = false, cover_anchors = false;
  bool arena_guard = false;
  bool sparse_emit = false;  // WP_OPT_SPARSE_EMIT: ids through the per-position emit array even where per-workgroup lists would do
  bool vocab_in_s = false;  // WP_OPT_VOCAB_IN_S: always the reference's S = text . 1 . vocab layout
  int n_devices = 1;  // WP_OPT_DEVICES: GPUs wp_linear_encode shards a host buffer over (-1: all visible)
  wp_stats stats{};
  ~wp_vocab();
};

namespace wp {

static void free_vocab_tables(Context *c) {
  for (void **p : {reinterpret_cast<void **>(&c->d_stream), reinterpret_cast<void **>(&c->d_elig_start),
                   reinterpret_cast<void **>(&c->d_elig_info), reinterpret_cast<void **>(&c->d_soft),
                   reinterpret_cast<void **>(&c->d_elig_id), reinterpret_cast<void **>(&c->d_tok_len),
                   reinterpret_cast<void **>(&c->d_trie_key), reinterpret_cast<void **>(&c->d_trie_child),
                   reinterpret_cast<void **>(&c->d_trie_id), reinterpret_cast<void **>(&c->d_vocab_word_idx),
                   reinterpret_cast<void **>(&c->d_vocab_word_bits), reinterpret_cast<void **>(&c->d_cls_bmp),
                   reinterpret_cast<void **>(&c->d_lt_chain_len), reinterpret_cast<void **>(&c->d_lt_chain_off),
                   reinterpret_cast<void **>(&c->d_lt_child_begin), reinterpret_cast<void **>(&c->d_lt_child_cp),
                   reinterpret_cast<void **>(&c->d_lt_child_node), reinterpret_cast<void **>(&c->d_elig_node),
                   reinterpret_cast<void **>(&c->d_elig_subtree)}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
}

// idempotent: every resource is cleared as it is released (runs from ~Context too)
static void destroy_context(Context *c) {
  if (!c) return;
  const bool owns = c->stream || c->stream2 || c->d_used || c->d_lut || c->d_scan_tmp || c->d_scalars || c->d_code ||
                    c->d_symhist || c->h_scalars || c->h_code || c->d_stream || c->text_buf.p || c->a_buf.p ||
                    c->b_buf.p || c->fmt_buf.p;
  if (!owns) return;
  DeviceGuard keep;
  (void)hipSetDevice(c->device);
  free_vocab_tables(c);
  for (void **p : {reinterpret_cast<void **>(&c->d_used), reinterpret_cast<void **>(&c->d_lut),
                   reinterpret_cast<void **>(&c->d_scan_tmp), reinterpret_cast<void **>(&c->d_scalars),
                   reinterpret_cast<void **>(&c->d_code), reinterpret_cast<void **>(&c->d_symhist)}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  if (c->h_scalars) (void)hipHostFree(c->h_scalars);
  if (c->h_code) (void)hipHostFree(c->h_code);
  c->h_scalars = nullptr;
  c->h_code = nullptr;
  c->text_buf.release();
  c->a_buf.release();
  c->b_buf.release();
  c->fmt_buf.release();
  c->text_buf2.release();
  c->ids_stage[0].release();
  c->ids_stage[1].release();
  for (auto &e : c->pipe_ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
  if (c->down_stream) (void)hipStreamDestroy(c->down_stream);
  c->up_stream = c->down_stream = nullptr;
  for (auto &e : c->ev) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  for (auto &e : c->evs) {
    if (e) (void)hipEventDestroy(e);
    e = nullptr;
  }
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = c->stream2 = nullptr;
}
Context::~Context() { destroy_context(this); }

template <typename T>
static T *upload(const std::vector<T> &v, hipStream_t st) {
  T *d = nullptr;
  WP_HIP(hipMalloc(&d, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) WP_HIP(hipMemcpyAsync(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
  return d;
}

// The reference's API has no handles: every word_piece::linear::encode(text, vocab) parses the vocabulary and
// sets everything up again (linear.cpp:332-341), and its test-suite does that tens of thousands of times.
// Here a context (two streams, events, code point tables, scalars, the arenas) costs ~1.5 ms to make, so
// the contexts of destroyed handles are parked in a small process-wide pool and the next handle on the
// same device takes one over, replacing only the vocabulary tables.  What a parked context keeps is SMALL
// state: arenas of more than kPoolArenaBytes in total go back to the driver when the handle is destroyed (a
// destroyed handle must not sit on the gigabytes its last encode needed — the pool exists for the sub-millisecond
// one-shot calls on tiny inputs); wp_trim() releases the rest.
static constexpr size_t kPoolArenaBytes = size_t(256) << 20;
static constexpr size_t kPoolContexts = 4;
static std::mutex g_pool_mu;
static std::vector<std::unique_ptr<Context>> &context_pool() {
  static auto *pool = new std::vector<std::unique_ptr<Context>>();  // never destroyed: the HIP runtime may be gone by then
  return *pool;
}

static void release_arenas(Context *c) {
  c->text_buf2.release();
  c->ids_stage[0].release();
  c->ids_stage[1].release();
  c->text_buf.release();
  c->a_buf.release();
  c->b_buf.release();
  c->fmt_buf.release();
  c->d_ids = nullptr;
  c->dbg = {};
}

static void park_context(std::unique_ptr<Context> c) {
  if (!c) return;
  static const bool no_pool = getenv("WP_NO_CONTEXT_POOL") && atoi(getenv("WP_NO_CONTEXT_POOL")) != 0;
  DeviceGuard keep;
  (void)hipSetDevice(c->device);
  if (!no_pool && hipStreamSynchronize(c->stream) == hipSuccess && hipStreamSynchronize(c->stream2) == hipSuccess) {
    free_vocab_tables(c.get());
    if (c->text_buf.cap + c->text_buf2.cap + c->ids_stage[0].cap + c->ids_stage[1].cap + c->a_buf.cap + c->b_buf.cap + c->fmt_buf.cap >
        kPoolArenaBytes) {
      release_arenas(c.get());
    }
    c->d_ids = nullptr;
    c->dbg = {};
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (context_pool().size() < kPoolContexts) {
      context_pool().push_back(std::move(c));
      return;
    }
  }
  destroy_context(c.get());
}

static void upload_vocab_tables(Context *c, const HostVocab &hv) {
  c->d_stream = upload(hv.stream, c->stream);
  c->d_elig_start = upload(hv.elig_start, c->stream);
  c->d_elig_info = upload(hv.elig_info, c->stream);
  c->d_elig_id = upload(hv.elig_id, c->stream);
  c->d_tok_len = upload(hv.tok_len, c->stream);
  c->d_soft = upload(hv.soft, c->stream);
  c->d_vocab_word_idx = upload(hv.used_word_idx, c->stream);
  c->d_vocab_word_bits = upload(hv.used_word_bits, c->stream);
  c->d_cls_bmp = upload(hv.cls_bmp, c->stream);
  c->d_lt_chain_len = upload(hv.lt_chain_len, c->stream);
  c->d_lt_chain_off = upload(hv.lt_chain_off, c->stream);
  c->d_lt_child_begin = upload(hv.lt_child_begin, c->stream);
  c->d_lt_child_cp = upload(hv.lt_child_cp, c->stream);
  c->d_lt_child_node = upload(hv.lt_child_node, c->stream);
  c->d_elig_node = upload(hv.elig_node, c->stream);
  c->d_elig_subtree = upload(hv.elig_subtree, c->stream);
  {
    std::vector<unsigned long long> tk(hv.trie_key.begin(), hv.trie_key.end());
    c->d_trie_key = upload(tk, c->stream);
    WP_HIP(hipStreamSynchronize(c->stream));  // tk is a local
  }
  c->d_trie_child = upload(hv.trie_child, c->stream);
  c->d_trie_id = upload(hv.trie_id, c->stream);
  WP_HIP(hipStreamSynchronize(c->stream));
}

// a context (streams, vocab tables, scratch) on `device` (< 0: the calling thread's current device): a parked
// one if there is any, else a fresh one
static std::unique_ptr<Context> make_context(const wp_vocab *v, int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    throw HipError("no HIP device available: the Linear WordPiece path has no CPU fallback");
  }
  if (device >= 0) {
    if (device >= count) throw std::invalid_argument("no such HIP device: " + std::to_string(device));
  } else {
    WP_HIP(hipGetDevice(&device));
  }
  WP_HIP(hipSetDevice(device));
  std::unique_ptr<Context> c;
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    auto &pool = context_pool();
    for (size_t i = 0; i < pool.size(); i++) {
      if (pool[i]->device == device) {
        c = std::move(pool[i]);
        pool.erase(pool.begin() + static_cast<long>(i));
        break;
      }
    }
  }
  if (c) {
    upload_vocab_tables(c.get(), v->hv);  // (a throw destroys the context: ~Context)
    return c;
  }
  c.reset(new Context());  // (a throwing WP_HIP below releases what was built so far: ~Context)
  c->device = device;
  WP_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  WP_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
  for (auto &e : c->evs) WP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  WP_HIP(hipMalloc(&c->d_used, sizeof(uint32_t) * kCpWords));
  WP_HIP(hipMalloc(&c->d_lut, sizeof(uint32_t) * kCpTableSize));
  WP_HIP(hipMalloc(&c->d_scan_tmp, sizeof(uint32_t) * kCpWords));
  WP_HIP(hipMalloc(&c->d_scalars, sizeof(uint32_t) * kScalars));
  WP_HIP(hipMalloc(&c->d_code, 512 + 256 + kDecodeTableBytes));
  WP_HIP(hipHostMalloc(&c->h_code, 512 + 256 + kDecodeTableBytes));
  WP_HIP(hipMalloc(&c->d_symhist, sizeof(uint32_t) * 256));
  WP_HIP(hipHostMalloc(&c->h_scalars, sizeof(uint32_t) * kScalars));
  for (auto &e : c->ev) WP_HIP(hipEventCreate(&e));
  upload_vocab_tables(c.get(), v->hv);
  return c;
}

static Context *get_context(wp_vocab *v) {
  if (!v->ctx) v->ctx = make_context(v, v->device);
  WP_HIP(hipSetDevice(v->ctx->device));
  return v->ctx.get();
}

// copies `count` device scalars (from d_scalars) to the pinned mirror and waits
static void fetch_scalars(Context *c, int count) {
  WP_HIP(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(uint32_t) * count, hipMemcpyDeviceToHost, c->stream));
  WP_HIP(hipStreamSynchronize(c->stream));
}

template <typename SymT>
static void run_sa_and_beyond(const wp_vocab *v, Context *c, wp_stats &S, Arena &ar, Arena &aa, const uint8_t *d_text,
                              size_t nbytes, const uint32_t *d_tile_prefix, size_t n_text, size_t n, uint32_t *d_cps,
                              uint8_t *d_cls, int bits, bool text_only, size_t *n_ids_out);

static bool env_flag(const char *name) {
  const char *e = getenv(name);
  return e && atoi(e) != 0;
}

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
  static const bool env_guard = env_flag("WP_ARENA_GUARD");
  const bool guard = v->arena_guard || env_guard;

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
    d_cls = aa.take<uint8_t>(nbytes + 1);
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
  static const bool env_vocab_in_s = env_flag("WP_VOCAB_IN_S");
  const bool full_sa = v->full_depth || hv.n_dup_eligible > 0 || v->lcp_kasai;
  const bool text_only = !full_sa && !v->vocab_in_s && !env_vocab_in_s && !hv.low_cp && (c->h_scalars[20] & 3u) == 0 &&
                         !env_flag("WP_NO_PRUNE");
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
  Arena ab(&c->b_buf, guard);
  if (sigma <= 255) {
    run_sa_and_beyond<uint8_t>(v, c, S, ab, aa, d_text, nbytes, d_tile_cnt, n_text, n, d_cps, d_cls, bits, text_only,
                               n_ids_out);
  } else {
    run_sa_and_beyond<uint32_t>(v, c, S, ab, aa, d_text, nbytes, d_tile_cnt, n_text, n, d_cps, d_cls, bits, text_only,
                                n_ids_out);
  }
}

template <typename SymT>
static void run_sa_and_beyond(const wp_vocab *v, Context *c, wp_stats &S, Arena &ar, Arena &aa, const uint8_t *d_text,
                              size_t nbytes, const uint32_t *d_tile_prefix, size_t n_text, size_t n, uint32_t *d_cps,
                              uint8_t *d_cls, int bits, bool text_only, size_t *n_ids_out) {
  hipStream_t st = c->stream;
  hipStream_t st2 = c->stream2;
  // st2 starts after everything queued on st so far / st continues after everything queued on st2
  auto fork = [&] {
    WP_HIP(hipEventRecord(c->evs[0], st));
    WP_HIP(hipStreamWaitEvent(st2, c->evs[0], 0));
  };
  auto join = [&] {
    WP_HIP(hipEventRecord(c->evs[1], st2));
    WP_HIP(hipStreamWaitEvent(st, c->evs[1], 0));
  };
  const HostVocab &hv = v->hv;
  const bool full = v->full_depth || hv.n_dup_eligible > 0 || v->lcp_kasai;
  const uint32_t need_depth = static_cast<uint32_t>(std::min<int64_t>(hv.longest + 1, 0x7fffffff));
  S.symbol_bits = bits;
  S.full_depth = full;

  const int M = static_cast<int>(hv.elig_id.size());
  const unsigned sl_tiles = cdiv(n, kSlTile);
  const int P = kStepsPerMark * M + 1;  // steps of the scanline result (scanline.h)
  const int bucket_shift = std::max(0, bit_length(n) - 18);
  const unsigned nbuckets = static_cast<unsigned>(((n - 1) >> bucket_shift) + 1);

  const size_t rr_tiles = cdiv(n, kRrTile);
  const size_t radix_words = std::max(radix_tmp_words<uint64_t>(n), radix_tmp_words<uint32_t>(std::max<size_t>(n, kStepsPerMark * std::max(M, 1) + 1)));
  const size_t emit_tiles = cdiv(std::max<size_t>(n_text, 1), kScanTile);
  const size_t walk_blocks = cdiv(std::max<size_t>(n_text, 1), kBlock);  // (at most one anchor per position)
  // ids as per-workgroup lists (walk.h, StagedOut) unless several kernels contribute ids: decided here, except
  // for long words, which only the anchor gaps reveal
  static const bool env_sparse_emit = env_flag("WP_SPARSE_EMIT");
  const bool staged_possible = !env_sparse_emit && !v->sparse_emit && !v->cover_anchors && hv.soft.empty();

  // digit bytes (radix_sort.h): the round-0 sort's histograms read 1 byte per key instead of 8
  static const bool env_no_digit_bytes = env_flag("WP_NO_DIGIT_BYTES");
  const bool use_digit_bytes = !env_no_digit_bytes && kRadixBits <= 8 && n > kRadixSmallN;
  // text-only layout: the needed groups are resolved along the token trie (trie.h) instead of by doubling rounds
  // (env WP_DOUBLING_ROUNDS=1 keeps the rounds: the A/B switch of this round's change)
  static const bool env_doubling = env_flag("WP_DOUBLING_ROUNDS");
  const bool use_trie = text_only && !full && M > 0 && !env_doubling && !env_flag("WP_NO_PRUNE");
  S.trie_refine = use_trie ? 1 : 0;
  uint32_t *d_node_of_slot = nullptr, *d_child_sym = nullptr, *d_gnode = nullptr, *d_gdone = nullptr;
  SymT *d_vsym = nullptr;
  SymT *d_sym = nullptr;
  uint8_t *DG0 = nullptr, *DG1 = nullptr, *d_rng_long = nullptr;
  uint32_t *d_rng_lo = nullptr, *d_rng_hi = nullptr;
  uint32_t *d_claim = nullptr, *d_claim_need = nullptr, *d_gclaim = nullptr, *d_gneed0 = nullptr, *d_gneed1 = nullptr;
  size_t claim_size = 1024;  // hash table of claimed key ranges (prune.h): a power of two >= 2 M
  while (claim_size < 2 * static_cast<size_t>(std::max(M, 1))) claim_size *= 2;
  uint64_t *K0 = nullptr, *K1 = nullptr;
  RankEntry *d_rank = nullptr;
  uint32_t *AD0 = nullptr, *AD1 = nullptr, *d_tdep = nullptr, *d_gdepth = nullptr, *d_anchors = nullptr,
           *d_anchor_cnt = nullptr, *d_anchor_tmp = nullptr;
  int32_t *d_emit = nullptr;
  uint32_t *V0 = nullptr, *V1 = nullptr, *AS0 = nullptr, *AS1 = nullptr, *AG = nullptr, *d_sa = nullptr,
           *d_radix_tmp = nullptr, *d_mslot0 = nullptr, *d_mslot1 = nullptr, *d_midx0 = nullptr,
           *d_midx1 = nullptr, *d_minfo = nullptr, *d_tile_mlo = nullptr, *d_emit_cnt = nullptr,
           *d_emit_tmp = nullptr, *d_blk_cnt = nullptr, *d_blk_off = nullptr;
  int32_t *d_lcp = nullptr, *d_mid = nullptr, *d_rf = nullptr, *d_rb = nullptr;
  RerankAgg *d_agg = nullptr, *d_chunk_agg = nullptr;
  uint32_t *d_ghead = nullptr, *d_large_id = nullptr, *d_large_off = nullptr, *d_lg_head = nullptr,
           *d_lg_off = nullptr, *LV0 = nullptr,
           *LV1 = nullptr, *LPOS = nullptr;
  uint64_t *LK1 = nullptr;
  int32_t *d_cover_f = nullptr, *d_cover_b = nullptr;
  int32_t *d_tmin_f = nullptr, *d_tmin_b = nullptr, *d_gmin_f = nullptr, *d_gmin_b = nullptr, *d_pval_p = nullptr,
          *d_pval_s = nullptr;
  uint32_t *d_ps0 = nullptr, *d_ps1 = nullptr, *d_pv0 = nullptr, *d_pv1 = nullptr, *d_bidx = nullptr;
  const unsigned sl_groups = cdiv(sl_tiles, kSlGroup);
  for (int pass = 0; pass < 2; pass++) {
    d_sym = ar.take<SymT>(n + 16);
    K0 = ar.take<uint64_t>(n + 2);  // (+2: the rank store's scratch list starts at a multiple of 4 entries behind hd)
    K1 = ar.take<uint64_t>(n + 2);
    DG0 = use_digit_bytes ? ar.take<uint8_t>(n + 64) : nullptr;
    DG1 = use_digit_bytes ? ar.take<uint8_t>(n + 64) : nullptr;
    d_claim = ar.take<uint32_t>(claim_size);
    d_claim_need = ar.take<uint32_t>(claim_size);
    d_gclaim = ar.take<uint32_t>(M + 1);
    d_gneed0 = ar.take<uint32_t>(n / 2 + 4);
    d_gneed1 = ar.take<uint32_t>(n / 2 + 4);
    V0 = ar.take<uint32_t>(n);
    V1 = ar.take<uint32_t>(n);
    AS0 = ar.take<uint32_t>(n);
    AS1 = ar.take<uint32_t>(n);
    AG = ar.take<uint32_t>(n);
    d_sa = (v->keep_debug || v->lcp_kasai || (text_only && !use_trie)) ? ar.take<uint32_t>(n) : nullptr;
    d_rng_lo = ar.take<uint32_t>(M + 1);
    d_rng_hi = ar.take<uint32_t>(M + 1);
    d_rng_long = ar.take<uint8_t>(M + 1);
    d_rank = ar.take<RankEntry>(n);
    AD0 = ar.take<uint32_t>(n);
    AD1 = ar.take<uint32_t>(n);
    d_tdep = ar.take<uint32_t>(n + 2 + rr_tiles * 4 + 8);  // (tail: look-back state of the fused rerank)
    d_gdepth = ar.take<uint32_t>(n);
    d_lcp = (text_only && !v->keep_debug) ? nullptr : ar.take<int32_t>(n);  // (text-only layout: nothing reads the LCPs)
    d_node_of_slot = use_trie ? ar.take<uint32_t>(n) : nullptr;  // end node of the suffix in a slot of a needed group (trie.h)
    d_vsym = use_trie ? ar.take<SymT>(hv.stream.size() + 16) : nullptr;
    d_child_sym = use_trie ? ar.take<uint32_t>(hv.lt_child_cp.size() + 1) : nullptr;
    d_radix_tmp = ar.take<uint32_t>(radix_words);
    d_ghead = ar.take<uint32_t>(n / 2 + 4);
    d_large_id = ar.take<uint32_t>(static_cast<size_t>(M) + 4);   // per needed group (at most one per long token): first slot,
    d_large_off = ar.take<uint32_t>(static_cast<size_t>(M) + 4);  // depth,
    d_gnode = ar.take<uint32_t>(static_cast<size_t>(M) + 4);      // trie node its members share and the symbols behind it (trie.h)
    d_gdone = ar.take<uint32_t>(static_cast<size_t>(M) + 4);
    d_lg_head = ar.take<uint32_t>(n / kLsMaxGroup + 4);
    d_lg_off = ar.take<uint32_t>(n / kLsMaxGroup + 4);
    LK1 = ar.take<uint64_t>(n);
    LV0 = ar.take<uint32_t>(n);
    LV1 = ar.take<uint32_t>(n);
    LPOS = ar.take<uint32_t>(n + 16);
    d_agg = ar.take<RerankAgg>(rr_tiles + 1);
    d_chunk_agg = ar.take<RerankAgg>(cdiv(rr_tiles, kRrChunk) + 1);
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
    d_tile_mlo = ar.take<uint32_t>(sl_tiles + 2);
    d_tmin_f = ar.take<int32_t>(sl_tiles + 1);
    d_tmin_b = ar.take<int32_t>(sl_tiles + 1);
    d_gmin_f = ar.take<int32_t>(sl_groups + 1);
    d_gmin_b = ar.take<int32_t>(sl_groups + 1);
    d_ps0 = ar.take<uint32_t>(P + 1);
    d_ps1 = ar.take<uint32_t>(P + 1);
    d_pv0 = ar.take<uint32_t>(P + 1);
    d_pv1 = ar.take<uint32_t>(P + 1);
    d_pval_p = ar.take<int32_t>(P + 1);
    d_pval_s = ar.take<int32_t>(P + 1);
    d_bidx = ar.take<uint32_t>(static_cast<size_t>(nbuckets) + 2);
    d_emit = ar.take<int32_t>(n_text + 1);
    d_anchors = ar.take<uint32_t>(n_text + 1);
    d_anchor_cnt = ar.take<uint32_t>(emit_tiles + 1);
    d_anchor_tmp = ar.take<uint32_t>(cdiv(emit_tiles, kScanTile) + 8);
    d_emit_cnt = ar.take<uint32_t>(emit_tiles + 1);
    d_emit_tmp = ar.take<uint32_t>(cdiv(walk_blocks, kScanTile) + 8);  // (walk_blocks >= emit_tiles)
    d_blk_cnt = ar.take<uint32_t>(walk_blocks + 2);
    d_blk_off = ar.take<uint32_t>(walk_blocks + 2);
    if (pass == 0) ar.commit();
  }
  ar.arm(st);

  // side stream: the anchor list and the cleared emit array only need the class bytes
  auto launch_anchors = [&](bool do_fork) {
    if (do_fork) fork();
    const unsigned atiles = cdiv(n_text, kAnchorTile);
    if (!staged_possible) WP_HIP(hipMemsetAsync(d_emit, 0x80, n_text * sizeof(int32_t), st2));
    hipLaunchKernelGGL(anchor_count_kernel, dim3(atiles), dim3(kBlock), 0, st2, d_cls,
                       static_cast<const uint8_t *>(nullptr), n_text, d_anchor_cnt);
    device_exclusive_scan(d_anchor_cnt, d_anchor_cnt, atiles, d_anchor_tmp, c->d_scalars + 10, st2);
    hipLaunchKernelGGL(anchor_write_kernel, dim3(atiles), dim3(kBlock), 0, st2, d_cls,
                       static_cast<const uint8_t *>(nullptr), n_text, d_anchor_cnt, d_anchors);
    hipLaunchKernelGGL(anchor_gap_kernel, dim3(std::min<size_t>(atiles, 1024)), dim3(kBlock), 0, st2, d_anchors,
                       c->d_scalars + 10, n_text, d_cls, hv.soft.empty() ? 1 : 0, c->d_scalars + 11);
  };
  // Where they run (WP_ANCHOR_AT): 0 = next to the key build and the host's code construction
  // (compute-bound / idle GPU), 1 = next to the round-0 split, 2 = next to the scanline stage (small
  // latency-bound kernels).  Not next to the radix passes, which want the bandwidth themselves.
  static const int anchor_at = getenv("WP_ANCHOR_AT") ? atoi(getenv("WP_ANCHOR_AT")) : 2;
  const bool anchors_late = anchor_at != 1;  // the counts are fetched right before the walk
  // ---------------- S build: dense symbols, symbol code, round-0 keys ----------------
  static const bool allow_variable = !(getenv("WP_FIXED_CODE") && atoi(getenv("WP_FIXED_CODE")) != 0);
  // alphabets > 255: the code covers symbol >> lo_bits (<= 256 values), the low bits follow verbatim
  const int lo_bits = sizeof(SymT) == 1 ? 0 : std::max(0, bits - 8);
  constexpr int kCodeReuse = 64;
  static const bool env_no_code_cache = env_flag("WP_NO_CODE_CACHE");
  const bool reuse_code = allow_variable && !env_no_code_cache && c->code_cached && c->code_alphabet == static_cast<uint32_t>(S.alphabet) &&
                          c->code_bits == bits && c->code_lo == lo_bits && c->code_uses < kCodeReuse;
  if (!reuse_code) WP_HIP(hipMemsetAsync(c->d_symhist, 0, sizeof(uint32_t) * 256, st));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(decode_write_kernel<SymT>), dim3(cdiv(nbytes, kDecTile)), dim3(kBlock), 0, st,
                     d_text, nbytes, d_tile_prefix, c->d_lut, d_sym, d_cls, d_cps, c->d_cls_bmp, c->d_soft,
                     static_cast<int>(hv.soft.size()), allow_variable && !reuse_code ? c->d_symhist : nullptr, lo_bits);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(map_vocab_symbols_kernel<SymT>), dim3(cdiv(n - n_text, kBlock)), dim3(kBlock), 0,
                     st, c->d_stream, n_text, n, c->d_lut, d_sym);
  if (n_text > 0 && anchor_at == 0) launch_anchors(true);
  SymbolCode code;
  if (reuse_code) {
    code = c->code_cache;  // (the device tables still hold it)
    c->code_uses++;
  } else if (allow_variable) {
    // frequencies of symbol >> lo_bits -> optimal order-preserving code (host, <= 256 items) -> device tables
    std::vector<uint32_t> h32(256);
    WP_HIP(hipMemcpyAsync(h32.data(), c->d_symhist, sizeof(uint32_t) * 256, hipMemcpyDeviceToHost, st));
    WP_HIP(hipStreamSynchronize(st));
    const size_t nitems = (static_cast<size_t>(S.alphabet) >> lo_bits) + 1;  // dense symbols 0..sigma
    std::vector<uint64_t> freq(nitems);
    for (size_t i = 0; i < nitems; i++) freq[i] = h32[i];
    code = build_symbol_code(freq, bits, true);
    if (!code.uniform_bits) {
      code.lo_bits = lo_bits;
      code.avg_bits += lo_bits;
    }
  } else {
    code = build_symbol_code({}, bits, false);
  }
  DevCode dcode{reinterpret_cast<const uint16_t *>(c->d_code), c->d_code + 512, c->d_code + 768,
                code.uniform_bits ? code.uniform_bits : -code.lo_bits};
  if (allow_variable && !reuse_code) {
    c->code_cache = code;
    c->code_cached = true;
    c->code_alphabet = static_cast<uint32_t>(S.alphabet);
    c->code_bits = bits;
    c->code_lo = lo_bits;
    c->code_uses = 0;
  } else if (!allow_variable) {
    c->code_cached = false;
  }
  if (!code.uniform_bits && !reuse_code) {
    const size_t blob_bytes = 512 + 256 + kDecodeTableBytes;
    std::memset(c->h_code, 0, blob_bytes);
    std::memcpy(c->h_code, code.cw.data(), code.cw.size() * sizeof(uint16_t));
    std::memcpy(c->h_code + 512, code.len.data(), code.len.size());
    std::memcpy(c->h_code + 768, code.bmask.data(), kDecodeTableBytes);
    WP_HIP(hipMemcpyAsync(c->d_code, c->h_code, blob_bytes, hipMemcpyHostToDevice, st));
  }
  S.symbols_per_key = static_cast<int32_t>(kKeyBits / std::max(1.0, code.avg_bits));
  {
    // 8-bit symbols with no codeword shorter than kKeys8MinLen bits (every ordinary text): the register form
    int min_len = code.uniform_bits ? code.uniform_bits : 99;
    for (uint8_t l : code.len) min_len = std::min<int>(min_len, l);
    if (sizeof(SymT) == 1 && sizeof(Key0) == 4 && min_len >= kKeys8MinLen && !env_flag("WP_KEYS_GENERIC")) {
      hipLaunchKernelGGL(build_keys0_u8_kernel, dim3(cdiv(n, kKeys8Tile)), dim3(kBlock), 0, st,
                         reinterpret_cast<const uint8_t *>(d_sym), n, dcode, reinterpret_cast<Key0 *>(K0), DG0);
    } else {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(build_keys0_kernel<SymT>), dim3(cdiv(n, kKeyTile)), dim3(kBlock), 0, st, d_sym,
                         n, dcode, reinterpret_cast<Key0 *>(K0), DG0);
    }
  }
  WP_LAUNCH_CHECK();
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[2], st));

  // ---------------- suffix array by prefix doubling ----------------
  // round 0 only keeps the tied groups that carry the key of a long eligible token (prune.h)
  static const bool env_no_prune = env_flag("WP_NO_PRUNE");
  const bool prune = !full && !env_no_prune && (M > 0 || text_only);
  const DepthRule rule{need_depth, full ? 1 : 0, nullptr, nullptr};
  // after every rerank: classify the new groups (large ones take the global path next round)
  // (runs on the side stream, next to the rank scatter)
  auto classify_groups = [&](size_t list_len) {
    if (list_len <= static_cast<size_t>(kLsMaxGroup)) return false;  // no group can be large
    const size_t cap = list_len / 2 + 1;  // a group has >= 2 entries
    WP_HIP(hipMemsetAsync(c->d_scalars + 6, 0, 2 * sizeof(uint32_t), st2));
    hipLaunchKernelGGL(large_groups_kernel, dim3(std::min<size_t>(cdiv(cap, kBlock), 2048)), dim3(kBlock), 0, st2, d_ghead,
                       c->d_scalars + 5, reinterpret_cast<unsigned long long *>(c->d_scalars + 6), d_lg_head, d_lg_off);
    hipLaunchKernelGGL(large_groups_close_kernel, dim3(1), dim3(1), 0, st2, c->d_scalars + 6, d_lg_off);
    return true;
  };
  // rank[dst[k]] = val[k].  Random 4-byte stores leave the L2s as partial lines; one radix pass over
  // the top 8 bits of the destination first, and an XCD-aware scatter after it, lets the stores of a
  // workgroup (and of its neighbours on the same XCD) fall into one ~1/256 window of the rank table
  // and merge in that XCD's L2 (measured: 2.1 ms -> 1.0 ms for 1e8 stores).  t_dst/t_val: scratch.
  static const int bin_bits = getenv("WP_BIN_BITS") ? atoi(getenv("WP_BIN_BITS")) : 8;
  auto store_ranks = [&](uint32_t *dst, uint32_t *val, uint32_t *t_dst, uint32_t *t_val, size_t m) {
    if (bin_bits > 0 && m >= (1u << 22)) {
      const int hb = bit_length(n - 1);
      // (the top bits of a text position are uniformly distributed: histogram by LDS atomics)
      const int bc = radix_sort_pairs<uint32_t>(dst, val, t_dst, t_val, m, std::max(0, hb - bin_bits), hb, d_radix_tmp,
                                                radix_words, st, nullptr, false, hb + 1, DigitBytes(),
                                                static_cast<const PlainVals *>(nullptr), true);
      hipLaunchKernelGGL(scatter_pairs_kernel, dim3(cdiv(m, kSpTile)), dim3(kBlock), 0, st, bc ? t_dst : dst,
                         bc ? t_val : val, m, d_rank, n, 1);
    } else {
      hipLaunchKernelGGL(scatter_pairs_kernel, dim3(cdiv(m, kSpTile)), dim3(kBlock), 0, st, dst, val, m, d_rank, n, 0);
    }
  };
  // Round 0 stores a rank for every position: a permutation.  The list is partitioned by ALL destination bits
  // above kWinBits (one or two radix passes over 8-byte records; their histograms read the digit bytes the
  // pass before left behind) and every 2^kWinBits-slot window of the rank table is then assembled in LDS and
  // written with full-width stores (window_store_kernel).  The passes go through scratch pairs (a: behind hd and
  // the sorted keys, which nobody reads after this; b: large-group buffers, idle in round 0) because vals and
  // hd are still needed.  dig: digit bytes of dst bits [kWinBits, kWinBits + 8), other: the second byte buffer.
  static const bool env_old_store = env_flag("WP_RANK_STORE_SCATTER");
  const int hb_n = bit_length(n - 1);
  const bool window_store = !env_old_store && bin_bits > 0 && n >= (1u << 22) && sizeof(Key0) == 4;
  // (more than 8 bits above the window: two passes of about half the bits each — 2^6 bins instead of 2^8 and
  // 2^4 for 1e8 positions: with uniform digits the runs a tile appends to its bins are 4096 / bins entries, and
  // 16-entry runs leave the workgroup as half lines: 0.51 ms for that pass against 0.34 ms)
  const int win_mid = hb_n - kWinBits <= kRadixBits ? hb_n : kWinBits + (hb_n - kWinBits + 1) / 2;
  // rank_in_pass != nullptr: the first pass computes the ranks itself from the sorted keys (suffix_array.h,
  // RankVals) — there is no rank array then (val == nullptr), and pair a must not be the key buffer.
  // before_second: called between the two passes (the side stream's searches in the keys must be over before pair
  // b, which may be the key buffer, is written).
  auto store_ranks_round0 = [&](uint32_t *dst, uint32_t *val, uint32_t *a_dst, uint32_t *a_val, uint32_t *b_dst,
                                uint32_t *b_val, uint8_t *dig, uint8_t *other, const RankVals *rank_in_pass,
                                const std::function<void()> &before_second) {
    // (a per-device attribute: set on every call, the context may live on any device)
    WP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(window_store_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kWinLdsBytes)));
    const int mid = win_mid;
    DigitBytes d1;
    d1.dg0 = dig;
    d1.dg1 = other;
    d1.dg0_ready = dig != nullptr;
    if (mid < hb_n) {  // (the first pass leaves the second pass's digits)
      d1.tail_bit = mid;
      d1.tail_mask = (1u << (hb_n - mid)) - 1u;
    }
    // (the roofline statistics describe the plain scatter: the instantiation that also computes the ranks is another kernel)
    // (val == nullptr without a rank source: the values are the slots themselves, made up by the pass.  The digits of
    // both passes are position bits, uniform over 64..128 values: LDS atomics into interleaved copies of the counters)
    radix_sort_pairs<uint32_t, RankVals>(dst, val, a_dst, a_val, n, kWinBits, mid, d_radix_tmp, radix_words, st,
                                         rank_in_pass ? nullptr : &c->rstats, val == nullptr && !rank_in_pass, 0, d1, rank_in_pass,
                                         true);
    S.rank_in_pass = rank_in_pass ? 1 : 0;
    if (before_second) before_second();
    const uint32_t *f_dst = a_dst, *f_val = a_val;
    if (mid < hb_n) {
      DigitBytes d2;
      d2.dg0 = dig ? d1.tail_out(1) : nullptr;
      d2.dg1 = dig ? dig : nullptr;
      d2.dg0_ready = dig != nullptr;
      radix_sort_pairs<uint32_t>(a_dst, a_val, b_dst, b_val, n, mid, hb_n, d_radix_tmp, radix_words, st, &c->rstats,
                                 false, 0, d2);
      f_dst = b_dst;
      f_val = b_val;
    }
    hipLaunchKernelGGL(window_store_kernel, dim3(cdiv(n, size_t(1) << kWinBits)), dim3(kWinThreads), kWinLdsBytes, st, f_dst,
                       f_val, n, d_rank);
  };
  // (the low bits of a round-0 key are the tail of a compressed codeword stream: near-uniform digits)
  DigitBytes db;
  db.dg0 = DG0;
  db.dg1 = DG1;
  db.dg0_ready = DG0 != nullptr;
  if (window_store) {  // the last pass leaves the first digit of the rank store's destination partition
    db.tail_bit = kWinBits;
    db.tail_mask = (1u << (win_mid - kWinBits)) - 1u;
    db.tail_from_val = true;
  }
  // histogram: one per-wave LDS counter per digit for the digits below this bit (near-uniform: the tail of a
  // compressed codeword stream), 8 interleaved copies of the counters above it (skewed digits; radix_sort.h)
  static const int hist_atomic_bits = getenv("WP_HIST_ATOMIC_BITS") ? atoi(getenv("WP_HIST_ATOMIC_BITS")) : 8;
  // (round-0 keys of up to 32 bits live as uint32 in the first half of the 64-bit key buffers)
  // (round-0 keys of up to 32 bits live as uint32 in the first half of the 64-bit key buffers)
  int cur = radix_sort_pairs<Key0>(reinterpret_cast<Key0 *>(K0), V0, reinterpret_cast<Key0 *>(K1), V1, n, 0, kKeyBits,
                                   d_radix_tmp, radix_words, st, &c->rstats, true, code.uniform_bits ? 0 : hist_atomic_bits,
                                   db, static_cast<const PlainVals *>(nullptr), true);
  Key0 *keys = reinterpret_cast<Key0 *>(cur ? K1 : K0);
  S.key_bits = kKeyBits;
  uint32_t *vals = cur ? V1 : V0, *other_vals = cur ? V0 : V1;
  uint32_t *slots = AS0, *other_slots = AS1;
  uint32_t *adep = AD0, *other_dep = AD1;
  bool classified = false;
  if (n_text > 0 && anchor_at == 1) launch_anchors(true);
  // group split of a round: count / spine / apply kernels, or (WP_OPT_FUSED_RERANK, env WP_RERANK=fused)
  // one kernel with a chained scan across tiles
  static const bool env_fused = getenv("WP_RERANK") && std::strcmp(getenv("WP_RERANK"), "fused") == 0;
  const bool fused_rerank = v->fused_rerank || env_fused;
  LookbackState lb;  // lives behind the tdep buffer
  {
    const unsigned tiles = cdiv(n, kRrTile);
    lb.wa = reinterpret_cast<unsigned long long *>(d_tdep + ((n + 1) & ~static_cast<size_t>(1)));
    lb.wb = lb.wa + tiles;
    lb.ticket = reinterpret_cast<uint32_t *>(lb.wb + tiles);
    const size_t lb_bytes = static_cast<size_t>(tiles) * 16 + 16;
    RankEntry *hd = reinterpret_cast<RankEntry *>(cur ? K0 : K1);
    // full-depth mode: the count / prefix / apply kernels take 64-bit keys (32-bit keys are widened into the
    // large-group key buffer, which is free during round 0)
    auto keys64 = [&]() -> const uint64_t * {
      if (sizeof(Key0) == 8) return reinterpret_cast<const uint64_t *>(keys);
      hipLaunchKernelGGL(widen_keys_kernel, dim3(std::min<size_t>(cdiv(n, kBlock), 8192)), dim3(kBlock), 0, st, keys, LK1, n);
      return LK1;
    };
    // text-only layout, full-size rank store: no rank kernel — the first partition pass of the rank store computes
    // the ranks and the depths of the tied groups from the sorted keys (suffix_array.h, RankVals)
    static const bool env_no_fusion = env_flag("WP_NO_RANK_FUSION");
    const bool fuse_rank = prune && window_store && !d_lcp && !v->keep_debug && !v->lcp_kasai && !env_no_fusion;
    if (prune) {
      // Depth-capped mode: the groups that have to go on are found from the vocabulary (prune.h) and appended
      // to the active list by the kernel that finds them — on the side stream (a few thousand waves of
      // binary searches), next to the streaming kernel that turns the sorted keys into ranks and LCPs.
      fork();
      WP_HIP(hipMemsetAsync(d_claim, 0xff, claim_size * sizeof(uint32_t), st2));
      NeededList nl{slots, other_vals, AG, adep, d_ghead, d_large_id, d_large_off,  // (the large-group tables are free until the classification)
                    reinterpret_cast<unsigned long long *>(c->d_scalars + 4),
                    text_only && !v->keep_debug ? d_sa : nullptr, need_depth, d_claim_need, d_gclaim, d_gneed0};
      WP_HIP(hipMemsetAsync(d_claim_need, 0, claim_size * sizeof(uint32_t), st2));
      if (use_trie) {  // the vocabulary stream and the trie's child labels as dense symbols of this encode's alphabet
        const size_t ns = hv.stream.size(), nc = hv.lt_child_cp.size();
        hipLaunchKernelGGL(HIP_KERNEL_NAME(trie_map_symbols_kernel<SymT>), dim3(cdiv(std::max<size_t>(std::max(ns, nc), 1), kBlock)),
                           dim3(kBlock), 0, st2, c->d_stream, ns, c->d_lt_child_cp, nc, c->d_lut, d_vsym, d_child_sym);
      }
      if (M > 0) {  // (no eligible token at all: every tied group retires)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(need_groups_kernel<SymT>), dim3(cdiv(static_cast<size_t>(M) * kWave, kBlock)),
                           dim3(kBlock), 0, st2, keys, vals, n, d_sym, c->d_stream, c->d_elig_start, c->d_elig_info, M,
                           c->d_lut, dcode, d_claim, static_cast<uint32_t>(claim_size - 1), nl,
                           text_only ? d_rng_lo : nullptr, d_rng_hi, d_rng_long);
      }
      hipLaunchKernelGGL(needed_list_close_kernel, dim3(1), dim3(1), 0, st2, c->d_scalars + 4, d_ghead);
      if (M > 0) {
        hipLaunchKernelGGL(needed_need_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st2, nl);  // (a needed group per token at most)
        hipLaunchKernelGGL(needed_fill_kernel, dim3(1024), dim3(kBlock), 0, st2, nl, vals, n);
        if (use_trie) {
          const TokenTrie tt{c->d_lt_chain_len, c->d_lt_chain_off, c->d_lt_child_begin, c->d_lt_child_node, d_child_sym};
          hipLaunchKernelGGL(HIP_KERNEL_NAME(trie_group_start_kernel<SymT>), dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st2, vals,
                             d_large_id, d_large_off, c->d_scalars + 4, d_sym, n, d_vsym, tt, d_gnode, d_gdone);
        }
      }
      if (fuse_rank) {
        // (the ranks are computed inside the first partition pass of the rank store, below)
      } else if (d_lcp) {
        hipLaunchKernelGGL(round0_rank_kernel<true>, dim3(cdiv(n, kR0Tile)), dim3(kBlock), 0, st, keys, vals, n,
                           dcode.first_len, dcode.uniform_bits, (v->keep_debug || v->lcp_kasai) ? d_sa : nullptr, hd, d_lcp,
                           d_gdepth);
      } else {
        hipLaunchKernelGGL(round0_rank_kernel<false>, dim3(cdiv(n, kR0Tile)), dim3(kBlock), 0, st, keys, vals, n,
                           dcode.first_len, dcode.uniform_bits, (v->keep_debug || v->lcp_kasai) ? d_sa : nullptr, hd, d_lcp,
                           d_gdepth);
      }
      if (!fuse_rank) join();  // the rank store below reuses the key buffer as scratch: the searches in it must be over
    } else if (fused_rerank) {
      WP_HIP(hipMemsetAsync(lb.wa, 0, lb_bytes, st));
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_fused_kernel<SymT, true>), dim3(tiles), dim3(kBlock), 0, st, keys64(),
                         vals, static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr), n, tiles,
                         lb, d_sym, static_cast<const RankEntry *>(nullptr), static_cast<const uint32_t *>(nullptr), n,
                         dcode.first_len, dcode.uniform_bits, rule, d_sa, hd, d_lcp, slots, other_vals, AG, adep,
                         d_ghead, d_gdepth, static_cast<uint32_t *>(nullptr), c->d_scalars + 4);
    } else {
      const uint64_t *k64 = keys64();
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_agg_kernel<true>), dim3(tiles), dim3(kBlock), 0, st, k64, vals, n,
                         static_cast<const uint32_t *>(nullptr), static_cast<const RankEntry *>(nullptr),
                         static_cast<const uint32_t *>(nullptr), n, dcode.first_len, dcode.uniform_bits, rule, d_tdep,
                         d_agg);
      hipLaunchKernelGGL(rerank_chunk_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, tiles,
                         d_chunk_agg);
      hipLaunchKernelGGL(rerank_prefix_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, d_chunk_agg,
                         tiles);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_apply_kernel<SymT, true>), dim3(tiles), dim3(kBlock), 0, st, k64,
                         vals, static_cast<const uint32_t *>(nullptr), static_cast<const uint32_t *>(nullptr), d_tdep,
                         n, d_agg, d_sym, n, dcode.first_len, dcode.uniform_bits, rule, d_sa, hd, d_lcp,
                         slots, other_vals, AG, adep, d_ghead, d_gdepth, c->d_scalars + 4);
    }
    fork();
    if (fuse_rank && use_trie) {
      // Trie refinement: nobody asks for a group's head or depth — the step functions change at boundaries between
      // distinct keys only (short tokens) or inside needed groups (long tokens, resolved by the trie round, which stores
      // every rank it touches), so a suffix's own slot serves as its rank: the first partition pass makes the slot
      // column up (the identity) instead of deriving group heads from the sorted keys.
      store_ranks_round0(vals, nullptr, LV0, LV1, reinterpret_cast<uint32_t *>(hd) + ((n + 3) & ~static_cast<size_t>(3)),
                         reinterpret_cast<uint32_t *>(keys), DG0 ? db.tail_out(cur) : nullptr,
                         DG0 ? db.tail_out(cur ^ 1) : nullptr, nullptr, [&] { join(); });
    } else if (fuse_rank) {
      // pair a = large-group buffers (idle in round 0), pair b = behind hd and the key buffer, which the side
      // stream's searches and the first pass itself still read until the join
      const RankVals rv{keys, dcode.first_len, dcode.uniform_bits, d_gdepth};
      store_ranks_round0(vals, nullptr, LV0, LV1, reinterpret_cast<uint32_t *>(hd) + ((n + 3) & ~static_cast<size_t>(3)),
                         reinterpret_cast<uint32_t *>(keys), DG0 ? db.tail_out(cur) : nullptr,
                         DG0 ? db.tail_out(cur ^ 1) : nullptr, &rv, [&] { join(); });
    } else if (window_store) {
      store_ranks_round0(vals, hd, reinterpret_cast<uint32_t *>(hd) + ((n + 3) & ~static_cast<size_t>(3)),
                         reinterpret_cast<uint32_t *>(keys), LV0, LV1, DG0 ? db.tail_out(cur) : nullptr,
                         DG0 ? db.tail_out(cur ^ 1) : nullptr, nullptr, nullptr);
    } else {
      store_ranks(vals, hd, reinterpret_cast<uint32_t *>(hd) + n, reinterpret_cast<uint32_t *>(keys), n);
    }
    WP_LAUNCH_CHECK();
    classified = classify_groups(n);
    join();
  }
  uint32_t *avals = other_vals;  // active list values live in the vals buffer the sort did not end in
  uint32_t *spare_vals = vals;
  // second keys of a round: 1 + rank (<= n), or — trie refinement — 1 + trie node
  const int rb = use_trie ? bit_length(hv.lt_chain_len.size() + 1) : bit_length(n);
  const TokenTrie trie{c->d_lt_chain_len, c->d_lt_chain_off, c->d_lt_child_begin, c->d_lt_child_node, d_child_sym};
  // Between two rounds the host needs the new list sizes (grids, large-group path).  The copy of the
  // scalars and the LDS segmented sort of the next round are queued first — the sort reads its sizes on
  // the device and gets a grid for the largest possible list — and only then does the host wait for the
  // copy: the round trip hides behind the sort instead of idling the GPU.
  auto next_round_begin = [&](size_t upper) {
    WP_HIP(hipMemcpyAsync(c->h_scalars, c->d_scalars, sizeof(uint32_t) * 12, hipMemcpyDeviceToHost, st));
    WP_HIP(hipEventRecord(c->evs[2], st));
    if (upper > 0 && use_trie) {  // every list entry walks the token trie: its end node is its second key (in adep)
      hipLaunchKernelGGL(HIP_KERNEL_NAME(trie_walk_kernel<SymT>), dim3(std::min<size_t>(cdiv(upper, kBlock), 16384)), dim3(kBlock), 0,
                         st, avals, AG, d_gnode, d_gdone, c->d_scalars + 4, d_sym, n, d_vsym, trie, adep);
    }
    fork();  // the large-group path of the next round (side stream) may start from here
    if (upper > 0) {
      hipLaunchKernelGGL(local_sort_kernel, dim3(cdiv(upper, kLsT)), dim3(kBlock), 0, st, avals, AG, adep,
                         c->d_scalars + 4, d_ghead, d_rank, n, rb, K0, spare_vals,
                         use_trie ? adep : static_cast<const uint32_t *>(nullptr));
    }
    WP_HIP(hipEventSynchronize(c->evs[2]));
  };
  next_round_begin(n);
  size_t n_anchors = n_text > 0 && !anchors_late ? c->h_scalars[10] : 0;  // (the side stream was joined above)
  size_t max_anchor_gap = n_text > 0 && !anchors_late ? c->h_scalars[11] : 0;
  size_t n_act = c->h_scalars[4], n_groups = c->h_scalars[5];
  size_t n_large_groups = classified ? c->h_scalars[6] : 0, n_large = classified ? c->h_scalars[7] : 0;
  static const bool group_stats = getenv("WP_GROUP_STATS") && atoi(getenv("WP_GROUP_STATS")) != 0;
  if (group_stats && n_groups > 0) {  // tuning aid: active entries by group size class after round 0
    unsigned long long *d_gs = reinterpret_cast<unsigned long long *>(d_tdep), h_gs[18];
    WP_HIP(hipMemsetAsync(d_gs, 0, sizeof(h_gs), st));
    hipLaunchKernelGGL(group_stats_kernel, dim3(cdiv(n_groups, kBlock)), dim3(kBlock), 0, st, d_ghead,
                       static_cast<uint32_t>(n_groups), d_gs);
    WP_HIP(hipMemcpyAsync(h_gs, d_gs, sizeof(h_gs), hipMemcpyDeviceToHost, st));
    WP_HIP(hipStreamSynchronize(st));
    static const char *cls[9] = {"2", "3-4", "5-8", "9-16", "17-32", "33-64", "65-256", "257-2048", ">2048"};
    std::cerr << "group stats after round 0: n_act=" << n_act << " groups=" << n_groups << "\n";
    for (int i = 0; i < 9; i++) std::cerr << "  size " << cls[i] << ": entries " << h_gs[i] << " groups " << h_gs[9 + i] << "\n";
  }
  int rounds = 1;
  S.active_per_round[0] = static_cast<int64_t>(n);
  // behind a pruned round 0 every group carries the depth its own tokens need (DepthRule, prune.h)
  static const bool env_global_need = env_flag("WP_GLOBAL_NEED");
  uint32_t *gneed_cur = d_gneed0, *gneed_nxt = d_gneed1;
  const bool group_need = prune && M > 0 && !env_global_need && !use_trie;
  while (n_act > 0) {
    DepthRule rrule = rule;
    if (use_trie) {  // one split by the trie nodes resolves every needed group
      rrule.final_round = 1;
      rrule.second_out = d_node_of_slot;
    } else if (group_need) {
      rrule.gneed_in = gneed_cur;
      rrule.gneed_out = gneed_nxt;
      std::swap(gneed_cur, gneed_nxt);
    }
    // A round adds to a group's depth the depth of the group its second keys point into: that doubles the
    // depth while those groups are refined too (full depth: 31 rounds for 2^31 symbols), and adds at least the
    // depth of a round-0 group — one symbol or more — when they retired in round 0 (depth-capped mode: a
    // vocabulary of 512-symbol tokens takes ~60 rounds over its short list).  More rounds than that can only
    // mean corrupted ranks: stop instead of spinning.
    if (static_cast<uint64_t>(rounds) > 80 + (full ? 0ull : static_cast<uint64_t>(need_depth))) {
      throw HipError("prefix doubling did not converge (internal error)");
    }
    if (rounds < 40) S.active_per_round[rounds] = static_cast<int64_t>(n_act);
    // small groups: one LDS-resident segmented sort per window of the list (already queued by
    // next_round_begin: avals -> (K0, spare_vals))
    uint64_t *skeys = K0, *kfree = K1;
    uint32_t *svals = spare_vals, *nvals = avals;  // avals is free again once the sorts have consumed it
    if (n_large > 0) {  // large groups (side stream, disjoint list positions): extract, global radix sort on
                        // (dense large id, second key), write back
      const int lgb = bit_length(n_large_groups > 0 ? n_large_groups - 1 : 0);
      hipLaunchKernelGGL(large_extract_kernel, dim3(cdiv(cdiv(n_large, kLxSpan), kBlock / kWave)), dim3(kBlock), 0, st2,
                         avals, adep, d_lg_head, d_lg_off, static_cast<uint32_t>(n_large_groups), n_large, d_rank, n, rb,
                         K1, LV0, LPOS, use_trie ? adep : static_cast<const uint32_t *>(nullptr));
      const int lc = radix_sort_pairs<uint64_t>(K1, LV0, LK1, LV1, n_large, 0, rb + lgb, d_radix_tmp, radix_words, st2,
                                                nullptr);  // (the roofline statistics describe the round-0 sort only)
      hipLaunchKernelGGL(large_writeback_kernel, dim3(std::min<size_t>(cdiv(n_large, kBlock), 8192)), dim3(kBlock), 0,
                         st2, lc ? LK1 : K1, lc ? LV1 : LV0, LPOS, n_large, AG, rb, skeys, svals);
      join();
    }
    const unsigned tiles = cdiv(n_act, kRrTile);
    RankEntry *hd = reinterpret_cast<RankEntry *>(kfree);
    if (fused_rerank) {
      WP_HIP(hipMemsetAsync(lb.wa, 0, static_cast<size_t>(tiles) * 16 + 16, st));
      lb.wb = lb.wa + tiles;
      lb.ticket = reinterpret_cast<uint32_t *>(lb.wb + tiles);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_fused_kernel<SymT, false>), dim3(tiles), dim3(kBlock), 0, st, skeys,
                         svals, slots, adep, n_act, tiles, lb, d_sym, d_rank, d_gdepth, n, dcode.first_len,
                         dcode.uniform_bits, rrule, d_sa, hd, d_lcp, other_slots, nvals, AG, other_dep, d_ghead,
                         d_gdepth, d_tdep, c->d_scalars + 4);
      hipLaunchKernelGGL(gdepth_store_kernel, dim3(cdiv(n_act, kBlock)), dim3(kBlock), 0, st, d_tdep, slots, n_act,
                         d_gdepth);
    } else {
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_agg_kernel<false>), dim3(tiles), dim3(kBlock), 0, st, skeys, svals,
                         n_act, adep, d_rank, d_gdepth, n, dcode.first_len, dcode.uniform_bits, rrule, d_tdep, d_agg);
      hipLaunchKernelGGL(rerank_chunk_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, tiles,
                         d_chunk_agg);
      hipLaunchKernelGGL(rerank_prefix_kernel, dim3(cdiv(tiles, kRrChunk)), dim3(kBlock), 0, st, d_agg, d_chunk_agg,
                         tiles);
      hipLaunchKernelGGL(HIP_KERNEL_NAME(rerank_apply_kernel<SymT, false>), dim3(tiles), dim3(kBlock), 0, st, skeys,
                         svals, slots, adep, d_tdep, n_act, d_agg, d_sym, n, dcode.first_len,
                         dcode.uniform_bits, rrule, d_sa, hd, d_lcp, other_slots, nvals, AG, other_dep, d_ghead,
                         d_gdepth, c->d_scalars + 4);
    }
    fork();
    store_ranks(svals, hd, reinterpret_cast<uint32_t *>(hd) + n, reinterpret_cast<uint32_t *>(skeys), n_act);
    WP_LAUNCH_CHECK();
    if (use_trie) {  // (nothing stays on the list)
      join();
      rounds++;
      break;
    }
    classified = classify_groups(n_act);
    join();
    std::swap(slots, other_slots);
    std::swap(adep, other_dep);
    avals = nvals;
    spare_vals = svals;
    next_round_begin(n_act);  // (the next list is at most as long as this one)
    n_act = c->h_scalars[4];
    n_groups = c->h_scalars[5];
    n_large_groups = classified ? c->h_scalars[6] : 0;
    n_large = classified ? c->h_scalars[7] : 0;
    rounds++;
  }
  (void)n_groups;
  S.rounds = rounds;
  // every tie that is left shares at least this many symbols: need_depth for the groups that went through
  // the rounds, the shortest possible key (whole codewords in kKeyBits bits) for the groups round 0 let go
  {
    const int max_len = code.uniform_bits ? code.uniform_bits : kMaxCodeLen + code.lo_bits;
    const int32_t key_syms = std::max(1, kKeyBits / max_len);
    S.sorted_depth = full ? 0x7fffffff : (prune ? std::min<int32_t>(static_cast<int32_t>(need_depth), key_syms)
                                                 : static_cast<int32_t>(need_depth));
  }
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[3], st));

  if (v->lcp_kasai) {  // alternative LCP builder: chunked Kasai exactly as linear.cpp:18-70
    const size_t chunk = 64;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(kasai_kernel<SymT>), dim3(cdiv(cdiv(n, chunk), kBlock)), dim3(kBlock), 0, st,
                       d_sym, d_sa, d_rank, n, chunk, d_lcp);
    WP_LAUNCH_CHECK();
  }
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[4], st));

  // ---------------- who marks + scanlines ----------------
  // (the side stream may start now, but its launches are issued behind the first scanline kernels so
  // that the host does not keep the main stream waiting)
  if (n_text > 0 && anchor_at == 2) fork();
  StepTable steps{};
  MarkView mv{};
  // (step values carry the token length above the id where both fit: scanline.h)
  const int pack_steps = (hv.longest < kStepMaxLen && hv.tokens.size() < (size_t(1) << kStepIdBits) && !env_flag("WP_NO_STEP_PACK")) ? 1 : 0;
  {
    const size_t vocab_base = n_text + 1;
    uint32_t *mslot = d_mslot0, *midx = d_midx0;
    if (text_only) {
      // S = text . 1: the reach of every token is its range in the sorted keys (prune.h); long tokens are
      // narrowed inside their refined group.  The marks arrive sorted (tokens in lexicographic order).
      if (M > 0) {
        if (use_trie) {
          hipLaunchKernelGGL(trie_token_range_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, d_node_of_slot, c->d_elig_node,
                             c->d_elig_subtree, M, d_rng_lo, d_rng_hi, d_rng_long);
        } else {
          hipLaunchKernelGGL(HIP_KERNEL_NAME(long_token_range_kernel<SymT>), dim3(cdiv(static_cast<size_t>(M) * kWave, kBlock)),
                             dim3(kBlock), 0, st, d_sa,
                             d_sym, n, c->d_stream, c->d_elig_start, c->d_elig_info, M, c->d_lut, d_rng_lo, d_rng_hi,
                             d_rng_long);
        }
        hipLaunchKernelGGL(virtual_marks_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, d_rng_lo, d_rng_hi, M,
                           c->d_elig_id, c->d_elig_info, d_mslot0, d_mid, d_minfo, d_rf, d_rb);
      }
      mv = MarkView{mslot, d_mid, d_minfo, d_rf, d_rb, M, d_cover_f, d_cover_b};
    } else {
      if (M > 0) {
        hipLaunchKernelGGL(mark_slots_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, c->d_elig_start, M,
                           vocab_base, d_rank, d_mslot0, d_midx0);
        int mc = radix_sort_pairs<uint32_t>(d_mslot0, d_midx0, d_mslot1, d_midx1, M, 0, bit_length(n), d_radix_tmp,
                                            radix_words, st, nullptr);
        mslot = mc ? d_mslot1 : d_mslot0;
        midx = mc ? d_midx1 : d_midx0;
        hipLaunchKernelGGL(mark_gather_kernel, dim3(cdiv(M, kBlock)), dim3(kBlock), 0, st, midx, M, c->d_elig_id,
                           c->d_elig_info, d_mid, d_minfo);
      }
      hipLaunchKernelGGL(tile_mlo_kernel, dim3(cdiv(sl_tiles + 1, kBlock)), dim3(kBlock), 0, st, mslot, M, n, sl_tiles,
                         d_tile_mlo);
      hipLaunchKernelGGL(sl_summary_kernel, dim3(sl_tiles), dim3(kBlock), 0, st, d_lcp, n, mslot, d_minfo, d_tile_mlo,
                         d_tmin_f, d_tmin_b, d_rf, d_rb);
      hipLaunchKernelGGL(sl_group_min_kernel, dim3(cdiv(static_cast<size_t>(sl_groups) * kWave, kBlock)), dim3(kBlock),
                         0, st, d_tmin_f, d_tmin_b, sl_tiles, sl_groups, d_gmin_f, d_gmin_b);
      mv = MarkView{mslot, d_mid, d_minfo, d_rf, d_rb, M, d_cover_f, d_cover_b};
      if (M > 0) {
        hipLaunchKernelGGL(sl_reach_global_kernel, dim3(cdiv(static_cast<size_t>(M) * kWave, kBlock)), dim3(kBlock), 0,
                           st, d_lcp, n, sl_tiles, sl_groups, mslot, d_minfo, M, d_tmin_f, d_tmin_b, d_gmin_f, d_gmin_b,
                           d_rf, d_rb);
      }
    }
    if (M > 0) {
      hipLaunchKernelGGL(mark_cover_kernel, dim3(4), dim3(kCoverThreads), 0, st, d_minfo, d_rf, d_rb, M, d_cover_f,
                         d_cover_b);
    }
    if (n_text > 0 && anchor_at == 2) launch_anchors(false);
    hipLaunchKernelGGL(piece_starts_kernel, dim3(cdiv(std::max(M, 1), kBlock)), dim3(kBlock), 0, st, mv, n, d_ps0);
    const int pc = radix_sort_pairs<uint32_t>(d_ps0, d_pv0, d_ps1, d_pv1, P, 0, bit_length(n), d_radix_tmp, radix_words,
                                              st, nullptr);
    uint32_t *pstart = pc ? d_ps1 : d_ps0;
    hipLaunchKernelGGL(piece_values_kernel, dim3(cdiv(static_cast<size_t>(P) * kWave, kBlock)), dim3(kBlock), 0, st,
                       mv, pstart, P, d_pval_p, d_pval_s, pack_steps);
    hipLaunchKernelGGL(piece_bucket_kernel, dim3(cdiv(nbuckets + 1, kBlock)), dim3(kBlock), 0, st, pstart, P,
                       bucket_shift, nbuckets, d_bidx);
    WP_LAUNCH_CHECK();
    steps = StepTable{pstart, d_pval_p, d_pval_s, d_bidx, bucket_shift, pack_steps};
  }
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[5], st));

  // ---------------- greedy walk + id stream ----------------
  int32_t *d_ids = reinterpret_cast<int32_t *>(V1);
  size_t n_ids = 0;
  if (n_text > 0) {
    WalkArgs wa{d_cls, n_text, d_rank, steps, c->d_tok_len, hv.unk_id, d_emit, nullptr, nullptr, nullptr,
                hv.soft.empty() ? 1 : 0, static_cast<int32_t>(hv.tokens.size())};
    S.anchor_mode = 0;
    if (anchors_late) {
      join();
      fetch_scalars(c, 12);
      n_anchors = c->h_scalars[10];
      max_anchor_gap = c->h_scalars[11];
    }
    const bool all_hard = hv.soft.empty();
    bool staged = staged_possible && max_anchor_gap <= kMaxAnchorGap;
    if (staged_possible && !staged) WP_HIP(hipMemsetAsync(d_emit, 0x80, n_text * sizeof(int32_t), st));  // long words after all
    if (!v->cover_anchors && all_hard && max_anchor_gap > kMaxAnchorGap) {
      // words longer than a lane should walk (walk.h, "long words"): pointer doubling instead
      LongWord *d_lw = reinterpret_cast<LongWord *>(LPOS);
      const uint32_t lw_cap = static_cast<uint32_t>(n_text / kMaxAnchorGap + 2);
      uint32_t *d_lw_off = LPOS + 2 * static_cast<size_t>(lw_cap);
      uint32_t *d_lw_fail = d_lw_off + lw_cap + 1;
      hipLaunchKernelGGL(long_word_collect_kernel, dim3(std::min<size_t>(cdiv(std::max<size_t>(n_anchors, 1), kBlock), 2048)),
                         dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, n_text, d_cls, d_lw, lw_cap,
                         c->d_scalars + 12);
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
        const uint32_t total = static_cast<uint32_t>(total64);  // <= n_text < 2^31
        WP_HIP(hipMemcpyAsync(d_lw_off, h_off.data(), sizeof(uint32_t) * (nw + 1), hipMemcpyHostToDevice, st));
        WP_HIP(hipMemsetAsync(d_lw_fail, 0, sizeof(uint32_t) * nw, st));
        int32_t *d_lid = reinterpret_cast<int32_t *>(LV0);
        uint32_t *jump_a = LV1, *jump_b = reinterpret_cast<uint32_t *>(K0);
        uint8_t *d_mark = reinterpret_cast<uint8_t *>(K1);
        const dim3 grid(cdiv(total, kBlock));
        hipLaunchKernelGGL(long_word_next_kernel, grid, dim3(kBlock), 0, st, wa, d_lw, d_lw_off, nw, total, d_lid, jump_a,
                           d_mark);
        WP_HIP(hipStreamSynchronize(st));  // h_off is a stack-owned upload source
        for (uint32_t reach = 1; reach < longest; reach *= 2) {  // after r rounds: chain prefixes of length 2^r
          hipLaunchKernelGGL(long_word_mark_kernel, grid, dim3(kBlock), 0, st, jump_a, total, d_mark);
          hipLaunchKernelGGL(long_word_double_kernel, grid, dim3(kBlock), 0, st, jump_a, total, jump_b);
          std::swap(jump_a, jump_b);
        }
        hipLaunchKernelGGL(long_word_mark_kernel, grid, dim3(kBlock), 0, st, jump_a, total, d_mark);
        hipLaunchKernelGGL(long_word_fail_kernel, grid, dim3(kBlock), 0, st, d_lid, d_mark, d_lw_off, nw, total, d_lw_fail);
        hipLaunchKernelGGL(long_word_emit_kernel, grid, dim3(kBlock), 0, st, wa, d_lw, d_lw_off, nw, total, d_lid, d_mark,
                           d_lw_fail);
        WP_LAUNCH_CHECK();
        S.anchor_mode = 2;
      }
    } else if (v->cover_anchors || max_anchor_gap > kMaxAnchorGap) {
      // long stretches without class-rule anchors ("soft" spacing chars): anchors from the matches
      // themselves (walk.h).  The large-group buffers of the suffix sort are free by now.
      uint32_t *d_reach = LV0, *d_reach_tiles = LPOS;
      uint8_t *d_aflags = reinterpret_cast<uint8_t *>(LV1);
      const unsigned rtiles = cdiv(n_text, kReachTile), atiles = cdiv(n_text, kAnchorTile);
      uint32_t *d_wp_tiles = d_reach_tiles + rtiles + 1;  // first word-prefix position at or behind each tile
      uint32_t *d_ns_tiles = d_wp_tiles + rtiles + 1;  // same for non-space positions
      uint32_t *d_gap_a = d_ns_tiles + rtiles + 1, *d_gap_b = d_gap_a + rtiles + 1;  // where the coverage rule applies (walk.h)
      // (the class-rule anchor list is still in d_anchors: the coverage rule is only needed inside its long gaps)
      hipLaunchKernelGGL(gap_tiles_kernel, dim3(cdiv(rtiles, kBlock)), dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, n_text,
                         rtiles, v->cover_anchors ? 1 : 0, d_gap_a, d_gap_b);
      hipLaunchKernelGGL(reach_kernel, dim3(rtiles), dim3(kBlock), 0, st, wa, d_reach, d_reach_tiles, d_gap_a, d_gap_b);
      hipLaunchKernelGGL(reach_spine_kernel, dim3(1), dim3(1024), 0, st, d_reach_tiles, static_cast<size_t>(rtiles));
      hipLaunchKernelGGL(cover_flags_kernel, dim3(rtiles), dim3(kBlock), 0, st, d_cls, d_reach, d_reach_tiles, n_text,
                         d_aflags, d_wp_tiles, d_ns_tiles, d_gap_a, d_gap_b);
      hipLaunchKernelGGL(suffix_min_kernel, dim3(1), dim3(1024), 0, st, d_wp_tiles, static_cast<size_t>(rtiles));
      hipLaunchKernelGGL(suffix_min_kernel, dim3(1), dim3(1024), 0, st, d_ns_tiles, static_cast<size_t>(rtiles));
      hipLaunchKernelGGL(anchor_count_kernel, dim3(atiles), dim3(kBlock), 0, st, d_cls, d_aflags, n_text,
                         d_anchor_cnt);
      device_exclusive_scan(d_anchor_cnt, d_anchor_cnt, atiles, d_anchor_tmp, c->d_scalars + 10, st);
      hipLaunchKernelGGL(anchor_write_kernel, dim3(atiles), dim3(kBlock), 0, st, d_cls, d_aflags, n_text,
                         d_anchor_cnt, d_anchors);
      WP_LAUNCH_CHECK();
      fetch_scalars(c, 11);
      n_anchors = c->h_scalars[10];
      wa.aflags = d_aflags;
      wa.wp_from_tile = d_wp_tiles;
      wa.ns_from_tile = d_ns_tiles;
      S.anchor_mode = 1;
      // coverage anchors: every id still comes from the lanes of the walk kernel, each inside its own stretch
      // [anchor, next anchor) — the id lists work as they do for the class rule (the cleared emit array is not used)
      staged = !env_sparse_emit && !v->sparse_emit;
    }
    S.n_anchors = static_cast<int64_t>(n_anchors);
    // the anchor list and the cleared emit array were produced on the side stream; one lane per anchor
    // (a grid sized for the worst case, every position an anchor, costs 0.35 ms of empty workgroups)
    const size_t acap = std::max<size_t>(n_anchors, 1);
    const unsigned wblocks = cdiv(acap, kBlock);
    if (staged) {
      int32_t *d_ctmp = reinterpret_cast<int32_t *>(K0);  // (the key buffers are free after the suffix sort)
      const int words = kWbWords;
      const unsigned sblocks = cdiv(acap, static_cast<size_t>(words));
      // stretches of more than kWideMin positions (class rule, hard spacing chars only: one word each) go to a whole
      // wave each first (walk.h, wide walk); the large-group buffers of the suffix sort are free by now
      const uint32_t *d_wide_cnt = nullptr;
      if (S.anchor_mode == 0 && all_hard && !wa.aflags && max_anchor_gap > kWideMin && !env_flag("WP_NO_WIDE_WALK")) {
        uint32_t *d_wide_list = LV0, *d_wcnt = LV1;
        WP_HIP(hipMemsetAsync(c->d_scalars + 13, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(wide_collect_kernel, dim3(std::min<size_t>(cdiv(acap, kBlock), 2048)), dim3(kBlock), 0, st, d_anchors,
                           c->d_scalars + 10, n_text, d_wide_list, c->d_scalars + 13);
        hipLaunchKernelGGL(walk_wide_kernel, dim3(std::min<size_t>(cdiv(acap, kBlock / kWave), 8192)), dim3(kBlock), 0, st, wa,
                           d_anchors, c->d_scalars + 10, d_wide_list, c->d_scalars + 13, d_wcnt);
        d_wide_cnt = d_wcnt;
      }
      if (d_wide_cnt) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(walk_balanced_kernel<WalkArgs, LinearStep, true>), dim3(sblocks), dim3(kBlock), 0, st, wa,
                           d_anchors, c->d_scalars + 10, acap, d_ctmp, d_blk_cnt, d_wide_cnt);
      } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(walk_balanced_kernel<WalkArgs, LinearStep, false>), dim3(sblocks), dim3(kBlock), 0, st, wa,
                           d_anchors, c->d_scalars + 10, acap, d_ctmp, d_blk_cnt, d_wide_cnt);
      }
      device_exclusive_scan(d_blk_cnt, d_blk_off, sblocks, d_emit_tmp, c->d_scalars + 9, st);
      hipLaunchKernelGGL(emit_gather_kernel, dim3(sblocks), dim3(kBlock), 0, st, d_anchors, c->d_scalars + 10, acap, d_ctmp,
                         d_blk_cnt, d_blk_off, d_ids, words);
    } else {
      hipLaunchKernelGGL(walk_kernel, dim3(wblocks), dim3(kBlock), 0, st, wa, d_anchors, c->d_scalars + 10, acap);
      const unsigned tiles = cdiv(n_text, kScanTile);
      hipLaunchKernelGGL(emit_count_kernel, dim3(tiles), dim3(kBlock), 0, st, d_emit, n_text, d_emit_cnt);
      device_exclusive_scan(d_emit_cnt, d_emit_cnt, tiles, d_emit_tmp, c->d_scalars + 9, st);
      hipLaunchKernelGGL(emit_write_kernel, dim3(tiles), dim3(kBlock), 0, st, d_emit, n_text, d_emit_cnt, d_ids);
    }
    S.staged_emit = staged ? 1 : 0;
    WP_LAUNCH_CHECK();
  }
  if (v->stage_timing) WP_HIP(hipEventRecord(c->ev[6], st));
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
      throw HipError("debug bounds: out-of-range addresses skipped: radix scatter " + std::to_string(oob[0]) +
                     ", rank store " + std::to_string(oob[1]) + ", token id " + std::to_string(oob[2]) + ", list slot " +
                     std::to_string(oob[3]));
    }
    S.reserved0 = 1;  // this is the bounds-checking build
  }
#endif
  fetch_scalars(c, 20);
  n_ids = n_text > 0 ? c->h_scalars[9] : 0;
  S.needed_after_round0 = prune ? (rounds > 1 ? S.active_per_round[1] : 0) : -1;

  S.n_ids = static_cast<int64_t>(n_ids);
  S.radix_passes = c->rstats.passes;
  S.radix_pass_elems = c->rstats.elems;
  S.radix_digit_bytes = c->rstats.digit_bytes;
  S.radix_pass_bytes = c->rstats.bytes;
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
  c->dbg.best_scratch = reinterpret_cast<int32_t *>(K0);  // K0 is free after the suffix sort
  c->dbg.cps = d_cps;
  c->dbg.n = n;
  c->dbg.n_text = n_text;
  *n_ids_out = n_ids;
}

// ---------------- word_piece::fast on the device (fast.h) ----------------
// decode -> code points + class bytes -> anchors -> trie walk per word -> id stream.
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
  static const bool env_guard = env_flag("WP_ARENA_GUARD");
  Arena aa(&c->a_buf, v->arena_guard || env_guard);
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
  static const bool env_sparse_emit = env_flag("WP_SPARSE_EMIT");
  const bool staged_possible = !env_sparse_emit && !v->sparse_emit;
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
    hipLaunchKernelGGL(HIP_KERNEL_NAME(walk_balanced_kernel<FastArgs, FastStep>), dim3(sblocks), dim3(kBlock), 0, st, fa,
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

wp_vocab::~wp_vocab() {
  park_context(std::move(ctx));
  for (auto &c : multi) park_context(std::move(c));
}

// ======================================================================================
// C ABI
// ======================================================================================
// Nothing leaves the C ABI as an exception, and the caller's current HIP device is the same after the call as before.
template <typename F>
static int guarded(F &&f) {
  DeviceGuard keep_device;
  try {
    f();
    return WP_OK;
  } catch (const HipError &e) {
    g_last_error = e.what();
    return std::string(e.what()).find("no HIP device") != std::string::npos ? WP_ERR_NO_DEVICE : WP_ERR_HIP;
  } catch (const std::length_error &e) {
    g_last_error = e.what();
    return WP_ERR_TOO_LARGE;
  } catch (const std::invalid_argument &e) {
    g_last_error = e.what();
    return WP_ERR_ARG;
  } catch (const std::ios_base::failure &e) {
    g_last_error = e.what();
    return WP_ERR_IO;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return WP_ERR_HIP;
  } catch (...) {
    g_last_error = "unknown exception inside the HIP WordPiece library";
    return WP_ERR_HIP;
  }
}

static int vocab_from_lines(const std::vector<std::pair<const char *, size_t>> &lines, wp_vocab **out) {
  if (!out) {
    g_last_error = "null output pointer";
    return WP_ERR_ARG;
  }
  try {
  std::unique_ptr<wp_vocab> v(new wp_vocab());
  if (const char *e = getenv("WP_DEVICES")) {  // default of WP_OPT_DEVICES ("all" or a count): the C++ API has no handle to set it on
    v->n_devices = std::strcmp(e, "all") == 0 ? -1 : std::max(1, atoi(e));
  }
  std::string err = v->hv.build(lines);
  if (!err.empty()) {
    g_last_error = err;
    return WP_ERR_EMPTY_WORD;
  }
  *out = v.release();
  return WP_OK;
  } catch (const std::exception &e) {
    g_last_error = e.what();
    return WP_ERR_ARG;
  } catch (...) {
    g_last_error = "unknown exception while building the vocabulary";
    return WP_ERR_ARG;
  }
}

extern "C" {

int wp_vocab_create(const char *const *lines, const size_t *line_bytes, size_t n_lines, wp_vocab **out) {
  std::vector<std::pair<const char *, size_t>> ls;
  ls.reserve(n_lines);
  for (size_t i = 0; i < n_lines; i++) ls.emplace_back(lines[i], line_bytes[i]);
  return vocab_from_lines(ls, out);
}

int wp_vocab_create_packed(const char *buf, const int64_t *offsets, int64_t n_lines, wp_vocab **out) {
  std::vector<std::pair<const char *, size_t>> ls;
  ls.reserve(static_cast<size_t>(n_lines));
  for (int64_t i = 0; i < n_lines; i++) ls.emplace_back(buf + offsets[i], static_cast<size_t>(offsets[i + 1] - offsets[i]));
  return vocab_from_lines(ls, out);
}

int wp_vocab_from_file(const char *vocab_file, wp_vocab **out) {
  // utils.cpp:123-137: a missing file yields an empty vocabulary (ifstream fails silently)
  std::ifstream fin(vocab_file);
  std::vector<std::string> words;
  std::string w;
  while (std::getline(fin, w)) words.push_back(w);
  std::vector<std::pair<const char *, size_t>> ls;
  for (auto &s : words) ls.emplace_back(s.data(), s.size());
  return vocab_from_lines(ls, out);
}

void wp_vocab_destroy(wp_vocab *v) {
  if (!v) return;
  try {
    delete v;  // (parks the handle's contexts: park_context keeps the caller's current device)
  } catch (...) {
  }
}
int64_t wp_vocab_size(const wp_vocab *v) { return static_cast<int64_t>(v->hv.tokens.size()); }
int32_t wp_vocab_unk_id(const wp_vocab *v) { return v->hv.unk_id; }
int32_t wp_vocab_token_flags(const wp_vocab *v, int64_t i) {
  const HostToken &t = v->hv.tokens[static_cast<size_t>(i)];
  return (t.is_prefix ? 1 : 0) | (t.is_special ? 2 : 0) | (t.is_malformed ? 4 : 0);
}
int64_t wp_vocab_token_len(const wp_vocab *v, int64_t i) {
  return static_cast<int64_t>(v->hv.tokens[static_cast<size_t>(i)].word.size());
}

int wp_set_option(wp_vocab *v, int option, int64_t value) {
  switch (option) {
    case WP_OPT_FULL_DEPTH: v->full_depth = value != 0; return WP_OK;
    case WP_OPT_DEVICE:
      if (v->ctx) {
        g_last_error = "device already bound";
        return WP_ERR_ARG;
      }
      v->device = static_cast<int>(value);
      return WP_OK;
    case WP_OPT_KEEP_DEBUG: v->keep_debug = value != 0; return WP_OK;
    case WP_OPT_STAGE_TIMING: v->stage_timing = value != 0; return WP_OK;
    case WP_OPT_LCP_KASAI: v->lcp_kasai = value != 0; return WP_OK;
    case WP_OPT_FUSED_RERANK: v->fused_rerank = value != 0; return WP_OK;
    case WP_OPT_COVER_ANCHORS: v->cover_anchors = value != 0; return WP_OK;
    case WP_OPT_ARENA_GUARD: v->arena_guard = value != 0; return WP_OK;
    case WP_OPT_VOCAB_IN_S: v->vocab_in_s = value != 0; return WP_OK;
    case WP_OPT_SPARSE_EMIT: v->sparse_emit = value != 0; return WP_OK;
    case WP_OPT_DEVICES: v->n_devices = value < 0 ? -1 : static_cast<int>(std::max<int64_t>(value, 1)); return WP_OK;
  }
  g_last_error = "unknown option";
  return WP_ERR_ARG;
}

int wp_get_stats(const wp_vocab *v, wp_stats *out) {
  *out = v->stats;
  return WP_OK;
}

int wp_linear_encode_device(wp_vocab *v, const void *d_utf8, size_t nbytes, const int32_t **d_ids, size_t *n_ids) {
  return guarded([&] {
    if ((reinterpret_cast<uintptr_t>(d_utf8) & 3u) != 0) throw std::invalid_argument("device text must be 4-byte aligned");
    size_t n = 0;
    Context *c = get_context(v);
    encode_on_device(v, c, static_cast<const uint8_t *>(d_utf8), nbytes, &n, v->stats);
    v->stats.n_devices = 1;
    *d_ids = n ? c->d_ids : nullptr;
    *n_ids = n;
  });
}

}  // extern "C"

// ---- host buffers for the ids ------------------------------------------------------------------------
// The ids leave the device into page-locked host memory (a download into freshly malloc'd pages runs
// at 13-26 GB/s, into pinned memory at the link rate) and that very block is handed to the caller;
// wp_free() recognises it and puts it back into a small pool instead of unpinning it.
namespace {
struct PinnedPool {
  std::mutex mu;
  std::unordered_map<void *, size_t> owned;       // every live pinned block (handed out or pooled) -> capacity
  std::vector<std::pair<size_t, void *>> pooled;  // free blocks
  static constexpr size_t kMaxPooledBlocks = 4;
  static constexpr size_t kMaxPooledBytes = size_t(6) << 30;
  void *take(size_t bytes) {
    {
      std::lock_guard<std::mutex> g(mu);
      size_t best = pooled.size();
      for (size_t i = 0; i < pooled.size(); i++) {
        if (pooled[i].first >= bytes && (best == pooled.size() || pooled[i].first < pooled[best].first)) best = i;
      }
      if (best != pooled.size()) {
        void *p = pooled[best].second;
        pooled.erase(pooled.begin() + static_cast<long>(best));
        return p;
      }
    }
    void *p = nullptr;
    const size_t want = bytes + bytes / 8 + 4096;
    WP_HIP(hipHostMalloc(&p, want));
    std::lock_guard<std::mutex> g(mu);
    owned[p] = want;
    return p;
  }
  // true: p was one of ours (now pooled or released)
  bool give_back(void *p) {
    size_t cap = 0;
    {
      std::lock_guard<std::mutex> g(mu);
      auto it = owned.find(p);
      if (it == owned.end()) return false;
      cap = it->second;
      size_t bytes = cap;
      for (auto &b : pooled) bytes += b.first;
      if (pooled.size() < kMaxPooledBlocks && bytes <= kMaxPooledBytes) {
        pooled.emplace_back(cap, p);
        return true;
      }
      owned.erase(it);
    }
    (void)hipHostFree(p);
    return true;
  }
  void trim() {  // pooled (free) blocks go back to the driver
    std::vector<std::pair<size_t, void *>> drop;
    {
      std::lock_guard<std::mutex> g(mu);
      drop.swap(pooled);
      for (auto &b : drop) owned.erase(b.second);
    }
    for (auto &b : drop) (void)hipHostFree(b.second);
  }
};
PinnedPool &id_pool() {
  static PinnedPool *pool = new PinnedPool();  // never destroyed: blocks may outlive static destruction order
  return *pool;
}
struct PinnedBlock {  // returns the block to the pool unless release()d to the caller
  void *p = nullptr;
  explicit PinnedBlock(size_t bytes) : p(id_pool().take(bytes)) {}
  ~PinnedBlock() {
    if (p) id_pool().give_back(p);
  }
  void *release() {
    void *r = p;
    p = nullptr;
    return r;
  }
};

using wp_clock = std::chrono::steady_clock;
double ms_since(wp_clock::time_point t0) { return std::chrono::duration<double, std::milli>(wp_clock::now() - t0).count(); }

// uploads [utf8, utf8 + nbytes) into c's text buffer (padded as the decoder expects) on c's stream
void upload_text(Context *c, const char *utf8, size_t nbytes) {
  c->text_buf.ensure(nbytes + 64);
  WP_HIP(hipMemsetAsync(static_cast<char *>(c->text_buf.p) + (nbytes & ~static_cast<size_t>(15)), 0, 32, c->stream));
  WP_HIP(hipMemcpyAsync(c->text_buf.p, utf8, nbytes, hipMemcpyHostToDevice, c->stream));
}

bool ascii_space(uint8_t b) { return (b >= 0x09 && b <= 0x0d) || b == 0x20; }

// Cuts [0, nbytes) into `parts` ranges at ASCII whitespace (no word straddles two shards),
// balanced by code points rather than bytes: the cost of a shard follows its symbol count, and a
// mixed-script corpus has 1-3 bytes per code point depending on where one looks.  Code points are
// estimated from every 64th 4 KB page (lead bytes = bytes that are not 10xxxxxx).
std::vector<size_t> shard_cuts(const char *utf8, size_t nbytes, int parts) {
  std::vector<size_t> cuts(static_cast<size_t>(parts) + 1, nbytes);
  cuts[0] = 0;
  if (parts <= 1) return cuts;
  const uint8_t *b = reinterpret_cast<const uint8_t *>(utf8);
  const size_t blocks = std::min<size_t>(static_cast<size_t>(parts) * 256, std::max<size_t>(1, nbytes / 4096));
  const size_t blk = (nbytes + blocks - 1) / blocks;
  std::vector<double> cum(blocks + 1, 0.0);
  for (size_t i = 0; i < blocks; i++) {
    const size_t lo = i * blk, hi = std::min(nbytes, lo + blk);
    size_t leads = 0, seen = 0;
    for (size_t page = lo; page < hi; page += 64 * 4096) {
      const size_t e = std::min(hi, page + 4096);
      for (size_t q = page; q < e; q++) leads += (b[q] & 0xc0u) != 0x80u;
      seen += e - page;
    }
    const double density = seen ? static_cast<double>(leads) / static_cast<double>(seen) : 1.0;
    cum[i + 1] = cum[i] + density * static_cast<double>(hi > lo ? hi - lo : 0);
  }
  size_t i = 0;
  for (int r = 1; r < parts; r++) {
    const double want = cum[blocks] * r / parts;
    while (i + 1 < blocks && cum[i + 1] < want) i++;
    const double span = cum[i + 1] - cum[i];
    size_t pos = i * blk + (span > 0 ? static_cast<size_t>((want - cum[i]) / span * static_cast<double>(blk)) : 0);
    pos = std::max(pos, cuts[static_cast<size_t>(r) - 1]);
    while (pos < nbytes && !ascii_space(b[pos])) pos++;
    cuts[static_cast<size_t>(r)] = std::min(pos, nbytes);
  }
  return cuts;
}

// One shard per entry of `devices` (ordinals may repeat: several contexts on one GPU), one host thread
// per shard for upload + device path, then every shard's ids are downloaded straight to their place in
// one pinned host block (exact sizes, no padded gather).
void encode_multi(wp_vocab *v, const char *utf8, size_t nbytes, const std::vector<int> &devices_in, int32_t **ids,
                  size_t *n_ids) {
  // a vocabulary with whitespace inside a token can match across a cut (the reference's own chunking has the
  // same caveat): such a text stays in one piece on the first device
  const std::vector<int> devices = v->hv.space_in_token ? std::vector<int>(devices_in.begin(), devices_in.begin() + 1)
                                                        : devices_in;
  const int G = static_cast<int>(devices.size());
  const auto t_all = wp_clock::now();
  const std::vector<size_t> cuts = shard_cuts(utf8, nbytes, G);
  if (v->multi.size() < static_cast<size_t>(G)) v->multi.resize(static_cast<size_t>(G));
  for (int g = 0; g < G; g++) {  // (contexts are made on the calling thread: a failure here is a plain exception)
    if (cuts[static_cast<size_t>(g)] == cuts[static_cast<size_t>(g) + 1]) continue;  // an empty shard needs none
    Context *c = v->multi[static_cast<size_t>(g)].get();
    if (c && c->device != devices[static_cast<size_t>(g)]) park_context(std::move(v->multi[static_cast<size_t>(g)]));
    if (!v->multi[static_cast<size_t>(g)]) v->multi[static_cast<size_t>(g)] = make_context(v, devices[static_cast<size_t>(g)]);
  }
  std::vector<size_t> counts(static_cast<size_t>(G), 0);
  std::vector<wp_stats> stats(static_cast<size_t>(G));
  std::vector<std::string> errors(static_cast<size_t>(G));
  std::vector<int> codes(static_cast<size_t>(G), WP_OK);
  auto work = [&](int g) {
    Context *c = v->multi[static_cast<size_t>(g)].get();
    const size_t lo = cuts[static_cast<size_t>(g)], hi = cuts[static_cast<size_t>(g) + 1];
    codes[static_cast<size_t>(g)] = guarded([&] {
      WP_HIP(hipSetDevice(c->device));
      std::memset(&stats[static_cast<size_t>(g)], 0, sizeof(wp_stats));
      if (hi == lo) return;
      upload_text(c, utf8 + lo, hi - lo);
      encode_on_device(v, c, static_cast<const uint8_t *>(c->text_buf.p), hi - lo, &counts[static_cast<size_t>(g)],
                       stats[static_cast<size_t>(g)]);
    });
    if (codes[static_cast<size_t>(g)] != WP_OK) {
      try {
        errors[static_cast<size_t>(g)] = g_last_error;
      } catch (...) {  // (out of memory while copying the message: the code alone is reported)
      }
    }
  };
  {
    // Worker threads are joined on every way out of this scope (a std::thread that is still joinable when it is
    // destroyed ends the process: std::terminate), work() itself cannot throw (everything that can sits inside
    // guarded(), the error slots are sized up front), and a shard without bytes gets no thread at all.
    struct Joiner {
      std::vector<std::thread> threads;
      ~Joiner() {
        for (auto &t : threads) {
          if (t.joinable()) t.join();
        }
      }
    } pool;
    pool.threads.reserve(static_cast<size_t>(G));
    int own = -1;  // the first non-empty shard runs on the calling thread
    for (int g = 0; g < G; g++) {
      if (cuts[static_cast<size_t>(g)] == cuts[static_cast<size_t>(g) + 1]) {
        std::memset(&stats[static_cast<size_t>(g)], 0, sizeof(wp_stats));
        continue;
      }
      if (own < 0) {
        own = g;
        continue;
      }
      try {
        pool.threads.emplace_back(work, g);
      } catch (const std::exception &e) {  // (no thread to be had: the shard runs here, after the others were started)
        work(g);
      }
    }
    if (own >= 0) work(own);
  }
  for (int g = 0; g < G; g++) {
    if (codes[static_cast<size_t>(g)] == WP_OK) continue;
    const std::string msg = "shard " + std::to_string(g) + " (device " + std::to_string(devices[static_cast<size_t>(g)]) + "): " +
                            errors[static_cast<size_t>(g)];
    if (codes[static_cast<size_t>(g)] == WP_ERR_TOO_LARGE) throw std::length_error(errors[static_cast<size_t>(g)]);
    throw HipError(msg);
  }
  size_t total = 0;
  std::vector<size_t> offs(static_cast<size_t>(G), 0);
  for (int g = 0; g < G; g++) {
    offs[static_cast<size_t>(g)] = total;
    total += counts[static_cast<size_t>(g)];
  }
  const auto t_d2h = wp_clock::now();
  if (total) {
    PinnedBlock blk(total * sizeof(int32_t));
    int32_t *h = static_cast<int32_t *>(blk.p);
    for (int g = 0; g < G; g++) {  // all downloads in flight together, each on its own device's stream
      Context *c = v->multi[static_cast<size_t>(g)].get();
      if (!counts[static_cast<size_t>(g)]) continue;
      WP_HIP(hipSetDevice(c->device));
      WP_HIP(hipMemcpyAsync(h + offs[static_cast<size_t>(g)], c->d_ids, counts[static_cast<size_t>(g)] * sizeof(int32_t),
                            hipMemcpyDeviceToHost, c->stream));
    }
    for (int g = 0; g < G; g++) {
      Context *c = v->multi[static_cast<size_t>(g)].get();
      if (!counts[static_cast<size_t>(g)]) continue;
      WP_HIP(hipSetDevice(c->device));
      WP_HIP(hipStreamSynchronize(c->stream));
    }
    *ids = static_cast<int32_t *>(blk.release());
    *n_ids = total;
  }
  // statistics of the call: sums over the shards, the slowest shard's device times
  wp_stats &S = v->stats;
  S = stats[0];
  for (int g = 1; g < G; g++) {
    const wp_stats &T = stats[static_cast<size_t>(g)];
    S.n_bytes += T.n_bytes;
    S.n_text += T.n_text;
    S.n_total += T.n_total;
    S.n_ids += T.n_ids;
    S.n_anchors += T.n_anchors;
    S.alphabet = std::max(S.alphabet, T.alphabet);
    S.rounds = std::max(S.rounds, T.rounds);
    S.radix_passes += T.radix_passes;
    S.radix_pass_elems += T.radix_pass_elems;
    S.radix_digit_bytes += T.radix_digit_bytes;
    S.radix_pass_bytes += T.radix_pass_bytes;
    S.ms_total = std::max(S.ms_total, T.ms_total);
    S.ms_decode = std::max(S.ms_decode, T.ms_decode);
    S.ms_sa = std::max(S.ms_sa, T.ms_sa);
    S.ms_lcp = std::max(S.ms_lcp, T.ms_lcp);
    S.ms_scan = std::max(S.ms_scan, T.ms_scan);
    S.ms_walk = std::max(S.ms_walk, T.ms_walk);
  }
  S.n_devices = G;
  S.ms_d2h = ms_since(t_d2h);
  S.ms_host_total = ms_since(t_all);
}

std::vector<int> resolve_devices(const int *devices, int n_devices) {
  std::vector<int> out;
  if (devices && n_devices > 0) {
    out.assign(devices, devices + n_devices);
    return out;
  }
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count == 0) {
    throw HipError("no HIP device available: the Linear WordPiece path has no CPU fallback");
  }
  const int want = n_devices <= 0 ? count : std::min(n_devices, count);
  for (int d = 0; d < want; d++) out.push_back(d);
  return out;
}
}  // namespace

extern "C" {

int wp_linear_encode_multi(wp_vocab *v, const char *utf8, size_t nbytes, const int *devices, int n_devices,
                           int32_t **ids, size_t *n_ids) {
  return guarded([&] {
    *ids = nullptr;
    *n_ids = 0;
    if (nbytes == 0) return;  // linear.cpp:323-325
    encode_multi(v, utf8, nbytes, resolve_devices(devices, n_devices), ids, n_ids);
  });
}

int wp_linear_encode(wp_vocab *v, const char *utf8, size_t nbytes, int32_t **ids, size_t *n_ids) {
  return guarded([&] {
    *ids = nullptr;
    *n_ids = 0;
    if (nbytes == 0) return;  // linear.cpp:323-325: the vocab path is not touched
    // WP_OPT_DEVICES / env WP_DEVICES: shard over several GPUs (inputs too small to be worth it stay on one)
    if (v->n_devices != 1 && nbytes >= (size_t(1) << 22)) {
      std::vector<int> devs = resolve_devices(nullptr, v->n_devices);
      const size_t per = size_t(1) << 21;  // at least 2 MB per shard
      if (devs.size() > nbytes / per) devs.resize(std::max<size_t>(1, nbytes / per));
      if (devs.size() > 1) {
        encode_multi(v, utf8, nbytes, devs, ids, n_ids);
        return;
      }
    }
    const auto t_all = wp_clock::now();
    Context *c = get_context(v);
    auto t0 = wp_clock::now();
    upload_text(c, utf8, nbytes);
    if (v->stage_timing) WP_HIP(hipStreamSynchronize(c->stream));
    const double ms_h2d = ms_since(t0);
    size_t n = 0;
    encode_on_device(v, c, static_cast<const uint8_t *>(c->text_buf.p), nbytes, &n, v->stats);
    t0 = wp_clock::now();
    if (n) {
      PinnedBlock blk(n * sizeof(int32_t));
      WP_HIP(hipMemcpyAsync(blk.p, c->d_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      WP_HIP(hipStreamSynchronize(c->stream));
      *ids = static_cast<int32_t *>(blk.release());
      *n_ids = n;
    }
    v->stats.n_devices = 1;
    v->stats.ms_h2d = v->stage_timing ? ms_h2d : 0.0;
    v->stats.ms_d2h = ms_since(t0);
    v->stats.ms_host_total = ms_since(t_all);
  });
}

int wp_reserve(wp_vocab *v, size_t nbytes) {
  return guarded([&] {
    Context *c = get_context(v);
    // arenas as an encode of `nbytes` of text would size them (about 100 bytes per symbol, DESIGN.md section 3;
    // an estimate: an encode that needs more grows them as before)
    const size_t n = nbytes + 1 + v->hv.stream.size();
    c->text_buf.ensure(nbytes + 64);
    c->a_buf.ensure(nbytes + nbytes / 512 + (size_t(1) << 20) + (v->keep_debug ? 4 * nbytes : 0));
    c->b_buf.ensure(108 * n + (v->keep_debug ? 4 * n : 0) + (size_t(64) << 20));
    PinnedBlock warm(nbytes + (size_t(1) << 20));  // about a quarter of an id per byte, 4 bytes each
  });
}

// A sequence of shards through one handle as a pipeline: while shard i is on the GPU, shard i + 1 is uploaded (a
// helper thread, its own stream, the second text buffer) and the ids of shard i - 1 are downloaded (their own
// stream, out of a staging buffer — the next encode overwrites the arena the ids were produced in).  For one call
// nothing can overlap the sort; for a corpus that arrives as shards (the reference's own encodeExternal batches,
// linear.cpp:355-371; the per-GPU stream of a sharded run) host to host then costs what the device path costs.
int wp_linear_encode_batch(wp_vocab *v, const char *const *texts, const size_t *nbytes, size_t n_texts, int32_t **ids,
                           size_t *n_ids) {
  return guarded([&] {
    for (size_t i = 0; i < n_texts; i++) {
      ids[i] = nullptr;
      n_ids[i] = 0;
    }
    if (n_texts == 0) return;
    const auto t_all = wp_clock::now();
    Context *c = get_context(v);
    if (!c->up_stream) {
      WP_HIP(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
      WP_HIP(hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
      for (auto &e : c->pipe_ev) WP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    size_t longest = 0;
    for (size_t i = 0; i < n_texts; i++) longest = std::max(longest, nbytes[i]);
    c->text_buf.ensure(longest + 64);
    c->text_buf2.ensure(longest + 64);
    DeviceBuffer *tb[2] = {&c->text_buf, &c->text_buf2};
    const int device = c->device;
    // upload of shard i into its text buffer, on the upload stream, finished when the call returns
    auto upload = [&](size_t i) -> std::string {
      try {
        if (nbytes[i] == 0) return "";
        WP_HIP(hipSetDevice(device));
        char *dst = static_cast<char *>(tb[i & 1]->p);
        WP_HIP(hipMemsetAsync(dst + (nbytes[i] & ~static_cast<size_t>(15)), 0, 32, c->up_stream));
        WP_HIP(hipMemcpyAsync(dst, texts[i], nbytes[i], hipMemcpyHostToDevice, c->up_stream));
        WP_HIP(hipStreamSynchronize(c->up_stream));
        return "";
      } catch (const std::exception &e) {
        return e.what()[0] ? e.what() : "upload failed";
      }
    };
    std::string up_err = upload(0);
    if (!up_err.empty()) throw HipError(up_err);
    std::vector<PinnedBlock> blocks;  // (released to the caller only when every shard is through)
    blocks.reserve(n_texts);
    bool staged_pending[2] = {false, false};
    wp_stats total{};
    for (size_t i = 0; i < n_texts; i++) {
      std::future<std::string> next_up;
      if (i + 1 < n_texts) next_up = std::async(std::launch::async, upload, i + 1);
      struct Wait {  // the helper must be done with the text buffers before anything unwinds
        std::future<std::string> &f;
        ~Wait() {
          if (f.valid()) f.wait();
        }
      } wait_up{next_up};
      size_t n = 0;
      wp_stats st{};
      if (nbytes[i]) encode_on_device(v, c, static_cast<const uint8_t *>(tb[i & 1]->p), nbytes[i], &n, st);
      blocks.emplace_back(std::max<size_t>(n, 1) * sizeof(int32_t));
      if (n) {
        const int slot = static_cast<int>(i & 1);
        if (staged_pending[slot]) WP_HIP(hipEventSynchronize(c->pipe_ev[2 + slot]));  // the download of shard i - 2 has left the staging buffer
        c->ids_stage[slot].ensure(n * sizeof(int32_t));
        WP_HIP(hipMemcpyAsync(c->ids_stage[slot].p, c->d_ids, n * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
        WP_HIP(hipEventRecord(c->pipe_ev[slot], c->stream));
        WP_HIP(hipStreamWaitEvent(c->down_stream, c->pipe_ev[slot], 0));
        WP_HIP(hipMemcpyAsync(blocks.back().p, c->ids_stage[slot].p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->down_stream));
        WP_HIP(hipEventRecord(c->pipe_ev[2 + slot], c->down_stream));
        staged_pending[slot] = true;
        // (the next encode may overwrite the arena: the copy into the staging buffer is ordered in front of it on c->stream)
      }
      n_ids[i] = n;
      total.n_bytes += st.n_bytes;
      total.n_text += st.n_text;
      total.n_total += st.n_total;
      total.n_ids += st.n_ids;
      total.ms_total += st.ms_total;
      total.rounds = std::max(total.rounds, st.rounds);
      if (next_up.valid()) {
        up_err = next_up.get();
        if (!up_err.empty()) throw HipError(up_err);
      }
    }
    WP_HIP(hipStreamSynchronize(c->down_stream));
    for (size_t i = 0; i < n_texts; i++) {
      if (n_ids[i]) ids[i] = static_cast<int32_t *>(blocks[i].release());
    }
    v->stats = total;
    v->stats.n_devices = 1;
    v->stats.ms_host_total = ms_since(t_all);
  });
}

// Gives cached memory back to the driver: the device arenas of this handle's contexts (v may be NULL), the arenas
// of the parked contexts of destroyed handles, and the pooled pinned id blocks.  The next encode allocates again.
int wp_trim(wp_vocab *v) {
  return guarded([&] {
    auto drop = [](Context *c) {
      if (!c) return;
      WP_HIP(hipSetDevice(c->device));
      WP_HIP(hipStreamSynchronize(c->stream));
      WP_HIP(hipStreamSynchronize(c->stream2));
      release_arenas(c);
    };
    if (v) {
      drop(v->ctx.get());
      for (auto &c : v->multi) drop(c.get());
    }
    {
      std::lock_guard<std::mutex> g(g_pool_mu);
      for (auto &c : context_pool()) drop(c.get());
    }
    id_pool().trim();
  });
}

struct MappedFile {
  const char *data = nullptr;
  size_t size = 0;
  int fd = -1;
  explicit MappedFile(const char *path) {
    fd = ::open(path, O_RDONLY);
    if (fd < 0) throw std::ios_base::failure(std::string("cannot open ") + path);
    struct stat sb;
    if (fstat(fd, &sb) != 0) {
      ::close(fd);
      throw std::ios_base::failure(std::string("cannot stat ") + path);
    }
    size = static_cast<size_t>(sb.st_size);
    if (size) {
      void *p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
      if (p == MAP_FAILED) {
        ::close(fd);
        throw std::ios_base::failure(std::string("cannot mmap ") + path);
      }
      data = static_cast<const char *>(p);
    }
  }
  ~MappedFile() {
    if (data) munmap(const_cast<char *>(data), size);
    if (fd >= 0) ::close(fd);
  }
};

int wp_linear_encode_file(const char *text_file, const char *vocab_file, int32_t **ids, size_t *n_ids) {
  wp_vocab *v = nullptr;
  int rc = wp_vocab_from_file(vocab_file, &v);
  if (rc != WP_OK) return rc;
  std::unique_ptr<wp_vocab> guard(v);
  rc = guarded([&] {
    MappedFile mm(text_file);
    int r2 = wp_linear_encode(v, mm.data, mm.size, ids, n_ids);
    if (r2 != WP_OK) throw std::runtime_error(g_last_error);
  });
  return rc;
}

// Host side of encodeExternal (linear.cpp:343-374): same batch rule and file format as the reference.
// Per batch: upload, encode, format the ids as text on the device (format.h), download into one of
// two pinned buffers; a writer thread appends that buffer to the file while the next batch is on the
// GPU.
namespace {
struct PinnedText {
  char *p = nullptr;
  size_t cap = 0;
  void ensure(size_t bytes) {
    if (bytes <= cap) return;
    if (p) WP_HIP(hipHostFree(p));
    p = nullptr;
    cap = 0;
    const size_t want = bytes + bytes / 8 + (1 << 20);
    WP_HIP(hipHostMalloc(reinterpret_cast<void **>(&p), want));
    cap = want;
  }
  ~PinnedText() {
    if (p) (void)hipHostFree(p);
  }
};
}  // namespace

static int encode_external_impl(const char *text_file, const char *vocab_file, const char *out_file,
                                size_t max_batch, bool fast) {
  wp_vocab *v = nullptr;
  int rc = wp_vocab_from_file(vocab_file, &v);
  if (rc != WP_OK) return rc;
  std::unique_ptr<wp_vocab> guard(v);
  return guarded([&] {
    if (max_batch == 0) throw std::invalid_argument("memory_limit too small");
    MappedFile mm(text_file);
    const char *begin = mm.data;
    size_t size = mm.size;
    FILE *fout = std::fopen(out_file, "wb");
    if (!fout) throw std::ios_base::failure(std::string("cannot open ") + out_file);
    struct Closer {
      FILE *f;
      ~Closer() { std::fclose(f); }
    } closer{fout};
    PinnedText host_text[2];
    std::future<void> pending[2];
    struct Drain {  // a failing batch must not leave a writer thread behind
      std::future<void> *p;
      ~Drain() {
        for (int i = 0; i < 2; i++) {
          if (p[i].valid()) p[i].wait();
        }
      }
    } drain{pending};
    size_t batch_no = 0;
    while (size > 0) {
      size_t batch;
      if (size > max_batch) {  // linear.cpp:357-362: grow until the batch's last byte starts a space
        batch = max_batch;
        while (batch < size) {
          const uint8_t *p = reinterpret_cast<const uint8_t *>(begin + batch - 1);
          uint32_t cp = ((p[0] & 0xc0u) == 0x80u) ? kInvalidUnicode : decode_one(p, static_cast<int64_t>(size - batch));
          if (is_space(cp)) break;
          batch++;
        }
      } else {
        batch = size;
      }
      Context *c = get_context(v);
      hipStream_t st = c->stream;
      c->text_buf.ensure(batch + 64);
      WP_HIP(hipMemsetAsync(static_cast<char *>(c->text_buf.p) + (batch & ~static_cast<size_t>(15)), 0, 32, st));
      WP_HIP(hipMemcpyAsync(c->text_buf.p, begin, batch, hipMemcpyHostToDevice, st));
      size_t n = 0;
      if (fast) {
        encode_fast_on_device(v, c, static_cast<const uint8_t *>(c->text_buf.p), batch, &n, v->stats);
      } else {
        encode_on_device(v, c, static_cast<const uint8_t *>(c->text_buf.p), batch, &n, v->stats);
      }
      if (n > 0) {
        // utils.cpp:30-35 format ("<id> " per id) on the device: byte counts, 64-bit offsets, text
        const size_t tiles = cdiv(n, kFmtTile);
        const size_t head = (tiles * (sizeof(uint32_t) + sizeof(unsigned long long)) + 8 + 255) & ~static_cast<size_t>(255);
        c->fmt_buf.ensure(head + n * 7);  // typical: <= 6 digits + space; grown below if the ids are longer
        auto layout = [&](uint32_t *&tb, unsigned long long *&to, unsigned long long *&total, char *&text) {
          char *base = static_cast<char *>(c->fmt_buf.p);
          to = reinterpret_cast<unsigned long long *>(base);
          total = to + tiles;
          tb = reinterpret_cast<uint32_t *>(total + 1);
          text = base + head;
        };
        uint32_t *tb;
        unsigned long long *to, *total;
        char *d_out;
        layout(tb, to, total, d_out);
        hipLaunchKernelGGL(fmt_count_kernel, dim3(tiles), dim3(kBlock), 0, st, c->d_ids, n, tb);
        hipLaunchKernelGGL(fmt_offsets_kernel, dim3(1), dim3(1024), 0, st, tb, tiles, to, total);
        WP_LAUNCH_CHECK();
        unsigned long long nbytes_out = 0;
        WP_HIP(hipMemcpyAsync(&nbytes_out, total, sizeof(nbytes_out), hipMemcpyDeviceToHost, st));
        WP_HIP(hipStreamSynchronize(st));
        if (head + nbytes_out > c->fmt_buf.cap) {  // longer ids than assumed: regrow and redo the (cheap) counts
          c->fmt_buf.ensure(head + nbytes_out);
          layout(tb, to, total, d_out);
          hipLaunchKernelGGL(fmt_count_kernel, dim3(tiles), dim3(kBlock), 0, st, c->d_ids, n, tb);
          hipLaunchKernelGGL(fmt_offsets_kernel, dim3(1), dim3(1024), 0, st, tb, tiles, to, total);
        }
        hipLaunchKernelGGL(fmt_write_kernel, dim3(tiles), dim3(kBlock), 0, st, c->d_ids, n, to, d_out);
        WP_LAUNCH_CHECK();
        const int slot = static_cast<int>(batch_no & 1);
        if (pending[slot].valid()) pending[slot].get();  // the writer of batch_no - 2 is done with this buffer
        host_text[slot].ensure(nbytes_out);
        WP_HIP(hipMemcpyAsync(host_text[slot].p, d_out, nbytes_out, hipMemcpyDeviceToHost, st));
        WP_HIP(hipStreamSynchronize(st));
        if (batch_no > 0 && pending[slot ^ 1].valid()) pending[slot ^ 1].get();  // keep the file in batch order
        const char *src = host_text[slot].p;
        const size_t cnt = static_cast<size_t>(nbytes_out);
        pending[slot] = std::async(std::launch::async, [fout, src, cnt] {
          if (std::fwrite(src, 1, cnt, fout) != cnt) throw std::ios_base::failure("short write to the id file");
        });
        batch_no++;
      }
      begin += batch;
      size -= batch;
    }
    for (int i = 0; i < 2; i++) {
      if (pending[i].valid()) pending[i].get();
    }
  });
}

int wp_linear_encode_external(const char *text_file, const char *vocab_file, const char *out_file,
                              size_t memory_limit) {
  return encode_external_impl(text_file, vocab_file, out_file, memory_limit / 20, false);  // linear.cpp:349
}

// ---- word_piece::fast (fast.cpp:152-220) ----------------------------------------------------------------
int wp_fast_encode_device(wp_vocab *v, const void *d_utf8, size_t nbytes, const int32_t **d_ids, size_t *n_ids) {
  return guarded([&] {
    if ((reinterpret_cast<uintptr_t>(d_utf8) & 3u) != 0) throw std::invalid_argument("device text must be 4-byte aligned");
    size_t n = 0;
    Context *c = get_context(v);
    encode_fast_on_device(v, c, static_cast<const uint8_t *>(d_utf8), nbytes, &n, v->stats);
    *d_ids = n ? c->d_ids : nullptr;
    *n_ids = n;
  });
}

int wp_fast_encode(wp_vocab *v, const char *utf8, size_t nbytes, int32_t **ids, size_t *n_ids) {
  return guarded([&] {
    *ids = nullptr;
    *n_ids = 0;
    if (nbytes == 0) return;  // fast.cpp:154-156
    const auto t_all = wp_clock::now();
    Context *c = get_context(v);
    upload_text(c, utf8, nbytes);
    size_t n = 0;
    encode_fast_on_device(v, c, static_cast<const uint8_t *>(c->text_buf.p), nbytes, &n, v->stats);
    if (n) {
      PinnedBlock blk(n * sizeof(int32_t));
      WP_HIP(hipMemcpyAsync(blk.p, c->d_ids, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      WP_HIP(hipStreamSynchronize(c->stream));
      *ids = static_cast<int32_t *>(blk.release());
      *n_ids = n;
    }
    v->stats.ms_host_total = ms_since(t_all);
  });
}

int wp_fast_encode_file(const char *text_file, const char *vocab_file, int32_t **ids, size_t *n_ids) {
  wp_vocab *v = nullptr;
  int rc = wp_vocab_from_file(vocab_file, &v);
  if (rc != WP_OK) return rc;
  std::unique_ptr<wp_vocab> guard(v);
  return guarded([&] {
    MappedFile mm(text_file);
    if (wp_fast_encode(v, mm.data, mm.size, ids, n_ids) != WP_OK) throw std::runtime_error(g_last_error);
  });
}

int wp_fast_encode_external(const char *text_file, const char *vocab_file, const char *out_file, size_t memory_limit) {
  return encode_external_impl(text_file, vocab_file, out_file, memory_limit / 2, true);  // fast.cpp:195
}

// UTF-8 of the stored word of line i (without the "##" of continuation tokens: utils.cpp:83-85), for
// word_piece::fast::decode (fast.cpp:163-187).  Returns the byte length; copies at most `cap` bytes.
int64_t wp_vocab_token_utf8(const wp_vocab *v, int64_t i, char *buf, size_t cap) {
  if (i < 0 || static_cast<size_t>(i) >= v->hv.tokens.size()) return -1;
  std::string out;
  for (uint32_t cp : v->hv.tokens[static_cast<size_t>(i)].word) {  // utf8.cpp:98-121 utf8_to_chars
    if (cp < 0x80) {
      out.push_back(static_cast<char>(cp));
    } else if (cp < 0x800) {
      out.push_back(static_cast<char>(0xc0 | (cp >> 6)));
      out.push_back(static_cast<char>(0x80 | (cp & 0x3f)));
    } else if (cp < 0x10000) {
      out.push_back(static_cast<char>(0xe0 | (cp >> 12)));
      out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3f)));
      out.push_back(static_cast<char>(0x80 | (cp & 0x3f)));
    } else {
      out.push_back(static_cast<char>(0xf0 | (cp >> 18)));
      out.push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3f)));
      out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3f)));
      out.push_back(static_cast<char>(0x80 | (cp & 0x3f)));
    }
  }
  if (buf && cap) std::memcpy(buf, out.data(), std::min(cap, out.size()));
  return static_cast<int64_t>(out.size());
}

int wp_linear_debug_fetch(const wp_vocab *v, int which, int32_t *out, size_t capacity, size_t *n_out) {
  return guarded([&] {
    if (!v->ctx || v->ctx->dbg.n == 0) throw std::invalid_argument("no encode has run on this handle");
    Context *c = v->ctx.get();
    WP_HIP(hipSetDevice(c->device));
    const auto &d = c->dbg;
    const void *src = nullptr;
    size_t cnt = d.n;
    switch (which) {
      case 0: src = d.sym; break;
      case 1:
        if (!d.sa) throw std::invalid_argument("the suffix array is kept only with WP_OPT_KEEP_DEBUG");
        src = d.sa;
        break;
      case 2: src = d.rank; break;
      case 3: src = d.lcp; cnt = d.n - 1; break;
      case 4:
      case 5: {  // materialise the reference's per-slot arrays from the step functions
        hipLaunchKernelGGL(step_expand_kernel, dim3(cdiv(d.n, kBlock)), dim3(kBlock), 0, c->stream, d.steps, d.n,
                           d.best_scratch, d.best_scratch + d.n);
        WP_LAUNCH_CHECK();
        WP_HIP(hipStreamSynchronize(c->stream));
        src = which == 4 ? d.best_scratch : d.best_scratch + d.n;
        break;
      }
      case 6:
        if (!d.cps) throw std::invalid_argument("code points are kept only with WP_OPT_KEEP_DEBUG");
        src = d.cps;
        cnt = d.n_text;
        break;
      default: throw std::invalid_argument("unknown debug array");
    }
    if (cnt > capacity) throw std::invalid_argument("debug buffer too small");
    *n_out = cnt;
    if (cnt == 0) return;
    if (which == 0 && d.sym_bytes == 1) {
      std::vector<uint8_t> tmp(cnt);
      WP_HIP(hipMemcpy(tmp.data(), src, cnt, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < cnt; i++) out[i] = tmp[i];
    } else {
      WP_HIP(hipMemcpy(out, src, cnt * sizeof(int32_t), hipMemcpyDeviceToHost));
    }
  });
}

void wp_free(void *p) {
  if (!p) return;
  if (!id_pool().give_back(p)) std::free(p);
}
const char *wp_last_error(void) { return g_last_error.c_str(); }
int wp_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) return 0;
  return count;
}

}  // extern "C"
