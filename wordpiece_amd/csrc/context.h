// context.h — what a vocabulary handle owns on a device: streams, vocabulary tables, bump arenas, the pool of parked
// contexts (the C ABI itself is in encoder.hip; the device path in linear_path.h and fast_path.h).
#pragma once
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/wordpiece_amd.h"
#include "code.h"
#include "decode.h"
#include "radix_sort.h"
#include "scanline.h"
#include "suffix_array.h"
#include "vocab.h"

namespace wp {

static thread_local std::string g_last_error;

// The environment switches of the library, read once per process (every other choice is an option of the handle,
// include/wordpiece_amd.h): debugging aids and process-wide defaults of tested behaviours, nothing that tunes.
struct EnvOptions {
  bool arena_guard;   // WP_ARENA_GUARD=1: guard zones behind every arena allocation, for every handle
  bool vocab_in_s;    // WP_VOCAB_IN_S=1: the reference's S = text . 1 . vocab layout, for every handle
  bool sparse_emit;   // WP_SPARSE_EMIT=1: ids through the per-position array, for every handle
  bool no_pool;       // WP_NO_CONTEXT_POOL=1: destroyed handles do not park their contexts
  static bool flag(const char *name) {
    const char *e = getenv(name);
    return e && atoi(e) != 0;
  }
  static const EnvOptions &get() {
    static const EnvOptions o{flag("WP_ARENA_GUARD"), flag("WP_VOCAB_IN_S"), flag("WP_SPARSE_EMIT"), flag("WP_NO_CONTEXT_POOL")};
    return o;
  }
};

struct DeviceBuffer {
  void *p = nullptr;
  size_t cap = 0;
  // slack: room for a somewhat larger next text without another hipFree / hipMalloc
  void ensure(size_t bytes, bool slack = true) {
    if (bytes <= cap) return;
    if (p) WP_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    size_t want = slack ? bytes + bytes / 16 + (1 << 20) : bytes;
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
  size_t off = 0, planned = 0;
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
    // (the two rounds of take() must ask for the same sizes: a layout decided by a pointer that is null while planning
    // would hand out memory behind the buffer)
    if (off > planned) throw std::logic_error("arena: the allocation sequence differs from the planned one");
    return reinterpret_cast<T *>(static_cast<char *>(buf->p) + o);
  }
  void commit() {
    buf->ensure(off + (guard ? 8 * 512 : 0));  // (guard mode: room for the zone table behind the arena)
    planned = off;
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
  hipStream_t stream3 = nullptr;  // second side stream: the large-group path of the trie round beside its LDS sort
  hipEvent_t evs[7] = {};         // fork / join / scalars fetched / side stream done with the sorted keys / partition passes queued /
                                  // trie nodes known / large groups sorted
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
  size_t list_hint = 0;  // entries the needed list of the last encode held (sizes the list arenas of the next one)
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
  bool full_depth = false, keep_debug = false, stage_timing = false, lcp_kasai = false, cover_anchors = false;
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
  const bool owns = c->stream || c->stream2 || c->stream3 || c->d_used || c->d_lut || c->d_scan_tmp || c->d_scalars || c->d_code ||
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
  if (c->stream3) (void)hipStreamDestroy(c->stream3);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  c->stream = c->stream2 = c->stream3 = nullptr;
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
  const bool no_pool = EnvOptions::get().no_pool;
  DeviceGuard keep;
  (void)hipSetDevice(c->device);
  if (!no_pool && hipStreamSynchronize(c->stream) == hipSuccess && hipStreamSynchronize(c->stream2) == hipSuccess &&
      hipStreamSynchronize(c->stream3) == hipSuccess) {
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
  WP_HIP(hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));
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

}  // namespace wp
