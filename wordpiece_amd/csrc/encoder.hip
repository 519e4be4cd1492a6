// encoder.hip — the C ABI of the HIP Linear-WordPiece path (include/wordpiece_amd.h) and the one translation unit the
// device code is compiled in.  The device path itself: linear_path.h (word_piece::linear, stage by stage) and
// fast_path.h (word_piece::fast); what a handle owns on a device: context.h.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
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
#include "context.h"
#include "fast_path.h"
#include "format.h"
#include "linear_path.h"


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
    // arenas as an encode of `nbytes` of ASCII text would size them in the default layout (43
    // bytes per symbol + the refinement list for an eighth of the text; an estimate — an encode that needs more,
    // a larger alphabet or the reference layout, grows them as before)
    const size_t n = nbytes + 1 + v->hv.stream.size();
    const size_t per_symbol = (v->keep_debug || v->vocab_in_s || v->full_depth) ? 108 : 56;
    c->text_buf.ensure(nbytes + 64, false);
    c->a_buf.ensure(nbytes + nbytes / 512 + (size_t(1) << 20) + (v->keep_debug ? 4 * nbytes : 0), false);
    c->b_buf.ensure(per_symbol * n + (v->keep_debug ? 4 * n : 0) + (size_t(64) << 20), false);
    PinnedBlock warm(nbytes + (size_t(1) << 20));  // about a quarter of an id per byte, 4 bytes each
  });
}

// A sequence of shards through one handle as a pipeline: while shard i is on the GPU, shard i + 1 is uploaded (a
// helper thread, its own stream, the second text buffer) and the ids of shard i - 1 are downloaded (their own
// stream, out of a staging buffer — the next encode overwrites the arena the ids were produced in).  For one call
// nothing can overlap the sort; for a corpus that arrives as shards (the reference's own encodeExternal batches,
// linear.cpp:355-371; the per-GPU stream of a sharded run) host to host then costs what the device path costs.
//   next(i, &ptr, &len) -> false: no text i;   deliver(i, block, n): the ids of text i have arrived in `block`
//   (called in order, one text behind the encodes; a block that is not release()d goes back to the pinned pool)
}  // extern "C"

namespace {
template <typename Next, typename Deliver>
void encode_pipeline(wp_vocab *v, Next &&next, Deliver &&deliver) {
  const auto t_all = wp_clock::now();
  Context *c = get_context(v);
  if (!c->up_stream) {
    WP_HIP(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    WP_HIP(hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    for (auto &e : c->pipe_ev) WP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  DeviceBuffer *tb[2] = {&c->text_buf, &c->text_buf2};
  const int device = c->device;
  std::atomic<long long> up_us{0};
  // upload of a text into text buffer `slot`, on the upload stream, finished when the call returns
  auto upload = [&](const char *text, size_t len, int slot) -> std::string {
    try {
      if (len == 0) return "";
      const auto t_up = wp_clock::now();
      WP_HIP(hipSetDevice(device));
      char *dst = static_cast<char *>(tb[slot]->p);
      WP_HIP(hipMemsetAsync(dst + (len & ~static_cast<size_t>(15)), 0, 32, c->up_stream));
      WP_HIP(hipMemcpyAsync(dst, text, len, hipMemcpyHostToDevice, c->up_stream));
      WP_HIP(hipStreamSynchronize(c->up_stream));
      up_us += static_cast<long long>(ms_since(t_up) * 1e3);
      return "";
    } catch (const std::exception &e) {
      return e.what()[0] ? e.what() : "upload failed";
    }
  };
  wp_stats total{};
  const char *cur_text = nullptr, *next_text = nullptr;
  size_t cur_len = 0, next_len = 0;
  bool have = next(0, &cur_text, &cur_len);
  if (have) {
    tb[0]->ensure(cur_len + 64);
    const std::string err = upload(cur_text, cur_len, 0);
    if (!err.empty()) throw HipError(err);
  }
  std::unique_ptr<PinnedBlock> in_flight;  // ids of the text before the current one, on their way down
  size_t in_flight_n = 0, in_flight_i = 0;
  int in_flight_slot = 0;
  for (size_t i = 0; have; i++) {
    const int slot = static_cast<int>(i & 1);
    const bool more = next(i + 1, &next_text, &next_len);
    std::future<std::string> next_up;
    if (more) {
      tb[slot ^ 1]->ensure(next_len + 64);  // (the encode that read this buffer, of text i - 1, is over)
      next_up = std::async(std::launch::async, upload, next_text, next_len, slot ^ 1);
    }
    struct Wait {  // the helper must be done with the text buffers before anything unwinds
      std::future<std::string> &f;
      ~Wait() {
        if (f.valid()) f.wait();
      }
    } wait_up{next_up};
    size_t n = 0;
    wp_stats st{};
    if (cur_len) encode_on_device(v, c, static_cast<const uint8_t *>(tb[slot]->p), cur_len, &n, st);
    std::unique_ptr<PinnedBlock> blk(new PinnedBlock(std::max<size_t>(n, 1) * sizeof(int32_t)));
    if (n) {
      // (the staging buffer of this slot was last read by the download of text i - 2, which has been delivered)
      c->ids_stage[slot].ensure(n * sizeof(int32_t));
      WP_HIP(hipMemcpyAsync(c->ids_stage[slot].p, c->d_ids, n * sizeof(int32_t), hipMemcpyDeviceToDevice, c->stream));
      WP_HIP(hipEventRecord(c->pipe_ev[slot], c->stream));
      WP_HIP(hipStreamWaitEvent(c->down_stream, c->pipe_ev[slot], 0));
      WP_HIP(hipMemcpyAsync(blk->p, c->ids_stage[slot].p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->down_stream));
      WP_HIP(hipEventRecord(c->pipe_ev[2 + slot], c->down_stream));
      // (the next encode may overwrite the arena: the copy into the staging buffer is ordered in front of it on c->stream)
    }
    if (in_flight) {  // the text before this one: its download ran beside this encode
      if (in_flight_n) WP_HIP(hipEventSynchronize(c->pipe_ev[2 + in_flight_slot]));
      deliver(in_flight_i, *in_flight, in_flight_n);
    }
    in_flight = std::move(blk);
    in_flight_n = n;
    in_flight_i = i;
    in_flight_slot = slot;
    total.n_bytes += st.n_bytes;
    total.n_text += st.n_text;
    total.n_total += st.n_total;
    total.n_ids += st.n_ids;
    total.ms_total += st.ms_total;
    total.rounds = std::max(total.rounds, st.rounds);
    if (next_up.valid()) {
      const std::string err = next_up.get();
      if (!err.empty()) throw HipError(err);
    }
    have = more;
    cur_text = next_text;
    cur_len = next_len;
  }
  if (in_flight) {
    if (in_flight_n) WP_HIP(hipEventSynchronize(c->pipe_ev[2 + in_flight_slot]));
    deliver(in_flight_i, *in_flight, in_flight_n);
  }
  v->stats = total;
  v->stats.n_devices = 1;
  v->stats.ms_h2d = static_cast<double>(up_us.load()) / 1e3;  // wall time inside the uploads (beside the encodes, all but the first)
  v->stats.ms_host_total = ms_since(t_all);
}
}  // namespace

extern "C" {

int wp_linear_encode_batch(wp_vocab *v, const char *const *texts, const size_t *nbytes, size_t n_texts, int32_t **ids,
                           size_t *n_ids) {
  return guarded([&] {
    for (size_t i = 0; i < n_texts; i++) {
      ids[i] = nullptr;
      n_ids[i] = 0;
    }
    std::vector<void *> got(n_texts, nullptr);  // (handed to the caller only when every text is through)
    try {
      encode_pipeline(
          v,
          [&](size_t i, const char **t, size_t *len) {
            if (i >= n_texts) return false;
            *t = texts[i];
            *len = nbytes[i];
            return true;
          },
          [&](size_t i, PinnedBlock &blk, size_t n) {
            n_ids[i] = n;
            if (n) got[i] = blk.release();
          });
    } catch (...) {
      for (void *p : got) {
        if (p) id_pool().give_back(p);
      }
      for (size_t i = 0; i < n_texts; i++) n_ids[i] = 0;
      throw;
    }
    for (size_t i = 0; i < n_texts; i++) ids[i] = static_cast<int32_t *>(got[i]);
  });
}

int wp_linear_encode_stream(wp_vocab *v, wp_text_source next, wp_ids_sink out, void *user) {
  return guarded([&] {
    if (!next || !out) throw std::invalid_argument("wp_linear_encode_stream: null callback");
    encode_pipeline(
        v, [&](size_t i, const char **t, size_t *len) { return next(user, i, t, len) != 0; },
        [&](size_t i, PinnedBlock &blk, size_t n) { out(user, i, n ? static_cast<const int32_t *>(blk.p) : nullptr, n); });
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
