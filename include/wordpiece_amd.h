/*
 * wordpiece_amd.h — C ABI of the MI355X-native Linear WordPiece encoder.
 *
 * This is the drop-in boundary for the reference's Linear path
 * (gleb-kov/wordpiece, src/word_piece.hpp:12-19 → src/linear.cpp:321-374).  The
 * reference has no FFI of its own: its boundary is the C++ header
 * word_piece.hpp, re-implemented on top of this ABI in include/word_piece.hpp.
 * Every entry point below cites the reference interface it replaces.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; every function
 * returns WP_OK (0) or a WP_ERR_* code and wp_last_error() then holds the
 * message for the calling thread.  A vocab handle is single-caller (the
 * reference is effectively single-caller too: its global thread pool barrier
 * waits on all tasks, utils.cpp:25-28).  The library needs a HIP device: there
 * is NO CPU fallback — without a GPU every compute entry point fails loudly
 * with WP_ERR_NO_DEVICE.  No exception crosses the ABI, and every entry point leaves
 * the calling thread's current HIP device as it found it (hipGetDevice on entry,
 * hipSetDevice back on exit), whatever device the handle or its shards live on.
 */
#ifndef WORDPIECE_AMD_H
#define WORDPIECE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WP_OK 0
#define WP_ERR_EMPTY_WORD 1 /* "Vocab word is empty"   utils.cpp:99-101  */
#define WP_ERR_TOO_LARGE 2  /* "64bit not implemented" linear.cpp:104-106 */
#define WP_ERR_HIP 3        /* HIP runtime error (replaces "SACA return code: N", linear.cpp:139-141) */
#define WP_ERR_NO_DEVICE 4
#define WP_ERR_IO 5  /* file could not be opened / mapped (Boost throws there) */
#define WP_ERR_ARG 6

typedef struct wp_vocab wp_vocab;

/* ---- vocabulary: utils.cpp:81-137 (WordPieceToken, parseVocab, readVocabFromFile) ----
 * Lines are raw UTF-8 without terminators; `##` prefix, [special] and malformed
 * classification exactly as the reference.  The handle also caches the device
 * copy of the vocab symbol stream (a pure function of the vocab). */
int wp_vocab_create(const char *const *lines, const size_t *line_bytes, size_t n_lines,
                    wp_vocab **out);
/* same, lines given as one buffer + n_lines+1 offsets (ctypes/JNI friendly) */
int wp_vocab_create_packed(const char *buf, const int64_t *offsets, int64_t n_lines,
                           wp_vocab **out);
/* utils.cpp:123-137: std::getline semantics (LF only; a CR stays in the token) */
int wp_vocab_from_file(const char *vocab_file, wp_vocab **out);
void wp_vocab_destroy(wp_vocab *v);
int64_t wp_vocab_size(const wp_vocab *v);
int32_t wp_vocab_unk_id(const wp_vocab *v);              /* utils.hpp:30-33, -1 if absent */
int32_t wp_vocab_token_flags(const wp_vocab *v, int64_t i); /* bit0 prefix, bit1 special, bit2 malformed */
int64_t wp_vocab_token_len(const wp_vocab *v, int64_t i);

/* ---- the hot path: linear.cpp:321-328 encodeLinearWordPiece ----
 * replaces word_piece::linear::encode(text, vocab)        word_piece.hpp:12, linear.cpp:332-335
 * host UTF-8 in, malloc'd int32 ids out (free with wp_free). */
int wp_linear_encode(wp_vocab *v, const char *utf8, size_t nbytes, int32_t **ids, size_t *n_ids);

/* Same path with the text already resident in device memory and the ids left
 * there (what bench.py times; what a training input pipeline would consume).
 * `d_ids` is owned by the handle and valid until the next call on it.
 * `d_utf8` must be 4-byte aligned and readable up to the next multiple of 16 bytes behind
 * `nbytes` (the decoder loads whole words; the padding bytes are ignored). */
int wp_linear_encode_device(wp_vocab *v, const void *d_utf8, size_t nbytes,
                            const int32_t **d_ids, size_t *n_ids);

/* The same call sharded over several GPUs of the node, behind the boundary: the reference's own
 * precedent is the in-library chunking at whitespace of linear.cpp:283-299 (thread chunks) and
 * linear.cpp:355-367 (encodeExternal batches).  The text is cut at ASCII whitespace into one shard
 * per entry of `devices` (HIP ordinals; an ordinal may repeat), balanced by code points; every
 * shard is encoded on its device by its own host thread against the replicated vocabulary, and the
 * ids are downloaded in shard order into one host buffer (free with wp_free).
 * devices == NULL: the first n_devices visible GPUs (n_devices <= 0: all of them).
 * A vocabulary that holds whitespace inside a token can match across a cut (the reference's chunking
 * shares the caveat): such inputs are encoded in one piece on the first device of the list. */
int wp_linear_encode_multi(wp_vocab *v, const char *utf8, size_t nbytes, const int *devices,
                           int n_devices, int32_t **ids, size_t *n_ids);

/* A sequence of texts (shards / batches of one corpus) through one handle, pipelined: the upload of text i + 1 and
 * the download of the ids of text i - 1 run on copy streams beside the kernels of text i (second text buffer, id
 * staging buffers), so that host to host costs what the device path costs.  Same ids per text as n_texts calls of
 * wp_linear_encode — the reference's precedent for feeding a corpus piecewise is encodeExternal's batch loop,
 * linear.cpp:355-371.  ids[i] (free each with wp_free; NULL for a text without ids) and n_ids[i] per text. */
int wp_linear_encode_batch(wp_vocab *v, const char *const *texts, const size_t *nbytes, size_t n_texts,
                           int32_t **ids, size_t *n_ids);

/* The same pipeline for a corpus of any length, with constant memory: `next` supplies text i (return 0: no more; the
 * pointer must stay valid until the following call of `next`), `out` receives the ids of text i — in order, one text
 * behind the encodes, valid during the call only (two pinned blocks take turns).  This is the sustained form of the
 * host-to-host metric: uploads, kernels and downloads of neighbouring texts overlap. */
typedef int (*wp_text_source)(void *user, size_t index, const char **utf8, size_t *nbytes);
typedef void (*wp_ids_sink)(void *user, size_t index, const int32_t *ids, size_t n_ids);
int wp_linear_encode_stream(wp_vocab *v, wp_text_source next, wp_ids_sink out, void *user);

/* Sizes the handle's device arenas and host staging for inputs of up to `nbytes`, so that the
 * first encode does not pay for the allocations (about 100 bytes of HBM per input symbol). */
int wp_reserve(wp_vocab *v, size_t nbytes);

/* Memory the library keeps between calls, and how to get it back.  A handle keeps its device arenas (about
 * 100 bytes of HBM per input symbol of its largest encode) until it is destroyed.  A destroyed handle's
 * context (streams, events, code tables) is parked in a process-wide pool of at most 4 for the next handle on
 * the same device — the reference's API is one-shot, linear.cpp:332-335, and sets everything up per call —
 * but only with arenas of up to 256 MB in total: larger ones are released at wp_vocab_destroy.  Returned id
 * blocks are page-locked host memory; wp_free keeps up to 4 of them (6 GB) for reuse.
 * wp_trim releases all of that: the arenas of `v`'s contexts (v may be NULL), the parked contexts' arenas and
 * the pooled id blocks.  (Env WP_NO_CONTEXT_POOL=1 switches the context pool off.) */
int wp_trim(wp_vocab *v);

/* replaces word_piece::linear::encode(text_file, vocab_file)   word_piece.hpp:14, linear.cpp:337-341 */
int wp_linear_encode_file(const char *text_file, const char *vocab_file, int32_t **ids,
                          size_t *n_ids);
/* replaces word_piece::linear::encodeExternal(...)              word_piece.hpp:16-19, linear.cpp:343-374
 * batches of memory_limit/20 bytes extended to the next space; ids appended to
 * out_file as decimal text, each followed by one ' ' (utils.cpp:30-35). */
int wp_linear_encode_external(const char *text_file, const char *vocab_file, const char *out_file,
                              size_t memory_limit);

/* ---- the sibling algorithm: word_piece::fast (src/word_piece.hpp:23-36, src/fast.cpp) ----
 * Per word, longest-match-first against the vocabulary (fast.cpp:19-150) — on the GPU a trie walk per
 * word (csrc/fast.h).  Same ids as the Linear path on vocabularies whose tokens do not span spacing
 * chars (tests/tests.cpp:80-97 asserts linear == fast); an independent on-device cross-check.
 * replaces word_piece::fast::encode(text, vocab)              word_piece.hpp:25, fast.cpp:161-164 */
int wp_fast_encode(wp_vocab *v, const char *utf8, size_t nbytes, int32_t **ids, size_t *n_ids);
/* text and ids in device memory, same buffer contract as wp_linear_encode_device */
int wp_fast_encode_device(wp_vocab *v, const void *d_utf8, size_t nbytes, const int32_t **d_ids,
                          size_t *n_ids);
/* replaces word_piece::fast::encode(text_file, vocab_file)   word_piece.hpp:27, fast.cpp:166-170 */
int wp_fast_encode_file(const char *text_file, const char *vocab_file, int32_t **ids, size_t *n_ids);
/* replaces word_piece::fast::encodeExternal(...)              word_piece.hpp:31-34, fast.cpp:189-220
 * batches of memory_limit/2 bytes extended to the next space; same id text format. */
int wp_fast_encode_external(const char *text_file, const char *vocab_file, const char *out_file,
                            size_t memory_limit);
/* UTF-8 of the stored word of vocab line i (the "##" of continuation tokens stripped, utils.cpp:83-85):
 * what word_piece::fast::decode (fast.cpp:172-187) assembles its strings from.  Returns the length in
 * bytes (-1: no such line) and copies at most `cap` bytes into buf. */
int64_t wp_vocab_token_utf8(const wp_vocab *v, int64_t i, char *buf, size_t cap);

/* ---- options ---- */
#define WP_OPT_FULL_DEPTH 1   /* 1: sort suffixes to full depth (true suffix array).  0 (default):
                                 stop prefix doubling once the sorted depth exceeds the longest
                                 vocab token — token ids are identical for duplicate-free vocabs;
                                 vocabs with duplicate lines force full depth automatically. */
#define WP_OPT_DEVICE 2       /* HIP device ordinal used by this handle (default: current) */
#define WP_OPT_KEEP_DEBUG 3   /* 1: keep SA/rank/LCP/best arrays for wp_linear_debug_fetch */
#define WP_OPT_STAGE_TIMING 4 /* 1: record per-stage device times with HIP events */
#define WP_OPT_LCP_KASAI 5    /* 1: build LCP with the chunked Kasai kernel (linear.cpp:18-70)
                                 instead of deriving it inside the doubling rounds */
/* (option 6, a single-pass form of the group split, was measured slower and removed) */
#define WP_OPT_COVER_ANCHORS 7 /* 1: always derive the walk's start positions from the matches
                                 (default: only when the class rule leaves gaps > 2048 positions
                                 and some spacing char occurs inside a multi-char token, e.g. CJK
                                 text with multi-char CJK tokens; long words of ordinary
                                 vocabularies are walked by pointer doubling instead) */
#define WP_OPT_ARENA_GUARD 8  /* 1: debugging aid — every device arena allocation is followed by a
                                 guard zone that is checked after the encode; a kernel that wrote
                                 outside its buffer makes the call fail with WP_ERR_HIP
                                 (env WP_ARENA_GUARD=1 switches it on for every handle) */
#define WP_OPT_DEVICES 9      /* number of GPUs wp_linear_encode (and with it word_piece::linear::encode)
                                 shards a host buffer over, as wp_linear_encode_multi does:
                                 1 (default) = the handle's device only, -1 = all visible GPUs.
                                 Env WP_DEVICES=<count>|all sets the default for new handles. */
#define WP_OPT_VOCAB_IN_S 10  /* 1: always build S = text . 1 . vocab as linear.cpp:77-101 does.  Default 0:
                                 the vocabulary stays out of the suffix sort (S = text . 1) and enters
                                 through the handle's sorted token list and the tokens' code streams;
                                 the reference's layout is still used for the true suffix array (full
                                 depth, duplicate lines) and when text or tokens hold U+0000 / U+0001.
                                 Same token ids either way. */
#define WP_OPT_SPARSE_EMIT 11 /* 1: the walk writes each id into a per-position array that is compacted
                                 afterwards, always.  Default 0: that path is taken only when several
                                 kernels contribute ids (words longer than a lane walks);
                                 otherwise every workgroup of the walk leaves one compact id list.
                                 Same token ids either way (also env WP_SPARSE_EMIT=1). */
int wp_set_option(wp_vocab *v, int option, int64_t value);

/* ---- statistics of the last encode on this handle (for bench.py / roofline) ---- */
typedef struct {
  int64_t n_bytes, n_text, n_total, alphabet, longest_token, n_ids;
  int32_t symbol_bits, symbols_per_key, rounds, sorted_depth, full_depth;
  int64_t radix_pass_elems;   /* sum over all radix passes of the elements moved      */
  int32_t radix_passes;       /* number of radix scatter launches                      */
  int64_t active_per_round[40];
  double ms_total, ms_decode, ms_sa, ms_lcp, ms_scan, ms_walk; /* WP_OPT_STAGE_TIMING */
  double ms_radix_scatter;    /* device time inside radix scatter kernels (HIP events) */
  int64_t n_anchors;          /* start positions of the parallel walk                  */
  int32_t anchor_mode;        /* 0: class rule, 1: coverage rule (WP_OPT_COVER_ANCHORS),
                                 2: class rule + long words by pointer doubling          */
  double ms_h2d, ms_d2h;      /* wp_linear_encode only: host time of the text upload (with
                                 WP_OPT_STAGE_TIMING) and of the id download                */
  int64_t radix_digit_bytes;  /* digit bytes written next to the records by the radix scatter
                                 launches (1 per element and launch, read back by the next
                                 histogram instead of the 8-byte key)                       */
  double ms_host_total;       /* host entry points: wall time of the whole call (upload, device
                                 path, download)                                            */
  int32_t guard_zones;        /* WP_OPT_ARENA_GUARD: guard zones checked (all intact, or the call
                                 fails)                                                     */
  int32_t n_devices;          /* devices that took part (wp_linear_encode_multi), else 1    */
  int32_t vocab_in_s;         /* 1: S = text . 1 . vocab as in linear.cpp:77-101; 0: S = text . 1 and
                                 the vocab comes in through the per-handle vocab structure  */
  int32_t reserved0;          /* 1: this is the bounds-checking build (libwordpiece_amd_dbg.so)     */
  int64_t needed_after_round0; /* depth-capped mode: suffixes in tied groups that carry the key of an
                                 eligible token longer than the key — the only ones that go on to
                                 round 1 (-1: every tied group does, e.g. full depth)          */
  int32_t key_bits;           /* bits of the codeword stream in a round-0 key; keys of up to 32 bits
                                 are sorted as 8-byte (key, index) records, longer ones as 12-byte */
  int32_t staged_emit;        /* 1: ids left the walk as per-workgroup lists (see WP_OPT_SPARSE_EMIT) */
  int32_t rank_in_pass;       /* 1: the ranks of round 0 were computed inside the first partition pass of the rank
                                 store (one more full-size launch of the radix scatter, not in radix_passes) */
  int32_t trie_refine;        /* 1: the needed groups of round 0 were resolved along the token trie (one walk + one segmented
                                 sort) instead of by prefix-doubling rounds (default in the text-only layout) */
  int64_t arena_bytes;        /* device memory of the handle's two bump arenas after this encode                     */
  int32_t list_retries;       /* 1: the needed list outgrew the room it was given and the encode ran a second time   */
  int32_t hist_in_keys;       /* 1: the key builder took the histogram of the sort's first pass (no digit bytes for it) */
  int64_t radix_pass_bytes;   /* algorithmic bytes of the counted radix scatter launches: record read (without the index
                                 column where the pass makes it up) + record written + digit byte written */
} wp_stats;
int wp_get_stats(const wp_vocab *v, wp_stats *out);

/* ---- debug fetch (WP_OPT_KEEP_DEBUG): copies device intermediates to host ----
 * which: 0 S (dense symbols as int32, n), 1 SA (n), 2 rank (n), 3 lcp (n-1; -1 = "at least
 * sorted_depth"), 4 best_prefix (n), 5 best_suffix (n), 6 code points (n_text) */
int wp_linear_debug_fetch(const wp_vocab *v, int which, int32_t *out, size_t capacity,
                          size_t *n_out);

void wp_free(void *p);
const char *wp_last_error(void);
int wp_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
