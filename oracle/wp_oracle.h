/*
 * wp_oracle.h — CPU restatement of gleb-kov/wordpiece's Linear WordPiece path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under wordpiece_amd/ (the product) may
 * include, link or call this.  Allowed users: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg — as the checker / reported baseline only.
 *
 * Parity pinning: the 28 known-answer vectors of the reference's own
 * tests/tests.cpp:137-217 (tests/golden/reference_tests_cpp.json), the edge
 * cases recorded from the reference in SURVEY.md §0.2, and component-level
 * cross-checks against reference sources compiled as-is into oracle/_ref
 * (libsais.c, utf8.cpp, utils.cpp).  src/linear.cpp itself needs a Boost header
 * the image lacks and is therefore NOT built.
 */
#ifndef WP_ORACLE_H
#define WP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WPO_OK 0
#define WPO_ERR_EMPTY_WORD 1   /* "Vocab word is empty"      utils.cpp:99-101 */
#define WPO_ERR_TOO_LARGE 2    /* "64bit not implemented"    linear.cpp:104-106 */
#define WPO_ERR_SACA 3         /* "SACA return code: N"      linear.cpp:139-141 */
#define WPO_ERR_NOMEM 4

#define WPO_INVALID_UNICODE 0x110000u /* utf8.hpp:12 */

typedef struct wpo_vocab wpo_vocab;

/* --- utf8.cpp:10-29 --------------------------------------------------- */
int wpo_is_space(uint32_t c);
int wpo_is_punctuation(uint32_t c);
int wpo_is_chinese(uint32_t c);
int wpo_is_spacing_char(uint32_t c);

/* --- utf8.cpp:54-90 (chars_to_utf8) ---------------------------------- */
uint32_t wpo_chars_to_utf8(const uint8_t *begin, int64_t size, uint64_t *utf8_len);

/* --- utf8.cpp:130-147 (decode_utf8): out must hold nbytes entries ---- */
size_t wpo_decode_utf8(const uint8_t *s, size_t nbytes, uint32_t *out, int *had_invalid);

/* --- utils.cpp:81-146: vocab lines given as one buffer + V+1 offsets -- */
int wpo_vocab_create(const uint8_t *buf, const int64_t *offsets, int64_t V, wpo_vocab **out);
void wpo_vocab_destroy(wpo_vocab *v);
int64_t wpo_vocab_size(const wpo_vocab *v);
int32_t wpo_vocab_unk_id(const wpo_vocab *v);
/* flags: bit0 is_prefix, bit1 is_special, bit2 is_malformed */
int32_t wpo_vocab_token_flags(const wpo_vocab *v, int64_t i);
int64_t wpo_vocab_token_len(const wpo_vocab *v, int64_t i);
const uint32_t *wpo_vocab_token_word(const wpo_vocab *v, int64_t i);

/* Optional: use the reference's own libsais (compiled from its sources into
 * oracle/_ref/libsais_ref.so) for the SA stage instead of the built-in
 * prefix-doubling SA.  Returns 0 on success. */
int wpo_use_libsais(const char *so_path);
void wpo_use_builtin_sa(void);

/* --- linear.cpp:72-328: the whole path; ids malloc'd, free with wpo_free */
int wpo_encode(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int32_t **ids,
               size_t *n_ids);
/* Same result for vocabularies without whitespace inside tokens; stages are
 * chunk-parallel as in the reference (linear.cpp:43-70,190-213,276-316). */
int wpo_encode_mt(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int threads,
                  int32_t **ids, size_t *n_ids);

/* --- fast.cpp:19-158: the sibling algorithm word_piece::fast::encode (hash-map longest match per
 * word).  Same ids as Linear on the reference's own test vocabularies (tests.cpp:80-97 asserts both);
 * it differs where a token spans a spacing char or on duplicate lines (Q9). */
int wpo_fast_encode(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int32_t **ids, size_t *n_ids);
int wpo_fast_encode_mt(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int threads, int32_t **ids,
                       size_t *n_ids);

/* Intermediates for kernel-level parity tests.  All arrays malloc'd. */
typedef struct {
  int64_t n_text;          /* code points after decode            */
  int64_t n;               /* total_length      linear.cpp:77-82  */
  int64_t longest;         /* longest_word_vocab                  */
  uint32_t alphabet_size;  /* after ++          linear.cpp:103    */
  int32_t *S;              /* [n]               linear.cpp:84-101 */
  int32_t *SA;             /* [n]               linear.cpp:118-137*/
  int32_t *rank;           /* [n] suf_array_index linear.cpp:144-147 */
  int32_t *lcp;            /* [n-1]             linear.cpp:18-41  */
  int32_t *who;            /* [n]               linear.cpp:153-160*/
  int32_t *best_left_prefix, *best_right_prefix;  /* [n] linear.cpp:161-213 */
  int32_t *best_left_suffix, *best_right_suffix;  /* right arrays in scan order */
  int32_t *ids;            /* [n_ids]           linear.cpp:221-274*/
  size_t n_ids;
} wpo_debug;
int wpo_encode_debug(const wpo_vocab *v, const uint8_t *text, size_t nbytes, wpo_debug *dbg);
void wpo_debug_free(wpo_debug *dbg);

/* stand-alone stage entry points */
int wpo_suffix_array(const int32_t *S, int64_t n, int32_t alphabet_size, int32_t *SA);
void wpo_kasai(const int32_t *S, const int32_t *SA, const int32_t *rank, int64_t n, int32_t *lcp);

void wpo_free(void *p);
const char *wpo_strerror(int rc);

#ifdef __cplusplus
}
#endif
#endif
