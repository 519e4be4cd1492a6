/*
 * wp_oracle.c — CPU restatement of the reference's Linear WordPiece path.
 *
 * TEST INFRASTRUCTURE ONLY (see wp_oracle.h).  Plain C99; every function cites
 * the reference file:line (relative to /root/reference) it restates.  The
 * suffix array is built by a small Larsson–Sadakane-style prefix-doubling
 * sorter of our own (the reference calls libsais, a vendored SA-IS; the suffix
 * array of a string is unique, so any correct sorter yields the same array);
 * wpo_use_libsais() swaps in the reference's libsais compiled from its own
 * sources (oracle/_ref) to cross-check that claim and to time the CPU baseline.
 */
#define _GNU_SOURCE
#include "wp_oracle.h"

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------ */
/* utf8.cpp:10-29 character classes ("C" locale isspace / ispunct)           */
/* ------------------------------------------------------------------------ */
#define SPACE_TOKEN 9601u /* utf8.hpp:14 */

int wpo_is_space(uint32_t c) { /* utf8.cpp:10-12 */
  return (c < 256 && ((c >= 0x09 && c <= 0x0D) || c == 0x20)) || c == SPACE_TOKEN;
}

int wpo_is_punctuation(uint32_t c) { /* utf8.cpp:14-17 */
  if (c < 256) {
    if ((c >= 0x21 && c <= 0x2F) || (c >= 0x3A && c <= 0x40) || (c >= 0x5B && c <= 0x60)
        || (c >= 0x7B && c <= 0x7E)) {
      return 1;
    }
  }
  return c == 183 || c == 171 || c == 187 || c == 8249 || c == 8250 || (8208 <= c && c <= 8248);
}

int wpo_is_chinese(uint32_t c) { /* utf8.cpp:19-27 */
  return (c >= 0x4E00 && c <= 0x9FFF) || (c >= 0x3400 && c <= 0x4DBF)
         || (c >= 0x20000 && c <= 0x2A6DF) || (c >= 0x2A700 && c <= 0x2B73F)
         || (c >= 0x2B740 && c <= 0x2B81F) || (c >= 0x2B820 && c <= 0x2CEAF)
         || (c >= 0xF900 && c <= 0xFAFF) || (c >= 0x2F800 && c <= 0x2FA1F);
}

int wpo_is_spacing_char(uint32_t c) { /* utf8.cpp:29 */
  return wpo_is_space(c) || wpo_is_punctuation(c) || wpo_is_chinese(c);
}

/* ------------------------------------------------------------------------ */
/* utf8.cpp:31-90 UTF-8 decoding                                             */
/* ------------------------------------------------------------------------ */
static int check_byte(uint8_t x) { return (x & 0xc0u) == 0x80u; } /* utf8.cpp:31 */
static int check_codepoint(uint32_t x) {                           /* utf8.cpp:35 */
  return (x < 0xd800) || (0xdfff < x && x < 0x110000);
}
static uint64_t utf_length(uint8_t ch) { /* utf8.cpp:37-52 */
  if ((ch & 0x80u) == 0) return 1;
  if ((ch & 0xe0u) == 0xc0) return 2;
  if ((ch & 0xf0u) == 0xe0) return 3;
  if ((ch & 0xf8u) == 0xf0) return 4;
  return 0;
}

uint32_t wpo_chars_to_utf8(const uint8_t *b, int64_t size, uint64_t *utf8_len) { /* utf8.cpp:54-90 */
  uint64_t length = utf_length(b[0]);
  if (length == 1) {
    *utf8_len = 1;
    return b[0];
  }
  uint32_t cp = 0;
  if (size >= 2 && length == 2 && check_byte(b[1])) {
    cp += (uint32_t)(b[0] & 0x1fu) << 6u;
    cp += (uint32_t)(b[1] & 0x3fu);
    if (cp >= 0x0080 && check_codepoint(cp)) {
      *utf8_len = 2;
      return cp;
    }
  } else if (size >= 3 && length == 3 && check_byte(b[1]) && check_byte(b[2])) {
    cp += (uint32_t)(b[0] & 0x0fu) << 12u;
    cp += (uint32_t)(b[1] & 0x3fu) << 6u;
    cp += (uint32_t)(b[2] & 0x3fu);
    if (cp >= 0x0800 && check_codepoint(cp)) {
      *utf8_len = 3;
      return cp;
    }
  } else if (size >= 4 && length == 4 && check_byte(b[1]) && check_byte(b[2]) && check_byte(b[3])) {
    cp += (uint32_t)(b[0] & 0x07u) << 18u;
    cp += (uint32_t)(b[1] & 0x3fu) << 12u;
    cp += (uint32_t)(b[2] & 0x3fu) << 6u;
    cp += (uint32_t)(b[3] & 0x3fu);
    if (cp >= 0x10000 && check_codepoint(cp)) {
      *utf8_len = 4;
      return cp;
    }
  }
  *utf8_len = 1;
  return WPO_INVALID_UNICODE;
}

size_t wpo_decode_utf8(const uint8_t *s, size_t nbytes, uint32_t *out, int *had_invalid) {
  /* utf8.cpp:130-147: invalid bytes are dropped one at a time */
  size_t n = 0, pos = 0;
  int invalid = 0;
  uint64_t len = 0;
  for (; pos < nbytes; pos += len) {
    uint32_t cp = wpo_chars_to_utf8(s + pos, (int64_t)(nbytes - pos), &len);
    if (cp != WPO_INVALID_UNICODE) {
      out[n++] = cp;
    } else {
      invalid = 1;
    }
  }
  if (had_invalid) *had_invalid = invalid;
  return n;
}

/* ------------------------------------------------------------------------ */
/* utils.cpp:81-146 vocabulary                                               */
/* ------------------------------------------------------------------------ */
typedef struct {
  uint8_t is_prefix, is_special, is_malformed;
  int64_t len;
  uint32_t *word;
} wpo_token;

struct wpo_vocab {
  int64_t V;
  wpo_token *tok;
  int32_t unk_id; /* utils.hpp:30-33: default -1 */
};

void wpo_vocab_destroy(wpo_vocab *v) {
  if (!v) return;
  for (int64_t i = 0; i < v->V; i++) free(v->tok[i].word);
  free(v->tok);
  free(v);
}

int wpo_vocab_create(const uint8_t *buf, const int64_t *off, int64_t V, wpo_vocab **out) {
  wpo_vocab *v = (wpo_vocab *)calloc(1, sizeof(*v));
  if (!v) return WPO_ERR_NOMEM;
  v->V = V;
  v->unk_id = -1;
  v->tok = (wpo_token *)calloc((size_t)(V > 0 ? V : 1), sizeof(wpo_token));
  for (int64_t i = 0; i < V; i++) {
    const uint8_t *line = buf + off[i];
    size_t nb = (size_t)(off[i + 1] - off[i]);
    if (nb == 5 && memcmp(line, "[UNK]", 5) == 0) v->unk_id = (int32_t)i; /* utils.cpp:113-115 */
    wpo_token *t = &v->tok[i];
    /* utils.cpp:81-106 WordPieceToken ctor */
    uint32_t *w = (uint32_t *)malloc(sizeof(uint32_t) * (nb + 1));
    int64_t len = (int64_t)wpo_decode_utf8(line, nb, w, NULL);
    t->is_prefix = 1;
    if (len >= 2 && w[0] == '#' && w[1] == '#') { /* utils.cpp:139-141 */
      t->is_prefix = 0;
      memmove(w, w + 2, sizeof(uint32_t) * (size_t)(len - 2));
      len -= 2;
    } else if (len > 2 && w[0] == '[' && w[len - 1] == ']') { /* utils.cpp:143-146 */
      t->is_special = 1;
    }
    int all_punct = 1;
    for (int64_t k = 0; k < len; k++) {
      if (w[k] == WPO_INVALID_UNICODE) t->is_malformed = 1;
      if (!wpo_is_punctuation(w[k]) && !wpo_is_space(w[k])) all_punct = 0;
    }
    t->word = w;
    t->len = len;
    if (len == 0) { /* utils.cpp:99-101 */
      v->V = i + 1;
      wpo_vocab_destroy(v);
      return WPO_ERR_EMPTY_WORD;
    }
    if (t->is_malformed || (all_punct && len > 1)) t->is_malformed = 1; /* utils.cpp:102-105 */
  }
  *out = v;
  return WPO_OK;
}

int64_t wpo_vocab_size(const wpo_vocab *v) { return v->V; }
int32_t wpo_vocab_unk_id(const wpo_vocab *v) { return v->unk_id; }
int32_t wpo_vocab_token_flags(const wpo_vocab *v, int64_t i) {
  return v->tok[i].is_prefix | (v->tok[i].is_special << 1) | (v->tok[i].is_malformed << 2);
}
int64_t wpo_vocab_token_len(const wpo_vocab *v, int64_t i) { return v->tok[i].len; }
const uint32_t *wpo_vocab_token_word(const wpo_vocab *v, int64_t i) { return v->tok[i].word; }

/* ------------------------------------------------------------------------ */
/* Suffix array (replaces libsais_int, linear.cpp:118-141)                   */
/* ------------------------------------------------------------------------ */
typedef int32_t (*libsais_int_fn)(int32_t *, int32_t *, int32_t, int32_t, int32_t);
typedef int32_t (*libsais_int_omp_fn)(int32_t *, int32_t *, int32_t, int32_t, int32_t, int32_t);
static void *g_libsais_handle = NULL;
static libsais_int_fn g_libsais_int = NULL;
static libsais_int_omp_fn g_libsais_int_omp = NULL;

int wpo_use_libsais(const char *so_path) {
  void *h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return -1;
  libsais_int_fn f = (libsais_int_fn)dlsym(h, "libsais_int");
  if (!f) {
    dlclose(h);
    return -2;
  }
  g_libsais_handle = h;
  g_libsais_int = f;
  g_libsais_int_omp = (libsais_int_omp_fn)dlsym(h, "libsais_int_omp");
  return 0;
}
void wpo_use_builtin_sa(void) {
  g_libsais_int = NULL;
  g_libsais_int_omp = NULL;
}

static const int32_t *g_key; /* sort key table for the comparators below */
static int64_t g_key_n, g_key_h;
static int cmp_first_symbol(const void *a, const void *b) {
  int32_t x = g_key[*(const int32_t *)a], y = g_key[*(const int32_t *)b];
  return (x > y) - (x < y);
}
static int cmp_second_rank(const void *a, const void *b) {
  int64_t i = (int64_t) * (const int32_t *)a + g_key_h, j = (int64_t) * (const int32_t *)b + g_key_h;
  int32_t x = i < g_key_n ? g_key[i] : -1, y = j < g_key_n ? g_key[j] : -1;
  return (x > y) - (x < y);
}

/* Plain lexicographic suffix order, a proper prefix sorts before its extension
 * (what libsais_int returns).  rank[i] = first SA slot of i's group. */
static int builtin_suffix_array(const int32_t *S, int64_t n, int32_t *SA) {
  int32_t *rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  int32_t *nrank = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  if (!rank || !nrank) return -2;
  for (int64_t i = 0; i < n; i++) SA[i] = (int32_t)i;
  g_key = S;
  qsort(SA, (size_t)n, sizeof(int32_t), cmp_first_symbol);
  int64_t unsorted = 0;
  for (int64_t x = 0, head = 0; x < n; x++) {
    if (x > 0 && S[SA[x]] != S[SA[x - 1]]) head = x;
    rank[SA[x]] = (int32_t)head;
  }
  for (int64_t x = 0; x < n;) {
    int64_t e = x + 1;
    while (e < n && rank[SA[e]] == rank[SA[x]]) e++;
    if (e - x > 1) unsorted += e - x;
    x = e;
  }
  for (int64_t h = 1; unsorted > 0; h *= 2) {
    g_key = rank;
    g_key_n = n;
    g_key_h = h;
    unsorted = 0;
    memcpy(nrank, rank, sizeof(int32_t) * (size_t)n);
    for (int64_t x = 0; x < n;) {
      int64_t e = x + 1;
      while (e < n && rank[SA[e]] == rank[SA[x]]) e++;
      if (e - x > 1) {
        qsort(SA + x, (size_t)(e - x), sizeof(int32_t), cmp_second_rank);
        int64_t head = x, run = 1;
        for (int64_t y = x; y < e; y++) {
          if (y > x && cmp_second_rank(&SA[y], &SA[y - 1]) != 0) {
            if (run > 1) unsorted += run;
            head = y;
            run = 0;
          }
          if (y > x) run++;
          nrank[SA[y]] = (int32_t)head;
        }
        if (run > 1) unsorted += run;
      }
      x = e;
    }
    int32_t *t = rank;
    rank = nrank;
    nrank = t;
  }
  free(rank);
  free(nrank);
  return 0;
}

int wpo_suffix_array(const int32_t *S, int64_t n, int32_t alphabet_size, int32_t *SA) {
  if (n <= 0) return 0;
  if (g_libsais_int) {
    /* linear.cpp:108-137: fs heuristic and the OpenMP switch at n > 1e7 */
    int64_t fs = 0, k = alphabet_size;
    if (n > 1000000 && n > k && k < 100000000) {
      fs = 6 * k;
      if (fs > n) fs = 4 * k;
      if (fs > n) fs = k;
    }
    int32_t *T = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n + fs));
    if (!T || !buf) return -2;
    memcpy(T, S, sizeof(int32_t) * (size_t)n);
    int32_t rc;
    if (g_libsais_int_omp) {
      rc = g_libsais_int_omp(T, buf, (int32_t)n, (int32_t)k, (int32_t)fs, n > 10000000 ? 0 : 1);
    } else {
      rc = g_libsais_int(T, buf, (int32_t)n, (int32_t)k, (int32_t)fs);
    }
    memcpy(SA, buf, sizeof(int32_t) * (size_t)n);
    free(T);
    free(buf);
    return rc;
  }
  return builtin_suffix_array(S, n, SA);
}

/* ------------------------------------------------------------------------ */
/* linear.cpp:18-41 calcLcpImpl (Kasai over text positions [begin,end))      */
/* ------------------------------------------------------------------------ */
static void kasai_range(const int32_t *S, const int32_t *SA, const int32_t *rank, int64_t n,
                        int32_t *lcp, int64_t begin, int64_t end) {
  int64_t prefix_len = 0;
  for (int64_t i = begin; i < end; i++) {
    int64_t sa_index = rank[i];
    if (sa_index + 1 != n) {
      int64_t suf_index = SA[sa_index + 1];
      int64_t m = i > suf_index ? i : suf_index;
      while (m + prefix_len < n && S[i + prefix_len] == S[suf_index + prefix_len]) prefix_len++;
      lcp[sa_index] = (int32_t)prefix_len;
      if (prefix_len > 0) prefix_len--;
    }
  }
}

void wpo_kasai(const int32_t *S, const int32_t *SA, const int32_t *rank, int64_t n, int32_t *lcp) {
  kasai_range(S, SA, rank, n, lcp, 0, n);
}

/* ------------------------------------------------------------------------ */
/* linear.cpp:161-189 get_closest                                            */
/* ------------------------------------------------------------------------ */
static void get_closest(const wpo_vocab *v, int64_t n, int64_t longest, const int32_t *lcp,
                        const int32_t *who, int right_side, int is_prefix_predicate,
                        int32_t *result) {
  int32_t *st_id = (int32_t *)malloc(sizeof(int32_t) * (size_t)(v->V + longest + 2));
  int32_t *st_len = (int32_t *)malloc(sizeof(int32_t) * (size_t)(v->V + longest + 2));
  int64_t sp = 0;
  for (int64_t i = 0; i < n; i++) {
    result[i] = -1;
    if (i > 0) {
      int64_t index = right_side ? n - i - 1 : i - 1;
      while (sp > 0 && st_len[sp - 1] > lcp[index]) sp--;
    }
    int64_t index = right_side ? n - 1 - i : i;
    if (who[index] != -1) {
      const wpo_token *t = &v->tok[who[index]];
      if (t->is_prefix == is_prefix_predicate && !t->is_malformed && !t->is_special) {
        st_id[sp] = who[index];
        st_len[sp] = (int32_t)t->len;
        sp++;
      }
    }
    if (sp > 0) result[i] = st_id[sp - 1];
  }
  free(st_id);
  free(st_len);
}

/* ------------------------------------------------------------------------ */
/* linear.cpp:215-274 is_word_prefix + match_word_piece                      */
/* ------------------------------------------------------------------------ */
typedef struct {
  const wpo_vocab *v;
  const uint32_t *text;
  int64_t n_text, n;
  const int32_t *rank, *blp, *brp, *bls, *brs;
} walk_ctx;

static int is_word_prefix(const walk_ctx *c, int64_t index) { /* linear.cpp:215-219 */
  return index == 0 || wpo_is_spacing_char(c->text[index]) || wpo_is_spacing_char(c->text[index - 1]);
}

typedef struct {
  int32_t *d;
  size_t n, cap;
} ivec;
static void ivec_push(ivec *a, int32_t x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 64;
    a->d = (int32_t *)realloc(a->d, a->cap * sizeof(int32_t));
  }
  a->d[a->n++] = x;
}

static void match_word_piece(const walk_ctx *c, int64_t match_index, int64_t end, ivec *out) {
  const wpo_vocab *v = c->v;
  while (match_index != end && wpo_is_space(c->text[match_index])) ++match_index;
  size_t tokens_since_prefix = 0;
  while (match_index < end) {
    int64_t left_sa_id = c->rank[match_index];
    int64_t right_sa_id = c->n - 1 - left_sa_id;
    int prefix = is_word_prefix(c, match_index);
    int32_t x = prefix ? c->blp[left_sa_id] : c->bls[left_sa_id];
    int32_t y = prefix ? c->brp[right_sa_id] : c->brs[right_sa_id];
    if (x != -1 || y != -1) {
      int32_t token_id;
      if (x != -1 && y != -1) {
        token_id = v->tok[x].len > v->tok[y].len ? x : y;
      } else {
        token_id = x > y ? x : y;
      }
      ++tokens_since_prefix;
      ivec_push(out, token_id);
      match_index += v->tok[token_id].len;
      if (match_index != end && is_word_prefix(c, match_index)) tokens_since_prefix = 0;
    } else {
      while (tokens_since_prefix > 0) {
        out->n--;
        --tokens_since_prefix;
      }
      ivec_push(out, v->unk_id);
      ++match_index;
      while (match_index != end && !is_word_prefix(c, match_index)) ++match_index;
    }
    while (match_index != end && wpo_is_space(c->text[match_index])) ++match_index;
  }
}

/* ------------------------------------------------------------------------ */
/* linear.cpp:72-328 encodeLinearWordPiece(Impl)                             */
/* ------------------------------------------------------------------------ */
static int nthreads_eff(int threads) {
#ifdef _OPENMP
  return threads > 0 ? threads : omp_get_max_threads();
#else
  (void)threads;
  return 1;
#endif
}

static size_t decode_text(const uint8_t *text, size_t nbytes, int threads, uint32_t *out) {
  /* utils.cpp:37-79 parseText: chunks cut at symbol starts, then concatenated */
  const size_t kWorkBatch = 5000000;
  if (threads <= 1 || nbytes < 2 * kWorkBatch) return wpo_decode_utf8(text, nbytes, out, NULL);
  size_t tc = nbytes / kWorkBatch;
  if ((size_t)threads < tc) tc = (size_t)threads;
  size_t work_batch = nbytes / tc + 1;
  size_t *b = (size_t *)calloc(tc + 1, sizeof(size_t)), *cnt = (size_t *)calloc(tc + 1, sizeof(size_t));
  size_t ws = 0;
  for (size_t t = 0; t < tc; t++) {
    b[t] = ws;
    size_t we = ws + work_batch < nbytes ? ws + work_batch : nbytes;
    while (we < nbytes && check_byte(text[we])) ++we;
    ws = we;
  }
  b[tc] = nbytes;
  uint32_t **tmp = (uint32_t **)calloc(tc, sizeof(uint32_t *));
#pragma omp parallel for num_threads(threads) schedule(static, 1)
  for (long t = 0; t < (long)tc; t++) {
    size_t len = b[t + 1] - b[t];
    tmp[t] = (uint32_t *)malloc(sizeof(uint32_t) * (len + 1));
    cnt[t] = wpo_decode_utf8(text + b[t], len, tmp[t], NULL);
  }
  size_t n = 0;
  for (size_t t = 0; t < tc; t++) {
    memcpy(out + n, tmp[t], cnt[t] * sizeof(uint32_t));
    n += cnt[t];
    free(tmp[t]);
  }
  free(tmp);
  free(b);
  free(cnt);
  return n;
}

static int encode_impl(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int threads,
                       wpo_debug *dbg) {
  memset(dbg, 0, sizeof(*dbg));
  if (nbytes == 0) return WPO_OK; /* linear.cpp:323-325 */
  threads = threads == 1 ? 1 : nthreads_eff(threads);

  uint32_t *tx = (uint32_t *)malloc(sizeof(uint32_t) * (nbytes + 1));
  if (!tx) return WPO_ERR_NOMEM;
  int64_t n_text = (int64_t)decode_text(text, nbytes, threads, tx);

  /* linear.cpp:77-103 */
  int64_t n = n_text + 1, longest = 1;
  for (int64_t i = 0; i < v->V; i++) {
    n += v->tok[i].len + 1;
    if (v->tok[i].len > longest) longest = v->tok[i].len;
  }
  int32_t *S = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  uint32_t alphabet = 1;
  int64_t pos = 0;
  for (int64_t i = 0; i < n_text; i++) {
    S[pos++] = (int32_t)tx[i];
    if (tx[i] > alphabet) alphabet = tx[i];
  }
  S[pos++] = 1;
  for (int64_t i = 0; i < v->V; i++) {
    for (int64_t k = 0; k < v->tok[i].len; k++) {
      uint32_t c = v->tok[i].word[k];
      S[pos++] = (int32_t)c;
      if (c > alphabet) alphabet = c;
    }
    S[pos++] = 1;
  }
  ++alphabet;
  if ((uint64_t)n > 2000000000ull || alphabet > 2000000000u) { /* linear.cpp:104-106 */
    free(tx);
    free(S);
    return WPO_ERR_TOO_LARGE;
  }

  int32_t *SA = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  int rc = wpo_suffix_array(S, n, (int32_t)alphabet, SA);
  if (rc != 0) { /* linear.cpp:139-141 */
    free(tx);
    free(S);
    free(SA);
    return WPO_ERR_SACA;
  }

  int32_t *rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)n); /* linear.cpp:144-147 */
  for (int64_t i = 0; i < n; i++) rank[SA[i]] = (int32_t)i;

  /* linear.cpp:43-70 calcLcp: chunks restart Kasai with prefix_len = 0 */
  int32_t *lcp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 1 ? n - 1 : 1));
  {
    const int64_t kWorkBatch = 1000000;
    if (threads <= 1 || n < 2 * kWorkBatch) {
      kasai_range(S, SA, rank, n, lcp, 0, n);
    } else {
      int64_t tc = n / kWorkBatch < threads ? n / kWorkBatch : threads;
      int64_t wb = n / tc + 1;
#pragma omp parallel for num_threads(threads) schedule(static, 1)
      for (long t = 0; t < (long)tc; t++) {
        int64_t b = t * wb, e = b + wb < n ? b + wb : n;
        if (b < e) kasai_range(S, SA, rank, n, lcp, b, e);
      }
    }
  }

  /* linear.cpp:153-160 */
  int32_t *who = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  for (int64_t i = 0; i < n; i++) who[i] = -1;
  {
    int64_t start = n_text + 1;
    for (int64_t i = 0; i < v->V; i++) {
      who[rank[start]] = (int32_t)i;
      start += v->tok[i].len + 1;
    }
  }

  /* linear.cpp:190-213 */
  int32_t *best[4];
  for (int k = 0; k < 4; k++) best[k] = (int32_t *)malloc(sizeof(int32_t) * (size_t)n);
  if (threads <= 1 || n < 1000000) {
    get_closest(v, n, longest, lcp, who, 0, 1, best[0]);
    get_closest(v, n, longest, lcp, who, 1, 1, best[1]);
    get_closest(v, n, longest, lcp, who, 0, 0, best[2]);
    get_closest(v, n, longest, lcp, who, 1, 0, best[3]);
  } else {
#pragma omp parallel for num_threads(threads < 4 ? threads : 4) schedule(static, 1)
    for (int k = 0; k < 4; k++) get_closest(v, n, longest, lcp, who, k & 1, k < 2, best[k]);
  }

  walk_ctx c = {v, tx, n_text, n, rank, best[0], best[1], best[2], best[3]};
  ivec out = {NULL, 0, 0};
  {
    /* linear.cpp:276-316: chunk ends advance to the next is_space */
    const int64_t kWorkBatch = 1000000;
    if (threads <= 1 || n_text < 2 * kWorkBatch) {
      match_word_piece(&c, 0, n_text, &out);
    } else {
      int64_t tc = n_text / kWorkBatch < threads ? n_text / kWorkBatch : threads;
      int64_t wb = n_text / tc + 1;
      int64_t *bnd = (int64_t *)calloc((size_t)tc + 1, sizeof(int64_t));
      int64_t ws = 0, used = 0;
      for (int64_t t = 0; t < tc && ws < n_text; t++) {
        int64_t we = ws + wb < n_text ? ws + wb : n_text;
        while (we < n_text && !wpo_is_space(tx[we])) ++we;
        bnd[t] = ws;
        bnd[t + 1] = we;
        ws = we;
        used = t + 1;
      }
      ivec *parts = (ivec *)calloc((size_t)tc, sizeof(ivec));
#pragma omp parallel for num_threads(threads) schedule(static, 1)
      for (long t = 0; t < (long)used; t++) match_word_piece(&c, bnd[t], bnd[t + 1], &parts[t]);
      for (int64_t t = 0; t < used; t++) {
        for (size_t k = 0; k < parts[t].n; k++) ivec_push(&out, parts[t].d[k]);
        free(parts[t].d);
      }
      free(parts);
      free(bnd);
    }
  }

  free(tx);
  dbg->n_text = n_text;
  dbg->n = n;
  dbg->longest = longest;
  dbg->alphabet_size = alphabet;
  dbg->S = S;
  dbg->SA = SA;
  dbg->rank = rank;
  dbg->lcp = lcp;
  dbg->who = who;
  dbg->best_left_prefix = best[0];
  dbg->best_right_prefix = best[1];
  dbg->best_left_suffix = best[2];
  dbg->best_right_suffix = best[3];
  dbg->ids = out.d;
  dbg->n_ids = out.n;
  return WPO_OK;
}

void wpo_debug_free(wpo_debug *d) {
  free(d->S);
  free(d->SA);
  free(d->rank);
  free(d->lcp);
  free(d->who);
  free(d->best_left_prefix);
  free(d->best_right_prefix);
  free(d->best_left_suffix);
  free(d->best_right_suffix);
  free(d->ids);
  memset(d, 0, sizeof(*d));
}

int wpo_encode_debug(const wpo_vocab *v, const uint8_t *text, size_t nbytes, wpo_debug *dbg) {
  return encode_impl(v, text, nbytes, 1, dbg);
}

static int encode_common(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int threads,
                         int32_t **ids, size_t *n_ids) {
  wpo_debug d;
  int rc = encode_impl(v, text, nbytes, threads, &d);
  if (rc != WPO_OK) return rc;
  *ids = d.ids;
  *n_ids = d.n_ids;
  d.ids = NULL;
  wpo_debug_free(&d);
  return WPO_OK;
}

int wpo_encode(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int32_t **ids, size_t *n_ids) {
  return encode_common(v, text, nbytes, 1, ids, n_ids);
}

int wpo_encode_mt(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int threads,
                  int32_t **ids, size_t *n_ids) {
  return encode_common(v, text, nbytes, threads <= 0 ? 0 : threads, ids, n_ids);
}

/* ------------------------------------------------------------------------ */
/* fast.cpp:19-150 encodeFastWordPieceImpl — the sibling algorithm            */
/* (word_piece::fast, src/word_piece.hpp:23-36): per word, longest-match-first */
/* lookups of text segments in two maps (prefix-class / ##-class tokens).      */
/* The reference's std::unordered_map<VectorSegment,int> is restated as an     */
/* open-addressing table over (length, polynomial hash), verified by a symbol  */
/* compare; operator[] assignment = the last of equal words wins (fast.cpp:34).*/
/* ------------------------------------------------------------------------ */
typedef struct {
  int32_t *slot; /* token index or -1 */
  size_t mask;
} fmap;

static uint64_t fhash_step(uint64_t h, uint32_t c) { return (h * 1099511628211ull) ^ (uint64_t)(c + 0x9e3779b9u); }

static void fmap_init(fmap *m, size_t items) {
  size_t cap = 16;
  while (cap < 2 * items + 2) cap *= 2;
  m->slot = (int32_t *)malloc(cap * sizeof(int32_t));
  for (size_t i = 0; i < cap; i++) m->slot[i] = -1;
  m->mask = cap - 1;
}

static void fmap_put(fmap *m, const wpo_vocab *v, int32_t id) { /* fast.cpp:34: (*word_to_id)[segment] = i */
  const wpo_token *t = &v->tok[id];
  uint64_t h = 1469598103934665603ull;
  for (int64_t k = 0; k < t->len; k++) h = fhash_step(h, t->word[k]);
  size_t p = (size_t)h & m->mask;
  for (;;) {
    int32_t cur = m->slot[p];
    if (cur < 0) {
      m->slot[p] = id;
      return;
    }
    const wpo_token *o = &v->tok[cur];
    if (o->len == t->len && memcmp(o->word, t->word, sizeof(uint32_t) * (size_t)t->len) == 0) {
      m->slot[p] = id; /* same word: the later line replaces the earlier one */
      return;
    }
    p = (p + 1) & m->mask;
  }
}

static int32_t fmap_find(const fmap *m, const wpo_vocab *v, const uint32_t *seg, int64_t len, uint64_t h) {
  size_t p = (size_t)h & m->mask;
  for (;;) {
    int32_t cur = m->slot[p];
    if (cur < 0) return -1;
    const wpo_token *o = &v->tok[cur];
    if (o->len == len && memcmp(o->word, seg, sizeof(uint32_t) * (size_t)len) == 0) return cur;
    p = (p + 1) & m->mask;
  }
}

typedef struct {
  const wpo_vocab *v;
  const uint32_t *text;
  int64_t n_text;
  fmap prefix_to_id, suffix_to_id;
  int64_t max_len;
  uint64_t *hbuf; /* per worker: prefix hashes of the current segment */
} fast_ctx;

static int fast_is_word_prefix(const fast_ctx *c, int64_t index) { /* fast.cpp:39-42 */
  return index == 0 || wpo_is_spacing_char(c->text[index]) || wpo_is_spacing_char(c->text[index - 1]);
}

/* fast.cpp:44-108 worker(begin, end) */
static void fast_worker(const fast_ctx *c, int64_t begin, int64_t end, uint64_t *hbuf, ivec *out) {
  const uint32_t *text = c->text;
  const int64_t max_len = c->max_len;
  while (begin != end && wpo_is_space(text[begin])) ++begin; /* fast.cpp:48-50 */
  size_t tokens_since_prefix = 0;
  while (begin != end) {
    int64_t word_len = 1; /* fast.cpp:55-61 */
    if (!wpo_is_punctuation(text[begin])) {
      int64_t lim = max_len < end - begin ? max_len : end - begin;
      while (word_len < lim && !wpo_is_spacing_char(text[begin + word_len])) ++word_len;
    }
    const uint32_t *seg = text + begin;
    const fmap *map = fast_is_word_prefix(c, begin) ? &c->prefix_to_id : &c->suffix_to_id; /* fast.cpp:65 */
    uint64_t h = 1469598103934665603ull;
    for (int64_t k = 0; k < word_len; k++) {
      h = fhash_step(h, seg[k]);
      hbuf[k] = h;
    }
    int64_t len = word_len; /* fast.cpp:67-78: find, else pop_back */
    int32_t id = -1;
    while (len > 0) {
      id = fmap_find(map, c->v, seg, len, hbuf[len - 1]);
      if (id >= 0) break;
      --len;
    }
    if (len > 0) { /* found */
      ++tokens_since_prefix;
      ivec_push(out, id);
      begin += len;
      if (begin != end && fast_is_word_prefix(c, begin)) tokens_since_prefix = 0; /* fast.cpp:90-92 */
    } else { /* fast.cpp:80-89 */
      while (tokens_since_prefix > 0) {
        out->n--;
        --tokens_since_prefix;
      }
      ivec_push(out, c->v->unk_id);
      begin += word_len;
      while (begin != end && !fast_is_word_prefix(c, begin)) ++begin;
    }
    while (begin != end && wpo_is_space(text[begin])) ++begin; /* fast.cpp:94-96 */
  }
}

/* fast.cpp:152-158 encodeFastWordPiece + fast.cpp:19-150.  threads > 1: the reference's own chunking
 * (fast.cpp:113-146: chunk ends advanced to the next is_space; results concatenated in order). */
int wpo_fast_encode_mt(const wpo_vocab *v, const uint8_t *text8, size_t nbytes, int threads, int32_t **ids,
                       size_t *n_ids) {
  *ids = NULL;
  *n_ids = 0;
  if (nbytes == 0) return WPO_OK; /* fast.cpp:154-156 */
  uint32_t *text = (uint32_t *)malloc(sizeof(uint32_t) * (nbytes + 1));
  if (!text) return WPO_ERR_NOMEM;
  const int T = nthreads_eff(threads);
  const int64_t n_text = (int64_t)decode_text(text8, nbytes, T, text);
  fast_ctx c;
  memset(&c, 0, sizeof(c));
  c.v = v;
  c.text = text;
  c.n_text = n_text;
  size_t np = 0, ns = 0;
  int64_t max_len = 0;
  for (int64_t i = 0; i < v->V; i++) { /* fast.cpp:25-35 */
    const wpo_token *t = &v->tok[i];
    if (t->is_special || t->is_malformed) continue;
    if (t->len > max_len) max_len = t->len;
    if (t->is_prefix) np++; else ns++;
  }
  fmap_init(&c.prefix_to_id, np);
  fmap_init(&c.suffix_to_id, ns);
  for (int64_t i = 0; i < v->V; i++) {
    const wpo_token *t = &v->tok[i];
    if (t->is_special || t->is_malformed) continue;
    fmap_put(t->is_prefix ? &c.prefix_to_id : &c.suffix_to_id, v, (int32_t)i);
  }
  c.max_len = max_len < n_text ? max_len : n_text; /* fast.cpp:37 */
  int64_t nchunks = 1;
  if (T > 1 && n_text >= 2000000) nchunks = n_text / 1000000 < T ? n_text / 1000000 : T; /* fast.cpp:111-116 */
  int64_t *bounds = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nchunks + 1));
  ivec *parts = (ivec *)calloc((size_t)nchunks, sizeof(ivec));
  const int64_t batch = n_text / nchunks + 1;
  bounds[0] = 0;
  for (int64_t k = 0; k < nchunks; k++) { /* fast.cpp:119-130 */
    int64_t e = bounds[k] + batch < n_text ? bounds[k] + batch : n_text;
    if (e < bounds[k]) e = bounds[k];
    while (e < n_text && !wpo_is_space(text[e])) ++e;
    if (k + 1 == nchunks) e = n_text;
    bounds[k + 1] = e;
  }
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
  for (int64_t k = 0; k < nchunks; k++) {
    uint64_t *hbuf = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)(c.max_len + 1));
    fast_worker(&c, bounds[k], bounds[k + 1], hbuf, &parts[k]);
    free(hbuf);
  }
  size_t total = 0;
  for (int64_t k = 0; k < nchunks; k++) total += parts[k].n;
  int32_t *outv = (int32_t *)malloc(sizeof(int32_t) * (total ? total : 1));
  size_t pos = 0;
  for (int64_t k = 0; k < nchunks; k++) {
    if (parts[k].n) memcpy(outv + pos, parts[k].d, sizeof(int32_t) * parts[k].n);
    pos += parts[k].n;
    free(parts[k].d);
  }
  free(parts);
  free(bounds);
  free(c.prefix_to_id.slot);
  free(c.suffix_to_id.slot);
  free(text);
  *ids = outv;
  *n_ids = total;
  return WPO_OK;
}

int wpo_fast_encode(const wpo_vocab *v, const uint8_t *text, size_t nbytes, int32_t **ids, size_t *n_ids) {
  return wpo_fast_encode_mt(v, text, nbytes, 1, ids, n_ids);
}

void wpo_free(void *p) { free(p); }

const char *wpo_strerror(int rc) {
  switch (rc) {
    case WPO_OK: return "ok";
    case WPO_ERR_EMPTY_WORD: return "Vocab word is empty";
    case WPO_ERR_TOO_LARGE: return "64bit not implemented";
    case WPO_ERR_SACA: return "SACA return code";
    case WPO_ERR_NOMEM: return "out of memory";
  }
  return "unknown";
}
