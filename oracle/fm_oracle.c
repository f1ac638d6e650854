/*
 * fm_oracle.c — TEST INFRASTRUCTURE ONLY (see fm_oracle.h).
 *
 * Plain-C restatement of the reference's match() hot path.  Faithful on purpose: same
 * narrowing binary search, same registration walk, same candidate order, same float
 * expression order, same heap discipline — including the parts that look odd.
 * Build with -ffp-contract=off and without -ffast-math / -march=native so that every
 * float operation rounds exactly like the reference's baseline x86-64 build.
 *
 * Parity pin: reference unbuildable here (Boost / OpenNMT Tokenizer absent); pinned by the
 * reference's own known answers and fixtures (tests/test_oracle_golden.py).
 */
#define _GNU_SOURCE
#include "fm_oracle.h"

#include <float.h>
#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* data                                                                                        */
/* ------------------------------------------------------------------------------------------ */

/* fuzzy::Sentence (include/fuzzy/sentence.hh:24-48): tab-joined real tokens + sparse itoks */
typedef struct fmo_real {
  char*     tokstring;
  uint32_t  n_itoks;
  uint64_t* itok_idx;
  char**    itok_str;
} fmo_real;

/* SuffixArray members (include/fuzzy/suffix_array.hh:50-61) + the SuffixArrayIndex/VocabIndexer
 * read side that match() touches (suffix_array_index.hh:50-54, vocab_indexer.hh:38-41) */
struct fmo_index {
  uint64_t  max_tokens;
  uint32_t* suf_sid;
  uint16_t* suf_pos;
  uint64_t  n_suf, cap_suf;
  uint32_t* buf;
  uint64_t  n_buf, cap_buf;
  uint32_t* sent_pos;
  uint64_t  n_sent, cap_sent;
  uint32_t* qva;
  uint64_t  n_qva;
  uint16_t* sentence_length; /* per suffix; compute_sentence_length, suffix_array.cc:253-261 */
  int       sorted;
  uint32_t  vocab_size;
  uint32_t* sfreq;
  /* the oracle's OWN count of VocabIndexer::sfreq (src/vocab_indexer.cc:73-90, addWords: once per distinct word of a sentence),
   * kept as sentences are added; fmo_index_set_vocab(.., NULL) adopts it, so that a parity test need not hand over the
   * product's counts */
  uint32_t* own_sfreq;
  uint32_t* own_stamp;  /* sentence (index + 1) that last bumped the word */
  uint32_t  own_cap;
  fmo_real* real;
  uint64_t  cap_real;
  int       has_real;
};

static void* xrealloc(void* p, size_t n) {
  void* q = realloc(p, n ? n : 1);
  if (!q) abort();
  return q;
}

fmo_index* fmo_index_create(uint64_t max_tokens_in_pattern) {
  fmo_index* ix = (fmo_index*)calloc(1, sizeof(*ix));
  if (!ix) return NULL;
  ix->max_tokens = max_tokens_in_pattern;
  return ix;
}

static void real_clear(fmo_real* r) {
  free(r->tokstring);
  for (uint32_t i = 0; i < r->n_itoks; i++) free(r->itok_str[i]);
  free(r->itok_idx);
  free(r->itok_str);
  memset(r, 0, sizeof(*r));
}

void fmo_index_free(fmo_index* ix) {
  if (!ix) return;
  free(ix->suf_sid); free(ix->suf_pos); free(ix->buf); free(ix->sent_pos); free(ix->qva);
  free(ix->sentence_length); free(ix->sfreq); free(ix->own_sfreq); free(ix->own_stamp);
  if (ix->real) {
    for (uint64_t i = 0; i < ix->cap_real; i++) real_clear(&ix->real[i]);
    free(ix->real);
  }
  free(ix);
}

static void real_set(fmo_real* r, const char* tokstring, uint32_t n_itoks,
                     const uint64_t* itok_idx, const char* const* itok_str) {
  real_clear(r);
  r->tokstring = strdup(tokstring ? tokstring : "");
  r->n_itoks = n_itoks;
  if (n_itoks) {
    r->itok_idx = (uint64_t*)xrealloc(NULL, n_itoks * sizeof(uint64_t));
    r->itok_str = (char**)xrealloc(NULL, n_itoks * sizeof(char*));
    for (uint32_t i = 0; i < n_itoks; i++) {
      r->itok_idx[i] = itok_idx[i];
      r->itok_str[i] = strdup(itok_str[i]);
    }
  }
}

static void ensure_real(fmo_index* ix, uint64_t n) {
  if (n <= ix->cap_real) return;
  uint64_t nc = ix->cap_real ? ix->cap_real : 16;
  while (nc < n) nc *= 2;
  ix->real = (fmo_real*)xrealloc(ix->real, nc * sizeof(fmo_real));
  memset(ix->real + ix->cap_real, 0, (nc - ix->cap_real) * sizeof(fmo_real));
  ix->cap_real = nc;
}

/* SuffixArray::add_sentence — src/suffix_array.cc:9-27.  Buffer gets [len, tok..., 0]; one
 * SuffixView{sid, i+1} per token (positions are 1-based). */
static int add_ids(fmo_index* ix, const uint32_t* ids, uint32_t n) {
  if (ix->n_sent + 1 > ix->cap_sent) {
    ix->cap_sent = ix->cap_sent ? ix->cap_sent * 2 : 1024;
    ix->sent_pos = (uint32_t*)xrealloc(ix->sent_pos, ix->cap_sent * sizeof(uint32_t));
  }
  if (ix->n_buf + n + 2 > ix->cap_buf) {
    while (ix->n_buf + n + 2 > ix->cap_buf) ix->cap_buf = ix->cap_buf ? ix->cap_buf * 2 : 4096;
    ix->buf = (uint32_t*)xrealloc(ix->buf, ix->cap_buf * sizeof(uint32_t));
  }
  if (ix->n_suf + n > ix->cap_suf) {
    while (ix->n_suf + n > ix->cap_suf) ix->cap_suf = ix->cap_suf ? ix->cap_suf * 2 : 4096;
    ix->suf_sid = (uint32_t*)xrealloc(ix->suf_sid, ix->cap_suf * sizeof(uint32_t));
    ix->suf_pos = (uint16_t*)xrealloc(ix->suf_pos, ix->cap_suf * sizeof(uint16_t));
  }
  uint32_t sidx = (uint32_t)ix->n_sent;
  ix->sent_pos[ix->n_sent++] = (uint32_t)ix->n_buf;
  ix->buf[ix->n_buf++] = n;
  for (uint32_t i = 0; i < n; i++) {
    /* VocabIndexer::addWords — src/vocab_indexer.cc:73-90: sfreq[idx]++ once per distinct word of the sentence */
    const uint32_t w = ids[i];
    if (w >= ix->own_cap) {
      uint32_t nc = ix->own_cap ? ix->own_cap : 1024;
      while (nc <= w) nc *= 2;
      ix->own_sfreq = (uint32_t*)xrealloc(ix->own_sfreq, (size_t)nc * sizeof(uint32_t));
      ix->own_stamp = (uint32_t*)xrealloc(ix->own_stamp, (size_t)nc * sizeof(uint32_t));
      memset(ix->own_sfreq + ix->own_cap, 0, (size_t)(nc - ix->own_cap) * sizeof(uint32_t));
      memset(ix->own_stamp + ix->own_cap, 0, (size_t)(nc - ix->own_cap) * sizeof(uint32_t));
      ix->own_cap = nc;
    }
    if (ix->own_stamp[w] != sidx + 1) { ix->own_stamp[w] = sidx + 1; ix->own_sfreq[w]++; }
    ix->buf[ix->n_buf++] = ids[i];
    ix->suf_sid[ix->n_suf] = sidx;
    ix->suf_pos[ix->n_suf] = (uint16_t)(i + 1);
    ix->n_suf++;
  }
  ix->buf[ix->n_buf++] = 0; /* VocabIndexer::SENTENCE_SEPARATOR */
  ix->sorted = 0;
  return (int)sidx;
}

int fmo_index_add(fmo_index* ix, const uint32_t* ids, uint32_t n, const char* real_tokstring,
                  uint32_t n_itoks, const uint64_t* itok_idx, const char* const* itok_str) {
  /* SuffixArrayIndex::add_tm guard — src/suffix_array_index.cc:16 (norm_tokens.size() <= max) */
  if ((uint64_t)n > ix->max_tokens) return -1;
  int sidx = add_ids(ix, ids, n);
  if (real_tokstring || n_itoks) {
    ix->has_real = 1;
    ensure_real(ix, ix->n_sent);
    real_set(&ix->real[sidx], real_tokstring, n_itoks, itok_idx, itok_str);
  }
  return sidx;
}

int fmo_index_add_bulk(fmo_index* ix, const uint32_t* off, const uint32_t* toks, uint32_t n_sent) {
  for (uint32_t s = 0; s < n_sent; s++) {
    uint32_t n = off[s + 1] - off[s];
    if ((uint64_t)n > ix->max_tokens) continue;
    add_ids(ix, toks + off[s], n);
  }
  return 0;
}

int fmo_index_set_vocab(fmo_index* ix, uint32_t vocab_size, const uint32_t* sfreq) {
  ix->vocab_size = vocab_size;
  free(ix->sfreq);
  ix->sfreq = NULL;
  if (sfreq) {
    ix->sfreq = (uint32_t*)xrealloc(NULL, (size_t)vocab_size * sizeof(uint32_t));
    memcpy(ix->sfreq, sfreq, (size_t)vocab_size * sizeof(uint32_t));
  } else if (ix->own_sfreq && vocab_size) {
    /* no counts handed over: the oracle's own (sentences added through fmo_index_add / _add_bulk) */
    ix->sfreq = (uint32_t*)calloc((size_t)vocab_size, sizeof(uint32_t));
    if (!ix->sfreq) abort();
    memcpy(ix->sfreq, ix->own_sfreq, (size_t)(vocab_size < ix->own_cap ? vocab_size : ix->own_cap) * sizeof(uint32_t));
  }
  return 0;
}

/* VocabIndexer::sfreq for an index adopted with fmo_index_set_arrays, counted HERE from its sentence buffer (src/vocab_indexer.cc:73-90:
 * a word counts once per sentence that holds it) — so that a parity test need not hand the product's counts over */
int fmo_index_count_vocab(fmo_index* ix, uint32_t vocab_size) {
  if (!ix || !ix->buf || !ix->sent_pos) return -1;
  ix->vocab_size = vocab_size;
  free(ix->sfreq);
  ix->sfreq = (uint32_t*)calloc(vocab_size ? vocab_size : 1, sizeof(uint32_t));
  uint32_t* stamp = (uint32_t*)calloc(vocab_size ? vocab_size : 1, sizeof(uint32_t));
  if (!ix->sfreq || !stamp) abort();
  for (uint64_t s = 0; s < ix->n_sent; s++) {
    const uint32_t sp = ix->sent_pos[s], len = ix->buf[sp];
    for (uint32_t k = 0; k < len; k++) {
      const uint32_t w = ix->buf[sp + 1 + k];
      if (w < vocab_size && stamp[w] != (uint32_t)s + 1u) { stamp[w] = (uint32_t)s + 1u; ix->sfreq[w]++; }
    }
  }
  free(stamp);
  return 0;
}

int fmo_index_set_real(fmo_index* ix, uint32_t s_id, const char* real_tokstring, uint32_t n_itoks,
                       const uint64_t* itok_idx, const char* const* itok_str) {
  if (s_id >= ix->n_sent) return -1;
  ix->has_real = 1;
  ensure_real(ix, ix->n_sent);
  real_set(&ix->real[s_id], real_tokstring, n_itoks, itok_idx, itok_str);
  return 0;
}

/* fmo_index_set_real for sentences [first_sid, first_sid + n) without penalty tokens: sentence i's tab-joined real tokens are the
 * NUL-terminated string at packed + off[i] (what SuffixArrayIndex::add_tm stores per sentence, src/suffix_array_index.cc:61-70) */
int fmo_index_set_real_bulk(fmo_index* ix, uint32_t first_sid, uint32_t n, const char* packed, const uint64_t* off) {
  if (!ix || !packed || !off || (uint64_t)first_sid + n > ix->n_sent) return -1;
  ix->has_real = 1;
  ensure_real(ix, ix->n_sent);
  for (uint32_t i = 0; i < n; i++) real_set(&ix->real[first_sid + i], packed + off[i], 0, NULL, NULL);
  return 0;
}

/* SuffixArray::get_sentence — include/fuzzy/suffix_array.hxx:11-19 */
static inline const uint32_t* get_sentence(const fmo_index* ix, uint64_t sid, uint64_t* length) {
  const uint32_t* s = ix->buf + ix->sent_pos[sid];
  if (length) *length = *s;
  return s + 1;
}

/* SuffixArray::get_suffix — include/fuzzy/suffix_array.hxx:21-29 */
static inline const uint32_t* get_suffix(const fmo_index* ix, uint32_t sid, uint16_t pos,
                                         uint64_t* length) {
  const uint32_t* s = get_sentence(ix, sid, length);
  uint64_t prefix_length = (uint64_t)pos - 1;
  if (length) *length -= prefix_length;
  return s + prefix_length;
}

/* compare_ngrams — src/suffix_array.cc:214-233; *cmp_count = tokens compared (for byte accounting) */
static inline int compare_ngrams(const uint32_t* v1, uint64_t n1, const uint32_t* v2, uint64_t n2,
                                 int equal_if_startby, uint64_t* cmp_count) {
  uint64_t mn = n1 < n2 ? n1 : n2;
  for (uint64_t i = 0; i < mn; i++) {
    if (v1[i] < v2[i]) { if (cmp_count) *cmp_count = i + 1; return -1; }
    if (v1[i] > v2[i]) { if (cmp_count) *cmp_count = i + 1; return 1; }
  }
  if (cmp_count) *cmp_count = mn ? mn : 1;
  if (n1 < n2) return -1;
  if (n1 > n2 && !equal_if_startby) return 1;
  return 0;
}

/* SuffixArray::comp — src/suffix_array.cc:235-251 */
static inline int suffix_comp(const fmo_index* ix, uint32_t sid_a, uint16_t pos_a, uint32_t sid_b,
                              uint16_t pos_b) {
  uint64_t la, lb;
  const uint32_t* a = get_suffix(ix, sid_a, pos_a, &la);
  const uint32_t* b = get_suffix(ix, sid_b, pos_b, &lb);
  int c = compare_ngrams(a, la, b, lb, 0, NULL);
  if (c != 0 || sid_a == sid_b) return c;
  return sid_a < sid_b ? -1 : 1;
}

static int qsort_comp(const void* pa, const void* pb, void* arg) {
  const fmo_index* ix = (const fmo_index*)arg;
  uint64_t a = *(const uint64_t*)pa, b = *(const uint64_t*)pb;
  return suffix_comp(ix, (uint32_t)(a >> 16), (uint16_t)(a & 0xffff), (uint32_t)(b >> 16),
                     (uint16_t)(b & 0xffff));
}

/* SuffixArray::compute_sentence_length — src/suffix_array.cc:253-261 */
static void compute_sentence_length(fmo_index* ix) {
  ix->sentence_length = (uint16_t*)xrealloc(ix->sentence_length, ix->n_suf * sizeof(uint16_t));
  for (uint64_t i = 0; i < ix->n_suf; i++)
    ix->sentence_length[i] = (uint16_t)ix->buf[ix->sent_pos[ix->suf_sid[i]]];
}

typedef struct sort_job {
  const fmo_index* ix;
  uint64_t*        keys;
  const uint64_t*  bucket_off;
  const uint32_t*  order; /* buckets, largest first */
  uint32_t         n_buckets;
  atomic_uint*     next;
} sort_job;

static void* sort_worker(void* arg) {
  sort_job* j = (sort_job*)arg;
  for (;;) {
    unsigned k = atomic_fetch_add(j->next, 1u);
    if (k >= j->n_buckets) break;
    uint32_t w = j->order[k];
    uint64_t lo = j->bucket_off[w], hi = j->bucket_off[w + 1];
    if (hi - lo > 1) qsort_r(j->keys + lo, hi - lo, sizeof(uint64_t), qsort_comp, (void*)j->ix);
  }
  return NULL;
}

static const uint64_t* g_bucket_off_for_order;
static int order_comp(const void* a, const void* b) {
  uint32_t x = *(const uint32_t*)a, y = *(const uint32_t*)b;
  uint64_t sx = g_bucket_off_for_order[x + 1] - g_bucket_off_for_order[x];
  uint64_t sy = g_bucket_off_for_order[y + 1] - g_bucket_off_for_order[y];
  return sx < sy ? 1 : sx > sy ? -1 : (x < y ? -1 : x > y);
}

/* SuffixArray::sort — src/suffix_array.cc:58-102: bucket by first word id, sort each bucket with
 * comp, concatenate; _quickVocabAccess[wid] = first suffix of the bucket, [vocab_size] = total. */
int fmo_index_sort(fmo_index* ix, int nthreads) {
  if (ix->sorted) return 0;
  uint32_t V = ix->vocab_size;
  for (uint64_t i = 0; i < ix->n_suf; i++) {
    uint32_t w = *get_suffix(ix, ix->suf_sid[i], ix->suf_pos[i], NULL);
    if (w >= V) V = w + 1; /* the reference asserts wid < vocab_size; be total instead */
  }
  if (V > ix->vocab_size && ix->sfreq) {
    ix->sfreq = (uint32_t*)xrealloc(ix->sfreq, (size_t)V * sizeof(uint32_t));
    memset(ix->sfreq + ix->vocab_size, 0, (size_t)(V - ix->vocab_size) * sizeof(uint32_t));
  }
  ix->vocab_size = V;
  uint64_t* off = (uint64_t*)calloc((size_t)V + 2, sizeof(uint64_t));
  for (uint64_t i = 0; i < ix->n_suf; i++)
    off[*get_suffix(ix, ix->suf_sid[i], ix->suf_pos[i], NULL) + 1]++;
  for (uint32_t w = 0; w < V; w++) off[w + 1] += off[w];
  uint64_t* keys = (uint64_t*)xrealloc(NULL, ix->n_suf * sizeof(uint64_t));
  uint64_t* cur = (uint64_t*)xrealloc(NULL, ((size_t)V + 1) * sizeof(uint64_t));
  memcpy(cur, off, ((size_t)V + 1) * sizeof(uint64_t));
  for (uint64_t i = 0; i < ix->n_suf; i++) {
    uint32_t w = *get_suffix(ix, ix->suf_sid[i], ix->suf_pos[i], NULL);
    keys[cur[w]++] = ((uint64_t)ix->suf_sid[i] << 16) | ix->suf_pos[i];
  }
  free(cur);
  uint32_t* order = (uint32_t*)xrealloc(NULL, ((size_t)V + 1) * sizeof(uint32_t));
  for (uint32_t w = 0; w < V; w++) order[w] = w;
  g_bucket_off_for_order = off;
  qsort(order, V, sizeof(uint32_t), order_comp);
  atomic_uint next;
  atomic_init(&next, 0);
  sort_job job = {ix, keys, off, order, V, &next};
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  pthread_t th[256];
  for (int t = 1; t < nthreads; t++) pthread_create(&th[t], NULL, sort_worker, &job);
  sort_worker(&job);
  for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
  for (uint64_t i = 0; i < ix->n_suf; i++) {
    ix->suf_sid[i] = (uint32_t)(keys[i] >> 16);
    ix->suf_pos[i] = (uint16_t)(keys[i] & 0xffff);
  }
  ix->qva = (uint32_t*)xrealloc(ix->qva, ((size_t)V + 1) * sizeof(uint32_t));
  for (uint32_t w = 0; w <= V; w++) ix->qva[w] = (uint32_t)off[w];
  ix->n_qva = (uint64_t)V + 1;
  free(order); free(keys); free(off);
  ix->sorted = 1;
  compute_sentence_length(ix);
  return 0;
}

int fmo_index_set_arrays(fmo_index* ix, const uint32_t* suf_sid, const uint16_t* suf_pos,
                         uint64_t n_suf, const uint32_t* sentence_buffer, uint64_t n_buf,
                         const uint32_t* sentence_pos, uint64_t n_sent, const uint32_t* qva,
                         uint64_t n_qva) {
  ix->suf_sid = (uint32_t*)xrealloc(ix->suf_sid, n_suf * sizeof(uint32_t));
  ix->suf_pos = (uint16_t*)xrealloc(ix->suf_pos, n_suf * sizeof(uint16_t));
  memcpy(ix->suf_sid, suf_sid, n_suf * sizeof(uint32_t));
  memcpy(ix->suf_pos, suf_pos, n_suf * sizeof(uint16_t));
  ix->n_suf = ix->cap_suf = n_suf;
  ix->buf = (uint32_t*)xrealloc(ix->buf, n_buf * sizeof(uint32_t));
  memcpy(ix->buf, sentence_buffer, n_buf * sizeof(uint32_t));
  ix->n_buf = ix->cap_buf = n_buf;
  ix->sent_pos = (uint32_t*)xrealloc(ix->sent_pos, n_sent * sizeof(uint32_t));
  memcpy(ix->sent_pos, sentence_pos, n_sent * sizeof(uint32_t));
  ix->n_sent = ix->cap_sent = n_sent;
  ix->qva = (uint32_t*)xrealloc(ix->qva, n_qva * sizeof(uint32_t));
  memcpy(ix->qva, qva, n_qva * sizeof(uint32_t));
  ix->n_qva = n_qva;
  ix->sorted = 1;
  compute_sentence_length(ix);
  return 0;
}

uint64_t        fmo_index_num_suffixes(const fmo_index* ix) { return ix->n_suf; }
uint64_t        fmo_index_num_sentences(const fmo_index* ix) { return ix->n_sent; }
uint64_t        fmo_index_buffer_size(const fmo_index* ix) { return ix->n_buf; }
uint64_t        fmo_index_qva_size(const fmo_index* ix) { return ix->n_qva; }
const uint32_t* fmo_index_suf_sid(const fmo_index* ix) { return ix->suf_sid; }
const uint16_t* fmo_index_suf_pos(const fmo_index* ix) { return ix->suf_pos; }
const uint32_t* fmo_index_buffer(const fmo_index* ix) { return ix->buf; }
const uint32_t* fmo_index_sentence_pos(const fmo_index* ix) { return ix->sent_pos; }
const uint32_t* fmo_index_qva(const fmo_index* ix) { return ix->qva; }

/* ------------------------------------------------------------------------------------------ */
/* probe                                                                                       */
/* ------------------------------------------------------------------------------------------ */

/* SuffixArray::start_by — src/suffix_array.cc:263-273.  Accounting per SURVEY §8(d):
 * 6 B SuffixView + 4 B _sentence_pos + 4 B length word + 4 B per token compared. */
static inline int start_by(const fmo_index* ix, uint64_t i, const uint32_t* ngram, uint64_t length,
                           fmo_counters* ctr) {
  uint64_t sl, c = 0;
  const uint32_t* suf = get_suffix(ix, ix->suf_sid[i], ix->suf_pos[i], &sl);
  int r = compare_ngrams(suf, sl, ngram, length, 1, &c);
  if (ctr) {
    ctr->compare_steps++;
    ctr->probe_bytes += 6 + 4 + 4 + 4 * c;
  }
  return r;
}

/* SuffixArray::equal_range — src/suffix_array.cc:105-212 (three phases: any hit, upper, lower) */
void fmo_equal_range(const fmo_index* ix, const uint32_t* ngram, uint64_t length, uint64_t min,
                     uint64_t max, uint64_t out[2], fmo_counters* ctr) {
  if (ctr) ctr->equal_range_calls++;
  out[0] = out[1] = 0;
  if (length == 0) return;
  if (max == 0) {
    if ((uint64_t)ngram[0] > ix->n_qva - 1) return;
    min = ix->qva[ngram[0]];
    if ((uint64_t)(uint32_t)(ngram[0] + 1) < ix->n_qva - 1)
      max = ix->qva[ngram[0] + 1];
    else
      max = ix->n_suf;
    if (length == 1) { out[0] = min; out[1] = max; return; }
  }
  uint64_t cur = (min + max) / 2;
  int r;
  while (max > min) {
    r = start_by(ix, cur, ngram, length, ctr);
    if (r == 0) break;
    if (r < 0) {
      if (min < cur) min = cur; else max = min;
    } else
      max = cur;
    cur = (min + max) / 2;
  }
  if (max - min < 2) { out[0] = min; out[1] = max; return; }
  uint64_t savecur = cur, savemin = min;
  min = cur;
  cur = (min + max) / 2;
  while (cur > min) {
    r = start_by(ix, cur, ngram, length, ctr);
    if (r == 0) min = cur; else max = cur;
    cur = (min + max) / 2;
  }
  out[1] = max;
  min = savemin;
  if (start_by(ix, min, ngram, length, ctr) == 0)
    max = min;
  else
    max = savecur;
  cur = (min + max) / 2;
  while (cur > min) {
    r = start_by(ix, cur, ngram, length, ctr);
    if (r == 0) max = cur; else min = cur;
    cur = (min + max) / 2;
  }
  out[0] = max;
}

/* ------------------------------------------------------------------------------------------ */
/* costs / bounds                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* Costs::get_normalizer — include/fuzzy/costs.hh:33-47 */
float fmo_normalizer(uint64_t p, uint64_t s, float ins, float del, float rep) {
  if (ins == 0. && del == 0. && rep == 0.) return 1.f;
  if (ins + del <= rep) return ins * (float)p + del * (float)s;
  else if (p <= s) return (rep - del) * (float)p + del * (float)s;
  else return (rep - ins) * (float)s + ins * (float)p;
}

/* NGramMatches::theoretical_rejection — src/ngram_matches.cc:31-39 */
static inline int theoretical_rejection(uint64_t p, uint64_t s, float ins, float del, float rep,
                                        float fuzzy) {
  const float size_difference = fabsf((float)p - (float)s);
  float remaining_cost = (p >= s) ? ins : del;
  float theoretical_bound = 1.f - remaining_cost * size_difference / fmo_normalizer(p, s, ins, del, rep);
  return (double)theoretical_bound + 0.000005 < (double)fuzzy;
}

/* NGramMatches::theoretical_rejection_cover — src/ngram_matches.cc:41-59 */
static inline int theoretical_rejection_cover(uint64_t p, uint64_t s, uint64_t cover, float ins,
                                              float del, float rep, float fuzzy) {
  float theoretical_bound;
  if (ins + del < rep) {
    theoretical_bound = 1.f - (ins * ((float)s - (float)cover) + del * ((float)p - (float)cover)) /
                                  fmo_normalizer(p, s, ins, del, rep);
  } else {
    float cost_remaining = (p > s) ? ins : del;
    float min_length = (p > s) ? (float)s : (float)p;
    float max_length = (p > s) ? (float)p : (float)s;
    theoretical_bound = 1.f - (rep * (min_length - (float)cover) + cost_remaining * (max_length - min_length)) /
                                  fmo_normalizer(p, s, ins, del, rep);
  }
  return (double)theoretical_bound + 0.000005 < (double)fuzzy;
}

/* score quantisation — src/fuzzy_match.cc:598,651 */
float fmo_score_from_cost(float cost) {
  return (float)((int)(10000 - cost * 100) / 10000.0);
}

/* ------------------------------------------------------------------------------------------ */
/* edit distances                                                                              */
/* ------------------------------------------------------------------------------------------ */

static inline float min3f(float a, float b, float c) {
  float m = a < b ? a : b; /* std::min({a,b,c}) is value-only */
  return c < m ? c : m;
}

/* _edit_distance_char — include/fuzzy/edit_distance.hxx:7-35 (byte Levenshtein) */
int fmo_edit_distance_char(const char* s1, int n1, const char* s2, int n2) {
  if (n1 == 0) return n2;
  if (n2 == 0) return n1;
  int  stack_row[128];
  int* prev = (n2 + 1) * 2 <= 128 ? stack_row : (int*)xrealloc(NULL, (size_t)(n2 + 1) * 2 * sizeof(int));
  int* curr = prev + (n2 + 1);
  for (int j = 0; j <= n2; j++) prev[j] = j;
  for (int i = 1; i <= n1; i++) {
    curr[0] = i;
    for (int j = 1; j <= n2; j++) {
      int a = prev[j] + 1, b = curr[j - 1] + 1, c = prev[j - 1] + (s1[i - 1] == s2[j - 1] ? 0 : 1);
      int m = a < b ? a : b;
      curr[j] = c < m ? c : m;
    }
    int* t = prev; prev = curr; curr = t;
  }
  int res = prev[n2];
  int* base = prev < curr ? prev : curr;
  if (base != stack_row) free(base);
  return res;
}

/* ids-only _edit_distance — src/edit_distance.cc:79-122 */
float fmo_edit_distance_ids(const uint32_t* s1, int n1, const uint32_t* s2, int n2, float ins,
                            float del, float rep, float diff_word, float max_fuzzyness) {
  float* arr = (float*)calloc((size_t)(n1 + 1) * (size_t)(n2 + 1), sizeof(float));
  const size_t W = (size_t)n2 + 1;
  for (int i = 1; i < n1 + 1; i++) arr[i * W] = arr[(i - 1) * W] + diff_word * del;
  for (int j = 1; j < n2 + 1; j++) arr[j] = arr[j - 1] + diff_word * ins;
  float result = 0;
  int   early = 0;
  for (int i = 1; i < n1 + 1 && !early; i++) {
    float mn = FLT_MAX;
    for (int j = 1; j < n2 + 1; j++) {
      float diff = 0;
      if (s1[i - 1] != s2[j - 1]) diff = rep * diff_word;
      const float distance = min3f(arr[(i - 1) * W + j] + del * diff_word,
                                   arr[i * W + j - 1] + ins * diff_word,
                                   arr[(i - 1) * W + j - 1] + diff);
      arr[i * W + j] = distance;
      mn = mn < distance ? mn : distance;
    }
    if (mn > max_fuzzyness) { result = mn; early = 1; }
  }
  if (!early) result = arr[(size_t)n1 * W + n2];
  free(arr);
  return result;
}

typedef struct strview { const char* p; int n; } strview;

/* Sentence::operator Tokens / split_string — src/sentence.cc:7-22,36-38: split on '\t', the
 * trailing part is dropped when empty. */
static int split_tokstring(const char* s, strview** out, int* cap) {
  int n = 0;
  const char* start = s;
  const char* c = s;
  for (;; c++) {
    if (*c == '\t' || *c == '\0') {
      int len = (int)(c - start);
      if (*c == '\t' || len > 0) {
        if (n + 1 > *cap) {
          *cap = *cap ? *cap * 2 : 64;
          *out = (strview*)xrealloc(*out, (size_t)*cap * sizeof(strview));
        }
        (*out)[n].p = start; (*out)[n].n = len; n++;
      }
      if (*c == '\0') break;
      start = c + 1;
    }
  }
  return n;
}

/* Sentence::get_itoks — src/sentence.cc:27-34 (st/sn sized n+1, zero-initialised by callers) */
static void get_itoks(const fmo_real* r, const char** st, int* sn, int n_slots) {
  if (!r) return;
  for (uint32_t k = 0; k < r->n_itoks; k++) {
    uint64_t idx = r->itok_idx[k];
    if (idx >= (uint64_t)n_slots) continue; /* reference would write out of bounds */
    sn[idx] = (int)strlen(r->itok_str[k]);
    st[idx] = r->itok_str[k];
  }
}

typedef struct dp_scratch {
  float*       arr;
  int*         tag;
  size_t       cap_cells;
  const char** st1;
  int*         sn1;
  int          cap1;
  strview*     real1;
  int          cap_real1;
} dp_scratch;

/* full _edit_distance — src/edit_distance.cc:5-77.  s1/real1/n1 = indexed sentence, s2/real2/n2 =
 * pattern.  compare_real == 0 means "real tokens are the normalised tokens on both sides", in
 * which case equal ids imply equal real tokens and the string test is skipped. */
static float edit_distance_full(const uint32_t* s1, const fmo_real* real1, int n1,
                                const uint32_t* s2, const strview* real2tok, int n_real2, int n2,
                                const char* const* st2, const int* sn2, const float* idf_penalty,
                                float idf_weight, float ins, float del, float rep, float diff_word,
                                float max_fuzzyness, int compare_real, dp_scratch* sc,
                                fmo_counters* ctr) {
  const size_t W = (size_t)n2 + 1;
  const size_t cells = (size_t)(n1 + 1) * W;
  if (cells > sc->cap_cells) {
    sc->cap_cells = cells * 2;
    sc->arr = (float*)xrealloc(sc->arr, sc->cap_cells * sizeof(float));
    sc->tag = (int*)xrealloc(sc->tag, sc->cap_cells * sizeof(int));
  }
  float* arr = sc->arr;
  int*   cost_tag = sc->tag;
  if (n1 + 1 > sc->cap1) {
    sc->cap1 = (n1 + 1) * 2;
    sc->st1 = (const char**)xrealloc((void*)sc->st1, (size_t)sc->cap1 * sizeof(char*));
    sc->sn1 = (int*)xrealloc(sc->sn1, (size_t)sc->cap1 * sizeof(int));
  }
  const char** st1 = sc->st1;
  int*         sn1 = sc->sn1;
  for (int i = 0; i <= n1; i++) { st1[i] = NULL; sn1[i] = 0; }
  get_itoks(real1, st1, sn1, n1 + 1);
  int n_real1 = 0;
  if (compare_real && real1 && real1->tokstring)
    n_real1 = split_tokstring(real1->tokstring, &sc->real1, &sc->cap_real1);
  const strview* real1tok = sc->real1;
  if (ctr) { ctr->dp_calls++; }

  arr[0] = (float)fmo_edit_distance_char(st1[n1], sn1[n1], st2[n2], sn2[n2]);
  cost_tag[0] = fmo_edit_distance_char(st1[0], sn1[0], st2[0], sn2[0]);
  for (int i = 1; i < n1 + 1; i++) {
    arr[i * W] = arr[(i - 1) * W] + diff_word * del + sn1[i];
    cost_tag[i * W] = fmo_edit_distance_char(st1[i], sn1[i], st2[0], sn2[0]);
  }
  for (int j = 1; j < n2 + 1; j++) {
    arr[j] = arr[j - 1] + diff_word * ins + sn2[j];
    if (idf_weight) arr[j] += idf_penalty[j - 1] * idf_weight;
    cost_tag[j] = fmo_edit_distance_char(st1[0], sn1[0], st2[j], sn2[j]);
  }
  for (int i = 1; i < n1 + 1; i++) {
    float mn = FLT_MAX;
    for (int j = 1; j < n2 + 1; j++) {
      float diff = 0;
      float penalty_j1 = 0;
      if (idf_weight) penalty_j1 = idf_penalty[j - 1] * idf_weight;
      if (s1[i - 1] != s2[j - 1]) {
        diff = rep * diff_word + penalty_j1;
      } else if (compare_real) {
        /* out-of-range real tokens (trailing empty token dropped by the split) read as "" */
        strview a = (i - 1 < n_real1) ? real1tok[i - 1] : (strview){"", 0};
        strview b = (j - 1 < n_real2) ? real2tok[j - 1] : (strview){"", 0};
        if (a.n != b.n || memcmp(a.p, b.p, (size_t)a.n) != 0) {
          char c0 = a.n ? a.p[0] : '\0';
          if (strchr("LUMC", c0)) /* '\0' matches the terminator, as in the reference */
            diff = rep * 1.0f;   /* costs.diff_case, costs.hh:63 */
          else
            diff = rep * 2.0f;   /* costs.diff_real, costs.hh:61 */
        }
      }
      cost_tag[i * W + j] = fmo_edit_distance_char(st1[i], sn1[i], st2[j], sn2[j]);
      const float distance =
          min3f(arr[(i - 1) * W + j] + del * diff_word + cost_tag[(i - 1) * W + j],
                arr[i * W + j - 1] + ins * diff_word + cost_tag[i * W + j - 1] + penalty_j1,
                arr[(i - 1) * W + j - 1] + diff + cost_tag[(i - 1) * W + j - 1]);
      arr[i * W + j] = distance;
      mn = mn < distance ? mn : distance;
    }
    if (ctr) ctr->dp_cells += (uint64_t)n2;
    if (mn > max_fuzzyness) return mn;
  }
  return arr[(size_t)n1 * W + n2];
}

/* ------------------------------------------------------------------------------------------ */
/* per-thread scratch                                                                          */
/* ------------------------------------------------------------------------------------------ */

typedef struct cand { uint32_t sid; uint32_t longest; } cand;

typedef struct acc_match {
  float    score, penalty;
  int32_t  max_subseq;
  uint32_t s_id;
  int      length;
  const uint32_t* s;
} acc_match;

typedef struct fmo_scratch {
  /* LongestMatches (ngram_matches.hh:20) as an open-addressing table keyed by sid+1 */
  uint32_t*  hkey;
  uint32_t*  hval;
  uint32_t*  hused;     /* slots occupied by the current query, for O(count) clearing */
  uint64_t   hcap, hcount;
  cand*      cands;
  uint64_t   cap_cands;
  float*     heap;       /* lowest_costs max-heap */
  uint64_t   heap_n, heap_cap;
  acc_match* result;
  uint64_t   n_result, cap_result;
  float*     idf;
  uint32_t*  cov_word;
  uint32_t*  cov_count;
  uint64_t   cap_p;
  const char** st2;
  int*       sn2;
  strview*   real2;
  int        cap_real2;
  dp_scratch dp;
  float*     memo;       /* contrastive pair memo, n x n, NaN = unset */
  uint64_t   cap_memo;
  float*     pen_sum;
  float*     pen_max;
  uint32_t*  pen_cnt;
  /* entries the caller's `matches` vector holds on entry (fmo_match_prior): token sequences as CSR */
  uint32_t        n_prior;
  const uint32_t* prior_off;
  const uint32_t* prior_tok;
  const uint32_t* prior_sid;  /* Match::s_id of those entries (the rerank's memo is keyed on ids), or NULL */
} fmo_scratch;

static void scratch_free(fmo_scratch* s) {
  free(s->hkey); free(s->hval); free(s->hused); free(s->cands); free(s->heap); free(s->result); free(s->idf);
  free(s->cov_word); free(s->cov_count); free((void*)s->st2); free(s->sn2); free(s->real2);
  free(s->dp.arr); free(s->dp.tag); free((void*)s->dp.st1); free(s->dp.sn1); free(s->dp.real1);
  free(s->memo); free(s->pen_sum); free(s->pen_max); free(s->pen_cnt);
}

static inline uint32_t int_hash(uint32_t x) { /* IntHash, ngram_matches.hh:9-17 */
  x = ((x >> 16) ^ x) * 0x45d9f3b;
  x = ((x >> 16) ^ x) * 0x45d9f3b;
  x = (x >> 16) ^ x;
  return x;
}

static void hash_reset(fmo_scratch* s) {
  /* a fresh map per query, like the reference's NGramMatches member.  Only the slots the previous
   * query touched are cleared: a table grown by one wide query must not cost a full memset (or an
   * mmap/munmap pair, which serialises all worker threads in the kernel) for every later query. */
  if (!s->hcap) {
    s->hcap = 1024;
    s->hkey = (uint32_t*)calloc(s->hcap, sizeof(uint32_t));
    s->hval = (uint32_t*)xrealloc(NULL, s->hcap * sizeof(uint32_t));
    s->hused = (uint32_t*)xrealloc(NULL, s->hcap * sizeof(uint32_t));
  } else {
    for (uint64_t i = 0; i < s->hcount; i++) s->hkey[s->hused[i]] = 0;
  }
  s->hcount = 0;
}

static void hash_grow(fmo_scratch* s) {
  uint64_t  oc = s->hcap;
  uint32_t *ok = s->hkey, *ov = s->hval;
  s->hcap = oc * 2;
  s->hkey = (uint32_t*)calloc(s->hcap, sizeof(uint32_t));
  s->hval = (uint32_t*)xrealloc(NULL, s->hcap * sizeof(uint32_t));
  s->hused = (uint32_t*)xrealloc(s->hused, s->hcap * sizeof(uint32_t));
  uint64_t n = 0;
  for (uint64_t i = 0; i < oc; i++) {
    if (!ok[i]) continue;
    uint64_t h = int_hash(ok[i] - 1) & (s->hcap - 1);
    while (s->hkey[h]) h = (h + 1) & (s->hcap - 1);
    s->hkey[h] = ok[i];
    s->hval[h] = ov[i];
    s->hused[n++] = (uint32_t)h;
  }
  free(ok); free(ov);
}

/* try_emplace(sid, L).first.value() = max(value, L) — src/ngram_matches.cc:79-81 */
static inline void hash_upsert_max(fmo_scratch* s, uint32_t sid, uint32_t L) {
  if ((s->hcount + 1) * 2 > s->hcap) hash_grow(s);
  uint64_t h = int_hash(sid) & (s->hcap - 1);
  while (s->hkey[h] && s->hkey[h] != sid + 1) h = (h + 1) & (s->hcap - 1);
  if (!s->hkey[h]) {
    s->hkey[h] = sid + 1;
    s->hval[h] = L;
    s->hused[s->hcount++] = (uint32_t)h;
  } else if (s->hval[h] < L)
    s->hval[h] = L;
}

/* NGramMatches::register_suffix_range_match — src/ngram_matches.cc:61-83 */
static void register_range(const fmo_index* ix, fmo_scratch* s, uint64_t begin, uint64_t end,
                           unsigned match_length, unsigned min_seq_len, unsigned p_length,
                           const fmo_params* prm, fmo_counters* ctr) {
  if (match_length < min_seq_len) return;
  for (uint64_t i = begin; i < end; i++) {
    const uint64_t s_length = ix->sentence_length[i];
    if (ctr) { ctr->suffix_visits++; ctr->reg_bytes += 2; }
    if (theoretical_rejection(p_length, s_length, prm->insert_cost, prm->delete_cost,
                              prm->replace_cost, prm->fuzzy))
      continue;
    if (ctr) { ctr->suffix_kept++; ctr->reg_bytes += 4; }
    hash_upsert_max(s, ix->suf_sid[i], match_length);
  }
}

/* get_longest_matches order — src/ngram_matches.cc:19-29: (longest desc, s_id asc) */
static int cand_comp(const void* pa, const void* pb) {
  const cand* a = (const cand*)pa;
  const cand* b = (const cand*)pb;
  if (a->longest != b->longest) return a->longest > b->longest ? -1 : 1;
  return a->sid < b->sid ? -1 : a->sid > b->sid;
}

/* in-place sort of the candidate list (heapsort): glibc's qsort() mallocs a merge buffer, which for the
 * wide queries means an mmap/munmap pair per query and serialises the worker threads in the kernel */
static inline int cand_less(const cand* a, const cand* b) { return cand_comp(a, b) < 0; }

static void cand_sift(cand* a, size_t root, size_t n) {
  for (;;) {
    size_t l = 2 * root + 1, r = l + 1, mx = root;
    if (l < n && cand_less(&a[mx], &a[l])) mx = l;
    if (r < n && cand_less(&a[mx], &a[r])) mx = r;
    if (mx == root) return;
    cand t = a[mx]; a[mx] = a[root]; a[root] = t;
    root = mx;
  }
}

static void cand_sort(cand* a, size_t n) {
  if (n < 2) return;
  if (n <= 16) {
    for (size_t i = 1; i < n; i++) {
      cand v = a[i];
      size_t j = i;
      while (j > 0 && cand_less(&v, &a[j - 1])) { a[j] = a[j - 1]; j--; }
      a[j] = v;
    }
    return;
  }
  for (size_t i = n / 2; i-- > 0;) cand_sift(a, i, n);
  for (size_t e = n - 1; e > 0; e--) {
    cand t = a[0]; a[0] = a[e]; a[e] = t;
    cand_sift(a, 0, e);
  }
}

/* std::priority_queue<float> — max-heap */
static void heap_push(fmo_scratch* s, float v) {
  if (s->heap_n + 1 > s->heap_cap) {
    s->heap_cap = s->heap_cap ? s->heap_cap * 2 : 64;
    s->heap = (float*)xrealloc(s->heap, s->heap_cap * sizeof(float));
  }
  uint64_t i = s->heap_n++;
  s->heap[i] = v;
  while (i > 0) {
    uint64_t par = (i - 1) / 2;
    if (s->heap[par] < s->heap[i]) {
      float t = s->heap[par]; s->heap[par] = s->heap[i]; s->heap[i] = t;
      i = par;
    } else
      break;
  }
}

static void heap_pop(fmo_scratch* s) {
  if (!s->heap_n) return;
  s->heap[0] = s->heap[--s->heap_n];
  uint64_t i = 0;
  for (;;) {
    uint64_t l = 2 * i + 1, r = l + 1, m = i;
    if (l < s->heap_n && s->heap[m] < s->heap[l]) m = l;
    if (r < s->heap_n && s->heap[m] < s->heap[r]) m = r;
    if (m == i) break;
    float t = s->heap[m]; s->heap[m] = s->heap[i]; s->heap[i] = t;
    i = m;
  }
}

/* CompareMatch — src/fuzzy_match.cc:25-33: pop order is (score desc, s_id asc) */
static int result_comp(const void* pa, const void* pb) {
  const acc_match* a = (const acc_match*)pa;
  const acc_match* b = (const acc_match*)pb;
  if (a->score != b->score) return a->score > b->score ? -1 : 1;
  return a->s_id < b->s_id ? -1 : a->s_id > b->s_id;
}

/* ------------------------------------------------------------------------------------------ */
/* match()                                                                                     */
/* ------------------------------------------------------------------------------------------ */

static int64_t match_impl(const fmo_index* ix, const fmo_params* prm, const uint32_t* pattern_wids,
                          uint32_t p, const fmo_real* real, fmo_rec* out, uint64_t cap,
                          fmo_counters* ctr, fmo_scratch* sc) {
  /* A. guards — src/fuzzy_match.cc:450-467 */
  const uint64_t p_length = p;
  const unsigned number_of_matches = prm->nmatch;
  int contrast_buffer = prm->contrast_buffer;
  if (contrast_buffer == -1) contrast_buffer = (int)number_of_matches;
  if (ctr) { ctr->queries++; ctr->out_bytes += 16ull * number_of_matches; }
  if (p_length > ix->max_tokens) return 0;
  if (!p_length) return 0;
  int min_subseq_length = prm->min_subseq_length;
  if ((size_t)(min_subseq_length) > p_length) min_subseq_length = (int)p_length;
  if ((int)(prm->min_subseq_ratio * (float)p_length) > min_subseq_length)
    min_subseq_length = (int)(prm->min_subseq_ratio * (float)p_length);
  const unsigned min_seq_len = (unsigned)min_subseq_length;
  const float ins = prm->insert_cost, del = prm->delete_cost, rep = prm->replace_cost;
  const float fuzzy = prm->fuzzy;

  if (p_length + 1 > sc->cap_p) {
    sc->cap_p = (p_length + 1) * 2;
    sc->idf = (float*)xrealloc(sc->idf, sc->cap_p * sizeof(float));
    sc->cov_word = (uint32_t*)xrealloc(sc->cov_word, sc->cap_p * sizeof(uint32_t));
    sc->cov_count = (uint32_t*)xrealloc(sc->cov_count, sc->cap_p * sizeof(uint32_t));
    sc->st2 = (const char**)xrealloc((void*)sc->st2, sc->cap_p * sizeof(char*));
    sc->sn2 = (int*)xrealloc(sc->sn2, sc->cap_p * sizeof(int));
  }

  /* C. idf — src/fuzzy_match.cc:367-390,472-477 */
  float idf_max = 0.01f;
  const float vocab_idf_penalty = prm->idf_penalty;
  if (vocab_idf_penalty) {
    const unsigned num_sentences = (unsigned)ix->n_sent;
    for (uint64_t j = 0; j < p_length; j++) {
      uint32_t wid = pattern_wids[j];
      if (wid != 1 /* VOCAB_UNK */) {
        uint32_t f = (ix->sfreq && wid < ix->vocab_size) ? ix->sfreq[wid] : 0;
        sc->idf[j] = logf((float)num_sentences / (float)f);
      } else
        sc->idf[j] = 0;
    }
    idf_max = (float)log((double)num_sentences);
  }

  /* D. probe loop — src/fuzzy_match.cc:484-551 */
  hash_reset(sc);
  uint64_t range[2];
  if (p_length == 1) {
    fmo_equal_range(ix, pattern_wids, p_length, 0, 0, range, ctr);
    if (range[0] != range[1])
      register_range(ix, sc, range[0], range[1], (unsigned)p_length, min_seq_len, (unsigned)p_length, prm, ctr);
  }
  for (uint64_t it = 0; it < p_length; it++) {
    uint64_t prev[2] = {0, 0};
    uint64_t subseq_length = 0;
    for (uint64_t jt = it; jt < p_length; jt++) {
      ++subseq_length;
      fmo_equal_range(ix, pattern_wids + it, subseq_length, prev[0], prev[1], range, ctr);
      if (range[0] != range[1]) {
        if (subseq_length > 2) {
          register_range(ix, sc, prev[0], range[0], (unsigned)(subseq_length - 1), min_seq_len, (unsigned)p_length, prm, ctr);
          register_range(ix, sc, range[1], prev[1], (unsigned)(subseq_length - 1), min_seq_len, (unsigned)p_length, prm, ctr);
        }
        prev[0] = range[0]; prev[1] = range[1];
      } else {
        --subseq_length;
        break;
      }
    }
    if (subseq_length >= 2)
      register_range(ix, sc, prev[0], prev[1], (unsigned)subseq_length, min_seq_len, (unsigned)p_length, prm, ctr);
  }

  /* E. candidate order — src/ngram_matches.cc:19-29 */
  if (sc->hcount > sc->cap_cands) {
    sc->cap_cands = sc->hcount * 2;
    sc->cands = (cand*)xrealloc(sc->cands, sc->cap_cands * sizeof(cand));
  }
  uint64_t n_cands = 0;
  for (uint64_t k = 0; k < sc->hcount; k++) {
    const uint32_t h = sc->hused[k];
    sc->cands[n_cands].sid = sc->hkey[h] - 1; sc->cands[n_cands].longest = sc->hval[h]; n_cands++;
  }
  cand_sort(sc->cands, n_cands);

  /* PatternCoverage ctor — src/pattern_coverage.cc:8-13 */
  uint64_t n_cov = 0;
  for (uint64_t j = 0; j < p_length; j++) {
    uint64_t k = 0;
    for (; k < n_cov; k++) if (sc->cov_word[k] == pattern_wids[j]) break;
    if (k == n_cov) { sc->cov_word[n_cov] = pattern_wids[j]; sc->cov_count[n_cov] = 0; n_cov++; }
    sc->cov_count[k]++;
  }
  /* pattern real tokens and itoks — src/fuzzy_match.cc:558-562 */
  for (uint64_t j = 0; j <= p_length; j++) { sc->st2[j] = NULL; sc->sn2[j] = 0; }
  get_itoks(real, sc->st2, sc->sn2, (int)p_length + 1);
  const int compare_real = ix->has_real;
  int n_real2 = 0;
  if (compare_real && real && real->tokstring)
    n_real2 = split_tokstring(real->tokstring, &sc->real2, &sc->cap_real2);

  /* F. candidate loop — src/fuzzy_match.cc:564-612 */
  sc->heap_n = 0;
  heap_push(sc, FLT_MAX);
  sc->n_result = 0;
  for (uint64_t c = 0; c < n_cands; c++) {
    const uint32_t s_id = sc->cands[c].sid;
    const uint64_t longest_match = sc->cands[c].longest;
    uint64_t s_length = 0;
    const uint32_t* sentence_wids = get_sentence(ix, s_id, &s_length);
    if (ctr) { ctr->candidates++; ctr->cand_bytes += 4 + 4 * (s_length + 1); }
    uint64_t num_covered_words;
    if (longest_match < p_length) {
      /* PatternCoverage::count_covered_words — src/pattern_coverage.cc:15-28 */
      num_covered_words = 0;
      for (uint64_t k = 0; k < n_cov; k++) {
        uint64_t t = 0;
        for (; t < s_length; t++) if (sentence_wids[t] == sc->cov_word[k]) break;
        if (ctr) ctr->coverage_tokens += (t < s_length) ? t + 1 : s_length;
        if (t < s_length) num_covered_words += sc->cov_count[k];
      }
    } else
      num_covered_words = p_length;
    if (theoretical_rejection_cover(p_length, s_length, num_covered_words, ins, del, rep, fuzzy))
      continue;
    const float diff_word = 100.f / fmo_normalizer(p_length, s_length, ins, del, rep); /* costs.hh:54-56 */
    const fmo_real* sentence_real = (ix->has_real && ix->real) ? &ix->real[s_id] : NULL;
    const float cost_upper_bound = sc->heap[0];
    float cost = edit_distance_full(sentence_wids, sentence_real, (int)s_length, pattern_wids,
                                    sc->real2, n_real2, (int)p_length, sc->st2, sc->sn2, sc->idf,
                                    diff_word * vocab_idf_penalty / idf_max, ins, del, rep, diff_word,
                                    cost_upper_bound, compare_real, &sc->dp, ctr);
    if ((prm->no_perfect && cost == 0 && (s_length == p_length)) || cost > cost_upper_bound)
      continue;
    float score = fmo_score_from_cost(cost);
    heap_push(sc, cost);
    if (score < fuzzy || (contrast_buffer > 0 && sc->heap_n > (size_t)contrast_buffer))
      heap_pop(sc);
    if (score >= fuzzy) {
      if (sc->n_result + 1 > sc->cap_result) {
        sc->cap_result = sc->cap_result ? sc->cap_result * 2 : 64;
        sc->result = (acc_match*)xrealloc(sc->result, sc->cap_result * sizeof(acc_match));
      }
      acc_match* m = &sc->result[sc->n_result++];
      m->score = score;
      m->penalty = 0; /* uninitialised in the reference (fuzzy_match.hh:34-40); observed 0 */
      m->max_subseq = (int32_t)longest_match;
      m->s_id = s_id;
      m->length = (int)s_length;
      m->s = sentence_wids;
    }
  }
  qsort(sc->result, sc->n_result, sizeof(acc_match), result_comp);

  /* G. output — src/fuzzy_match.cc:614-679 */
  uint64_t n_out = 0;
  if (prm->contrast > 0) {
    const uint64_t n = sc->n_result;
    const float factor = prm->contrast;
    if (n * 4 > sc->cap_memo) {
      sc->cap_memo = n * 4 + 64;
      sc->pen_sum = (float*)xrealloc(sc->pen_sum, sc->cap_memo * sizeof(float));
      sc->pen_max = (float*)xrealloc(sc->pen_max, sc->cap_memo * sizeof(float));
      sc->pen_cnt = (uint32_t*)xrealloc(sc->pen_cnt, sc->cap_memo * sizeof(uint32_t));
    }
    /* the list `candidates` keeps (score desc, s_id asc) order; erased entries are flagged.
     * Every round walks `matches` in order for every remaining candidate (src/fuzzy_match.cc:631-650): the entries the vector
     * held on entry first (sc->n_prior of them), then the picks.  The pair's edit cost is looked up in a memo keyed on the two
     * S_IDS, not on the sequences (:627, :637): a memory entry whose s_id an earlier memory entry already had — two prior
     * entries from another translation memory, or a pick whose id a prior entry carries — gets the cost that was computed for
     * the FIRST of them, whatever its own tokens are.  A cost depends on the candidate's and the memory entry's tokens only, so
     * the memo is restated on the memory side: `mem_*` lists the memory entries in `matches` order with their ids, and an entry
     * whose id appeared before is evaluated on the tokens of the first entry with that id — what every candidate's memo holds.
     * Without ids for the prior entries (prior_sid == NULL) each counts as an id of its own.
     * Every (candidate, memory position) contributes once, in `matches` order, so the running sums equal std::accumulate over
     * the penalties vector.  number_of_matches here counts NEW matches: the caller passes the reference's number_of_matches
     * minus the entries already in its vector (loop condition fuzzy_match.cc:629). */
    uint8_t* taken = (uint8_t*)calloc(n ? n : 1, 1);
    const uint64_t mem_cap = (uint64_t)sc->n_prior + n + 1;
    uint64_t*        mem_key = (uint64_t*)xrealloc(NULL, mem_cap * sizeof(uint64_t));
    const uint32_t** mem_s = (const uint32_t**)xrealloc(NULL, mem_cap * sizeof(uint32_t*));
    int*             mem_len = (int*)xrealloc(NULL, mem_cap * sizeof(int));
    uint64_t n_mem = 0;
    for (uint64_t i = 0; i < n; i++) { sc->pen_sum[i] = 0.f; sc->pen_max[i] = 0.f; sc->pen_cnt[i] = 0; }
    /* appends a memory entry; returns the entry whose tokens the memo makes every candidate see for it */
#define FMO_ADD_MEMORY(key, ms, mlen, eff)                                                     \
    do {                                                                                         \
      (eff) = n_mem;                                                                             \
      for (uint64_t k_ = 0; k_ < n_mem; k_++) if (mem_key[k_] == (key)) { (eff) = k_; break; }   \
      mem_key[n_mem] = (key); mem_s[n_mem] = (ms); mem_len[n_mem] = (mlen); n_mem++;             \
    } while (0)
#define FMO_FOLD_MEMORY(i, e)                                                                                            \
    do {                                                                                                                 \
      acc_match* m_ = &sc->result[i];                                                                                    \
      const float dw_ = 100.f / fmo_normalizer((uint64_t)m_->length, (uint64_t)mem_len[e], 1.f, 1.f, 1.f);              \
      const float cost_ = fmo_edit_distance_ids(m_->s, m_->length, mem_s[e], mem_len[e], 1.f, 1.f, 1.f, dw_, FLT_MAX);   \
      if (ctr) { ctr->dp_calls++; ctr->dp_cells += (uint64_t)m_->length * (uint64_t)mem_len[e]; }                        \
      const float pen_ = fmo_score_from_cost(cost_);                                                                     \
      sc->pen_sum[i] = sc->pen_sum[i] + pen_; /* std::accumulate(…, 0.f) in `matches` order */                           \
      sc->pen_max[i] = sc->pen_cnt[i] ? (sc->pen_max[i] < pen_ ? pen_ : sc->pen_max[i]) : pen_;                          \
      sc->pen_cnt[i]++;                                                                                                  \
      m_->penalty = prm->reduce == 1 ? sc->pen_max[i] : sc->pen_sum[i] / (float)sc->pen_cnt[i];                          \
    } while (0)
    for (uint32_t r = 0; r < sc->n_prior; r++) {
      uint64_t e;
      FMO_ADD_MEMORY(sc->prior_sid ? (uint64_t)sc->prior_sid[r] : ((1ull << 32) + r), sc->prior_tok + sc->prior_off[r],
                     (int)(sc->prior_off[r + 1] - sc->prior_off[r]), e);
      for (uint64_t i = 0; i < n; i++) FMO_FOLD_MEMORY(i, e);
    }
    uint64_t remaining = n;
    int64_t  last_pick = -1;
    while (remaining && (number_of_matches == 0 || n_out < number_of_matches)) {
      if (last_pick >= 0) {
        const acc_match* mem = &sc->result[last_pick];
        uint64_t e;
        FMO_ADD_MEMORY((uint64_t)mem->s_id, mem->s, mem->length, e);
        for (uint64_t i = 0; i < n; i++) {
          if (taken[i]) continue;
          FMO_FOLD_MEMORY(i, e);
        }
      }
      /* std::max_element: first maximum in list order */
      int64_t best = -1;
      float   best_key = 0;
      for (uint64_t i = 0; i < n; i++) {
        if (taken[i]) continue;
        float key = sc->result[i].score - factor * sc->result[i].penalty;
        if (best < 0 || best_key < key) { best = (int64_t)i; best_key = key; }
      }
      if (n_out < cap) {
        out[n_out].score = sc->result[best].score;
        out[n_out].penalty = sc->result[best].penalty;
        out[n_out].max_subseq = sc->result[best].max_subseq;
        out[n_out].s_id = sc->result[best].s_id;
      }
      n_out++;
      taken[best] = 1;
      remaining--;
      last_pick = best;
    }
#undef FMO_FOLD_MEMORY
#undef FMO_ADD_MEMORY
    free(taken); free(mem_key); free((void*)mem_s); free(mem_len);
  } else {
    for (uint64_t i = 0; i < sc->n_result && (number_of_matches == 0 || n_out < number_of_matches); i++) {
      if (n_out < cap) {
        out[n_out].score = sc->result[i].score;
        out[n_out].penalty = 0;
        out[n_out].max_subseq = sc->result[i].max_subseq;
        out[n_out].s_id = sc->result[i].s_id;
      }
      n_out++;
    }
  }
  if (ctr && n_out) ctr->queries_with_match++;
  return (int64_t)n_out;
}

int64_t fmo_match(const fmo_index* ix, const fmo_params* prm, const uint32_t* p_wids, uint32_t p,
                  const char* p_real_tokstring, uint32_t n_itoks, const uint64_t* itok_idx,
                  const char* const* itok_str, fmo_rec* out, uint64_t cap, fmo_counters* ctr) {
  if (!ix || !prm || !ix->sorted) return -1;
  fmo_scratch sc;
  memset(&sc, 0, sizeof(sc));
  fmo_real real;
  memset(&real, 0, sizeof(real));
  const fmo_real* rp = NULL;
  if (p_real_tokstring || n_itoks) {
    real_set(&real, p_real_tokstring, n_itoks, itok_idx, itok_str);
    rp = &real;
  }
  int64_t r = match_impl(ix, prm, p_wids, p, rp, out, cap, ctr, &sc);
  real_clear(&real);
  scratch_free(&sc);
  return r;
}

/* match() when `matches` already holds n_prior entries (their token sequences as CSR): only the contrastive rerank
 * looks at them (src/fuzzy_match.cc:622-650).  prm->nmatch = NEW matches wanted. */
int64_t fmo_match_prior(const fmo_index* ix, const fmo_params* prm, const uint32_t* p_wids, uint32_t p,
                        const uint32_t* prior_off, const uint32_t* prior_tok, uint32_t n_prior, fmo_rec* out, uint64_t cap) {
  return fmo_match_prior_ids(ix, prm, p_wids, p, prior_off, prior_tok, NULL, n_prior, out, cap);
}

/* ... with the prior entries' Match::s_id: the memo of the rerank is keyed on ids (src/fuzzy_match.cc:637) */
int64_t fmo_match_prior_ids(const fmo_index* ix, const fmo_params* prm, const uint32_t* p_wids, uint32_t p,
                            const uint32_t* prior_off, const uint32_t* prior_tok, const uint32_t* prior_sid, uint32_t n_prior,
                            fmo_rec* out, uint64_t cap) {
  if (!ix || !prm || !ix->sorted) return -1;
  fmo_scratch sc;
  memset(&sc, 0, sizeof(sc));
  sc.n_prior = n_prior; sc.prior_off = prior_off; sc.prior_tok = prior_tok; sc.prior_sid = prior_sid;
  int64_t r = match_impl(ix, prm, p_wids, p, NULL, out, cap, NULL, &sc);
  scratch_free(&sc);
  return r;
}

/* ------------------------------------------------------------------------------------------ */
/* FuzzyMatch::subsequence — src/fuzzy_match.cc:250-365, after _tokenize_and_normalize            */
/* ------------------------------------------------------------------------------------------ */

typedef struct subseq { float weight; uint64_t position; uint64_t length; } subseq;

/* Subseq::operator< — src/fuzzy_match.cc:240-243 */
static inline int subseq_less(const subseq* a, const subseq* b) {
  return a->weight < b->weight || (a->weight == b->weight && a->position > b->position);
}

/* std::priority_queue<Subseq> is std::push_heap / std::pop_heap over a vector.  Equal keys (same weight and
 * position, different length: only possible when an idf weight is exactly 0) pop in an order that depends on the
 * heap algorithm, so libstdc++'s published one is restated: __push_heap and __adjust_heap (bits/stl_heap.h). */
static void sq_push_heap(subseq* first, int64_t hole, int64_t top, subseq value) {
  int64_t parent = (hole - 1) / 2;
  while (hole > top && subseq_less(&first[parent], &value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}
static void sq_adjust_heap(subseq* first, int64_t hole, int64_t len, subseq value) {
  const int64_t top = hole;
  int64_t second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (subseq_less(&first[second], &first[second - 1])) second--;
    first[hole] = first[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    first[hole] = first[second - 1];
    hole = second - 1;
  }
  sq_push_heap(first, hole, top, value);
}

int fmo_subsequence(const fmo_index* ix, const uint32_t* pidx, uint32_t p, const char* p_real_tokstring,
                    uint32_t n_itoks, const uint64_t* itok_idx, const char* const* itok_str,
                    uint32_t number_of_matches, int no_perfect, int min_subseq_length, float min_subseq_ratio,
                    int idf_weighting, fmo_rec* out, uint32_t* out_position) {
  if (!ix || !ix->sorted || !out) return -1;
  const uint64_t p_length = p;
  /* :267-271 */
  if ((int)(min_subseq_ratio * p_length) > min_subseq_length) min_subseq_length = (int)(min_subseq_ratio * p_length);
  if ((int)p_length < min_subseq_length) return 0;
  fmo_real real;
  memset(&real, 0, sizeof(real));
  const fmo_real* rp = NULL;
  if (p_real_tokstring || n_itoks) { real_set(&real, p_real_tokstring, n_itoks, itok_idx, itok_str); rp = &real; }
  /* compute_idf_penalty(pidx, -1) — :276-277,372-390 */
  float* idf = (float*)xrealloc(NULL, (p_length + 1) * sizeof(float));
  const unsigned num_sentences = (unsigned)ix->n_sent;
  for (uint64_t j = 0; j < p_length; j++) {
    const uint32_t wid = pidx[j];
    if (wid != 1) {
      const uint32_t f = (ix->sfreq && wid < ix->vocab_size) ? ix->sfreq[wid] : 0;
      idf[j] = logf((float)num_sentences / (float)f);
    } else
      idf[j] = -1.f;
  }
  /* subsequences by weight — :279-290 */
  subseq* q = NULL;
  int64_t qn = 0, qcap = 0;
  for (uint64_t it = 0; it < p_length; it++) {
    float idf_weight = 0;
    for (uint64_t jt = it; jt < p_length; jt++) {
      const float weight = idf[jt];
      if (weight == -1) break;
      idf_weight += idf_weighting ? weight : 1;
      if ((int)(jt - it + 1) >= min_subseq_length) {
        if (qn + 1 > qcap) { qcap = qcap ? qcap * 2 : 256; q = (subseq*)xrealloc(q, (size_t)qcap * sizeof(subseq)); }
        subseq v; v.weight = idf_weight; v.position = it; v.length = jt - it + 1;
        q[qn] = v; qn++;
        sq_push_heap(q, qn - 1, 0, v);
      }
    }
  }
  int max_distance = 10000;
  fmo_rec best;
  memset(&best, 0, sizeof(best));
  uint32_t best_pos = 0;
  /* std::set<unsigned> candidates, perfect — :295-296 */
  uint32_t *cands = NULL, *perfect = NULL;
  uint64_t n_cands = 0, n_perfect = 0, cap_c = 0, cap_p = 0;
  /* pattern side of the DP — :298-301 */
  const char** st = (const char**)xrealloc(NULL, (p_length + 1) * sizeof(char*));
  int* sn = (int*)xrealloc(NULL, (p_length + 1) * sizeof(int));
  for (uint64_t j = 0; j <= p_length; j++) { st[j] = NULL; sn[j] = 0; }
  get_itoks(rp, st, sn, (int)p_length + 1);
  const int compare_real = ix->has_real;
  strview* real2 = NULL;
  int cap_real2 = 0, n_real2 = 0;
  if (compare_real && rp && rp->tokstring) n_real2 = split_tokstring(rp->tokstring, &real2, &cap_real2);
  dp_scratch dp;
  memset(&dp, 0, sizeof(dp));
  while (qn > 0 && max_distance == 10000) {  /* :303-304 */
    const subseq top = q[0];
    uint64_t range[2];
    fmo_equal_range(ix, pidx + top.position, top.length, 0, 0, range, NULL);
    for (uint64_t si = range[0]; si < range[1] && n_cands < number_of_matches; si++) {  /* :311-312 */
      const uint32_t s_id = ix->suf_sid[si];
      int seen = 0;
      for (uint64_t k = 0; k < n_cands && !seen; k++) seen = cands[k] == s_id;
      for (uint64_t k = 0; k < n_perfect && !seen; k++) seen = perfect[k] == s_id;
      if (seen) continue;
      uint64_t s_length = 0;
      const uint32_t* thes = get_sentence(ix, s_id, &s_length);
      const float diff_word = 100.f / fmo_normalizer(p_length, s_length, 1.f, 1.f, 1.f);
      const fmo_real* sentence_real = (ix->has_real && ix->real) ? &ix->real[s_id] : NULL;
      const float cost = edit_distance_full(thes, sentence_real, (int)s_length, pidx, real2, n_real2, (int)p_length, st, sn,
                                            idf, 0.f, 1.f, 1.f, 1.f, diff_word, (float)max_distance, compare_real, &dp, NULL);
      if (cost == 0 && no_perfect) {  /* :329-332 */
        if (n_perfect + 1 > cap_p) { cap_p = cap_p ? cap_p * 2 : 64; perfect = (uint32_t*)xrealloc(perfect, cap_p * 4); }
        perfect[n_perfect++] = s_id;
        continue;
      }
      if (cost < max_distance) {  /* :333-351; max_distance is an int: the assignment truncates */
        best.score = fmo_score_from_cost(cost);
        best.max_subseq = (int32_t)top.length;
        best.s_id = s_id;
        best_pos = (uint32_t)top.position;
        max_distance = (int)cost;
        if (cost == 0) break;
      }
      if (n_cands + 1 > cap_c) { cap_c = cap_c ? cap_c * 2 : 64; cands = (uint32_t*)xrealloc(cands, cap_c * 4); }
      cands[n_cands++] = s_id;
    }
    /* subseq_queue.pop() — :356 */
    const subseq last = q[qn - 1];
    qn--;
    if (qn > 0) sq_adjust_heap(q, 0, qn, last);
  }
  free(q); free(idf); free(cands); free(perfect); free((void*)st); free(sn); free(real2);
  free(dp.arr); free(dp.tag); free((void*)dp.st1); free(dp.sn1); free(dp.real1);
  real_clear(&real);
  if (max_distance != 10000) {
    *out = best;
    if (out_position) *out_position = best_pos;
    return 1;
  }
  return 0;
}

typedef struct batch_job {
  const fmo_index*  ix;
  const fmo_params* prm;
  const uint32_t*   q_off;
  const uint32_t*   q_tok;
  uint64_t          nq;
  fmo_rec*        out;
  uint32_t*         out_count;
  uint64_t          cap;
  atomic_ullong*    next;
  int               want_ctr;
  fmo_counters      ctr;
  const char*       real_packed;  /* optional: pattern q's tab-joined real tokens at real_packed + real_off[q] */
  const uint64_t*   real_off;
} batch_job;

static void* batch_worker(void* arg) {
  batch_job*  j = (batch_job*)arg;
  fmo_scratch sc;
  memset(&sc, 0, sizeof(sc));
  const uint64_t chunk = 64;
  for (;;) {
    uint64_t b = atomic_fetch_add(j->next, chunk);
    if (b >= j->nq) break;
    uint64_t e = b + chunk < j->nq ? b + chunk : j->nq;
    for (uint64_t q = b; q < e; q++) {
      fmo_real real;
      memset(&real, 0, sizeof(real));
      if (j->real_packed) real.tokstring = (char*)(j->real_packed + j->real_off[q]);  /* borrowed, never freed */
      int64_t n = match_impl(j->ix, j->prm, j->q_tok + j->q_off[q], j->q_off[q + 1] - j->q_off[q], j->real_packed ? &real : NULL,
                             j->out + q * j->cap, j->cap, j->want_ctr ? &j->ctr : NULL, &sc);
      j->out_count[q] = (uint32_t)((uint64_t)n < j->cap ? (uint64_t)n : j->cap);
    }
  }
  scratch_free(&sc);
  return NULL;
}

int fmo_match_batch(const fmo_index* ix, const fmo_params* prm, const uint32_t* q_off,
                    const uint32_t* q_tok, uint64_t nq, int nthreads, fmo_rec* out,
                    uint32_t* out_count, uint64_t cap, fmo_counters* ctr) {
  return fmo_match_batch_real(ix, prm, q_off, q_tok, nq, NULL, NULL, nthreads, out, out_count, cap, ctr);
}

/* the same batch with every pattern's `real` Sentence (match(const Sentence&, const Tokens&, ...), src/fuzzy_match.cc:435) */
int fmo_match_batch_real(const fmo_index* ix, const fmo_params* prm, const uint32_t* q_off,
                         const uint32_t* q_tok, uint64_t nq, const char* real_packed, const uint64_t* real_off,
                         int nthreads, fmo_rec* out, uint32_t* out_count, uint64_t cap, fmo_counters* ctr) {
  if (!ix || !prm || !ix->sorted) return -1;
  if ((real_packed == NULL) != (real_off == NULL)) return -1;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 512) nthreads = 512;
  atomic_ullong next;
  atomic_init(&next, 0);
  batch_job* jobs = (batch_job*)calloc((size_t)nthreads, sizeof(batch_job));
  pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
  for (int t = 0; t < nthreads; t++) {
    jobs[t].ix = ix; jobs[t].prm = prm; jobs[t].q_off = q_off; jobs[t].q_tok = q_tok; jobs[t].nq = nq;
    jobs[t].out = out; jobs[t].out_count = out_count; jobs[t].cap = cap; jobs[t].next = &next;
    jobs[t].want_ctr = ctr != NULL;
    jobs[t].real_packed = real_packed; jobs[t].real_off = real_off;
  }
  for (int t = 1; t < nthreads; t++) pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
  batch_worker(&jobs[0]);
  for (int t = 1; t < nthreads; t++) pthread_join(th[t], NULL);
  if (ctr) {
    memset(ctr, 0, sizeof(*ctr));
    for (int t = 0; t < nthreads; t++) {
      const uint64_t* src = (const uint64_t*)&jobs[t].ctr;
      uint64_t*       dst = (uint64_t*)ctr;
      for (size_t k = 0; k < sizeof(fmo_counters) / sizeof(uint64_t); k++) dst[k] += src[k];
    }
  }
  free(jobs); free(th);
  return 0;
}
