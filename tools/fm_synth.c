/*
 * fm_synth.c — portable synthetic corpus / query generator for the benchmark configs
 * (SURVEY.md §8d "Synthetic inputs").  Measurement infrastructure, not part of the product path:
 * it only produces inputs (vocabulary-id token streams) for bench.py and the tests.
 *
 * Everything is derived from splitmix64/xoshiro256** and explicit tables (no libc rand, no
 * std::*_distribution), so this container and the GPU box generate identical bytes.
 *
 *   vocabulary   V types, Zipf(s = 1.0) by inverse-CDF table; type k has vocabulary id k + 2
 *                (0 = sentence separator, 1 = unknown, like the reference's VocabIndexer)
 *   length       3 + Poisson(mean - 3), clipped to [3, 100]
 *   structure    near-duplicate families: a base sentence plus K-1 variants,
 *                K = 1 + Geometric(continue 0.875) capped at 64 (mean 8); each variant = base with
 *                i.i.d. per-token edits at 10 % (1/3 delete, 1/3 substitute, 1/3 insert-after);
 *                ~20 % of base tokens come as 3..6-token phrases from a Zipf pool of 10 000 phrases;
 *                sentence order shuffled
 *   queries      80 % = a random corpus sentence edited at rate e ~ U(0, 0.3); 20 % fresh sentences
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct fms_set {
  uint64_t  n_sent;
  uint64_t  n_tok;
  uint32_t* off; /* n_sent + 1 */
  uint32_t* tok;
  uint32_t  vocab_size; /* V + 2 */
} fms_set;

typedef struct rng { uint64_t s[4]; } rng;

static uint64_t splitmix64(uint64_t* x) {
  uint64_t z = (*x += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static void rng_seed(rng* r, uint64_t seed) { for (int i = 0; i < 4; i++) r->s[i] = splitmix64(&seed); }
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static inline uint64_t rng_next(rng* r) { /* xoshiro256** */
  const uint64_t result = rotl(r->s[1] * 5, 7) * 9;
  const uint64_t t = r->s[1] << 17;
  r->s[2] ^= r->s[0]; r->s[3] ^= r->s[1]; r->s[1] ^= r->s[2]; r->s[0] ^= r->s[3];
  r->s[2] ^= t; r->s[3] = rotl(r->s[3], 45);
  return result;
}
static inline double rng_u01(rng* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t rng_below(rng* r, uint64_t n) { return (uint64_t)(rng_u01(r) * (double)n); }

typedef struct tables {
  uint32_t V;
  double*  zipf_cdf;   /* V */
  double   pois_cdf[128];
  int      pois_n;
  uint32_t n_phr;
  uint32_t* phr_off;   /* n_phr + 1 */
  uint32_t* phr_tok;
  double*  phr_cdf;
} tables;

static uint32_t sample_cdf(const double* cdf, uint32_t n, double u) {
  uint32_t lo = 0, hi = n - 1;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (cdf[mid] > u) hi = mid; else lo = mid + 1;
  }
  return lo;
}

static void build_zipf(double* cdf, uint32_t n) {
  double h = 0;
  for (uint32_t k = 0; k < n; k++) { h += 1.0 / (double)(k + 1); cdf[k] = h; }
  for (uint32_t k = 0; k < n; k++) cdf[k] /= h;
  cdf[n - 1] = 1.0;
}

static void tables_init(tables* t, uint32_t V, double mean_len, rng* r) {
  t->V = V;
  t->zipf_cdf = (double*)malloc(sizeof(double) * V);
  build_zipf(t->zipf_cdf, V);
  /* Poisson(lambda) by weight recurrence w_k = w_{k-1} * lambda / k (no exp needed), k in [0, 97] */
  double lambda = mean_len - 3.0;
  if (lambda < 0.01) lambda = 0.01;
  double w = 1.0, sum = 0.0;
  t->pois_n = 98;
  for (int k = 0; k < t->pois_n; k++) {
    if (k) w = w * lambda / (double)k;
    sum += w;
    t->pois_cdf[k] = sum;
  }
  for (int k = 0; k < t->pois_n; k++) t->pois_cdf[k] /= sum;
  t->pois_cdf[t->pois_n - 1] = 1.0;
  /* phrase pool */
  t->n_phr = 10000;
  t->phr_off = (uint32_t*)malloc(sizeof(uint32_t) * (t->n_phr + 1));
  t->phr_tok = (uint32_t*)malloc(sizeof(uint32_t) * t->n_phr * 6);
  t->phr_cdf = (double*)malloc(sizeof(double) * t->n_phr);
  build_zipf(t->phr_cdf, t->n_phr);
  uint32_t o = 0;
  for (uint32_t i = 0; i < t->n_phr; i++) {
    t->phr_off[i] = o;
    uint32_t len = 3 + (uint32_t)rng_below(r, 4);
    for (uint32_t k = 0; k < len; k++) t->phr_tok[o++] = 2 + sample_cdf(t->zipf_cdf, V, rng_u01(r));
  }
  t->phr_off[t->n_phr] = o;
}

static void tables_free(tables* t) {
  free(t->zipf_cdf); free(t->phr_off); free(t->phr_tok); free(t->phr_cdf);
}

static uint32_t sample_len(const tables* t, rng* r) {
  uint32_t len = 3 + sample_cdf(t->pois_cdf, (uint32_t)t->pois_n, rng_u01(r));
  return len > 100 ? 100 : len;
}

/* base sentence into out[0..len) */
static uint32_t gen_base(const tables* t, rng* r, uint32_t* out) {
  const uint32_t len = sample_len(t, r);
  uint32_t n = 0;
  while (n < len) {
    if (rng_u01(r) < (1.0 / 19.0)) { /* phrase start: ~20 % of tokens end up inside phrases */
      uint32_t ph = sample_cdf(t->phr_cdf, t->n_phr, rng_u01(r));
      for (uint32_t k = t->phr_off[ph]; k < t->phr_off[ph + 1] && n < len; k++) out[n++] = t->phr_tok[k];
    } else
      out[n++] = 2 + sample_cdf(t->zipf_cdf, t->V, rng_u01(r));
  }
  return n;
}

/* per-token edits at total rate e; returns new length (>= 1, <= 200) */
static uint32_t edit_sentence(const tables* t, rng* r, const uint32_t* in, uint32_t n, double e, uint32_t* out) {
  uint32_t m = 0;
  for (uint32_t i = 0; i < n; i++) {
    if (rng_u01(r) < e) {
      uint32_t kind = (uint32_t)rng_below(r, 3);
      if (kind == 0) continue;                                                       /* delete */
      if (kind == 1) { out[m++] = 2 + sample_cdf(t->zipf_cdf, t->V, rng_u01(r)); }   /* substitute */
      else { out[m++] = in[i]; out[m++] = 2 + sample_cdf(t->zipf_cdf, t->V, rng_u01(r)); } /* insert after */
    } else
      out[m++] = in[i];
  }
  if (m == 0) out[m++] = in[0];
  return m;
}

static void set_reserve(fms_set* s, uint64_t* cap_tok, uint64_t need) {
  if (need <= *cap_tok) return;
  while (*cap_tok < need) *cap_tok = *cap_tok ? *cap_tok * 2 : (1u << 20);
  s->tok = (uint32_t*)realloc(s->tok, sizeof(uint32_t) * *cap_tok);
}

fms_set* fms_corpus_new(uint64_t n_sent, double mean_len, uint32_t V, uint64_t seed) {
  rng r;
  rng_seed(&r, seed);
  tables t;
  tables_init(&t, V, mean_len, &r);
  fms_set* s = (fms_set*)calloc(1, sizeof(fms_set));
  s->vocab_size = V + 2;
  s->off = (uint32_t*)malloc(sizeof(uint32_t) * (n_sent + 1));
  uint64_t cap_tok = 0;
  uint32_t base[128], var[256];
  uint64_t ns = 0;
  s->off[0] = 0;
  while (ns < n_sent) {
    const uint32_t bl = gen_base(&t, &r, base);
    uint32_t K = 1;
    while (K < 64 && rng_u01(&r) < 0.875) K++;
    for (uint32_t k = 0; k < K && ns < n_sent; k++) {
      const uint32_t* src = base;
      uint32_t n = bl;
      if (k > 0) { n = edit_sentence(&t, &r, base, bl, 0.10, var); src = var; }
      set_reserve(s, &cap_tok, s->n_tok + n);
      memcpy(s->tok + s->n_tok, src, sizeof(uint32_t) * n);
      s->n_tok += n;
      s->off[++ns] = (uint32_t)s->n_tok;
    }
  }
  /* shuffle sentence order (Fisher-Yates on a permutation, then gather) */
  uint32_t* perm = (uint32_t*)malloc(sizeof(uint32_t) * n_sent);
  for (uint64_t i = 0; i < n_sent; i++) perm[i] = (uint32_t)i;
  for (uint64_t i = n_sent; i > 1; i--) {
    uint64_t j = rng_below(&r, i);
    uint32_t tmp = perm[i - 1]; perm[i - 1] = perm[j]; perm[j] = tmp;
  }
  uint32_t* tok2 = (uint32_t*)malloc(sizeof(uint32_t) * (s->n_tok ? s->n_tok : 1));
  uint32_t* off2 = (uint32_t*)malloc(sizeof(uint32_t) * (n_sent + 1));
  uint64_t o = 0;
  off2[0] = 0;
  for (uint64_t i = 0; i < n_sent; i++) {
    const uint32_t a = s->off[perm[i]], b = s->off[perm[i] + 1];
    memcpy(tok2 + o, s->tok + a, sizeof(uint32_t) * (b - a));
    o += b - a;
    off2[i + 1] = (uint32_t)o;
  }
  free(s->tok); free(s->off); free(perm);
  s->tok = tok2; s->off = off2; s->n_sent = n_sent;
  tables_free(&t);
  return s;
}

fms_set* fms_queries_new(const fms_set* corpus, uint64_t nq, double mean_len, uint32_t V, uint64_t seed) {
  rng r;
  rng_seed(&r, seed);
  tables t;
  tables_init(&t, V, mean_len, &r);
  fms_set* s = (fms_set*)calloc(1, sizeof(fms_set));
  s->vocab_size = V + 2;
  s->off = (uint32_t*)malloc(sizeof(uint32_t) * (nq + 1));
  s->off[0] = 0;
  uint64_t cap_tok = 0;
  uint32_t base[128], var[512];
  for (uint64_t q = 0; q < nq; q++) {
    uint32_t n;
    const uint32_t* src;
    if (rng_u01(&r) < 0.8 && corpus->n_sent) {
      const uint64_t sid = rng_below(&r, corpus->n_sent);
      const double e = rng_u01(&r) * 0.3;
      const uint32_t a = corpus->off[sid], b = corpus->off[sid + 1];
      n = edit_sentence(&t, &r, corpus->tok + a, b - a, e, var);
      src = var;
    } else {
      n = gen_base(&t, &r, base);
      src = base;
    }
    set_reserve(s, &cap_tok, s->n_tok + n);
    memcpy(s->tok + s->n_tok, src, sizeof(uint32_t) * n);
    s->n_tok += n;
    s->off[q + 1] = (uint32_t)s->n_tok;
  }
  s->n_sent = nq;
  tables_free(&t);
  return s;
}

void fms_free(fms_set* s) {
  if (!s) return;
  free(s->off); free(s->tok); free(s);
}
uint64_t        fms_num_sentences(const fms_set* s) { return s->n_sent; }
uint64_t        fms_num_tokens(const fms_set* s) { return s->n_tok; }
uint32_t        fms_vocab_size(const fms_set* s) { return s->vocab_size; }
const uint32_t* fms_offsets(const fms_set* s) { return s->off; }
const uint32_t* fms_tokens(const fms_set* s) { return s->tok; }

/* spelling of vocabulary type k: bijective base-26, lower case, at least two letters */
int fms_word(uint32_t k, char* buf, int buflen) {
  char tmp[16];
  int n = 0;
  uint64_t v = (uint64_t)k + 26; /* skip the 26 one-letter words */
  v += 1;
  while (v > 0 && n < 15) { v--; tmp[n++] = (char)('a' + (v % 26)); v /= 26; }
  if (n + 1 > buflen) return -1;
  for (int i = 0; i < n; i++) buf[i] = tmp[n - 1 - i];
  buf[n] = '\0';
  return n;
}

/* A batch of patterns as a tokenizer would hand it over: the spellings of ids[0..n) back to back, each NUL-terminated (ids 0 and 1
 * — the empty word and the unknown word — are written as an empty spelling and as "?" — a word no vocabulary of this generator
 * holds).  Returns the bytes written, or the bytes needed when out == NULL / cap is too small (nothing is written past cap). */
uint64_t fms_pack_text(const uint32_t* ids, uint64_t n, char* out, uint64_t cap) {
  uint64_t at = 0;
  char w[16];
  for (uint64_t i = 0; i < n; i++) {
    int len;
    if (ids[i] == 0) { len = 0; w[0] = '\0'; }
    else if (ids[i] == 1) { len = 1; w[0] = '?'; w[1] = '\0'; }
    else len = fms_word(ids[i] - 2, w, (int)sizeof w);
    if (len < 0) len = 0;
    if (out && at + (uint64_t)len + 1 <= cap) memcpy(out + at, w, (size_t)len + 1);
    at += (uint64_t)len + 1;
  }
  return at;
}

/* ---- the same sentences as TEXT the reference CLI would index under its default -p nbr,tag,cas (cli/src/FuzzyMatch-cli.cc:328):
 * the first word of a sentence is capitalised, vocabulary types with k % 40 == 19 (~2 % of the Zipf mass) are NUMBERS written as the
 * decimal digits of k.  Under nbr + cas (src/fuzzy_match.cc:123-192) a word's normalised token is its lower-case spelling and its
 * real token the case feature ("C" capitalised, "L" lower case), a number's normalised token is the placeholder ｟num｠ and its
 * real token the digits: every sentence's real tokens differ from its normalised tokens, which is what such an index costs. ---- */
int fms_is_number(uint32_t k) { return k % 40u == 19u; }

static int spell_cased(uint32_t id, int first, char* w /* >= 16 */) {
  if (id == 0) { w[0] = '\0'; return 0; }
  if (id == 1) { w[0] = '?'; w[1] = '\0'; return 1; }
  const uint32_t k = id - 2;
  if (fms_is_number(k)) {
    char tmp[12];
    int n = 0;
    uint32_t v = k;
    do { tmp[n++] = (char)('0' + v % 10u); v /= 10u; } while (v);
    for (int i = 0; i < n; i++) w[i] = tmp[n - 1 - i];
    w[n] = '\0';
    return n;
  }
  const int n = fms_word(k, w, 16);
  if (n > 0 && first) w[0] = (char)(w[0] - 'a' + 'A');
  return n;
}

/* One '\n'-terminated line of blank-separated tokens per sentence.  Returns the bytes written, or the bytes needed when
 * out == NULL / cap is too small (nothing is written past cap). */
uint64_t fms_pack_lines_cased(const uint32_t* off, const uint32_t* ids, uint64_t n_sent, char* out, uint64_t cap) {
  uint64_t at = 0;
  char w[16];
  for (uint64_t s = 0; s < n_sent; s++) {
    for (uint32_t i = off[s]; i < off[s + 1]; i++) {
      const int len = spell_cased(ids[i], i == off[s], w);
      const uint64_t need = (uint64_t)(len > 0 ? len : 0) + 1;
      if (out && at + need <= cap) { memcpy(out + at, w, need - 1); out[at + need - 1] = (i + 1 == off[s + 1]) ? '\n' : ' '; }
      at += need;
    }
    if (off[s] == off[s + 1]) { if (out && at + 1 <= cap) out[at] = '\n'; at += 1; }
  }
  return at;
}

/* The two NUL-separated spelling streams _tokenize_and_normalize makes of those lines (what a caller hands fmx_match_batch_text as
 * `text` and `real_text`): which = 0 the normalised tokens, 1 the real tokens.  Same size convention as fms_pack_text. */
uint64_t fms_pack_cased_stream(const uint32_t* off, const uint32_t* ids, uint64_t n_sent, int which, char* out, uint64_t cap) {
  static const char kNum[] = "\xEF\xBD\x9F" "num" "\xEF\xBD\xA0"; /* ｟num｠ */
  uint64_t at = 0;
  char w[16];
  for (uint64_t s = 0; s < n_sent; s++)
    for (uint32_t i = off[s]; i < off[s + 1]; i++) {
      const uint32_t id = ids[i];
      const char* src = w;
      int len;
      if (id >= 2 && fms_is_number(id - 2)) {
        if (which == 0) { src = kNum; len = (int)sizeof kNum - 1; }
        else len = spell_cased(id, 0, w);
      } else if (id >= 2) {
        if (which == 0) len = spell_cased(id, 0, w);
        else { w[0] = i == off[s] ? 'C' : 'L'; w[1] = '\0'; len = 1; }
      } else
        len = spell_cased(id, 0, w); /* "" / "?": no case, real token = the token */
      if (len < 0) len = 0;
      if (out && at + (uint64_t)len + 1 <= cap) { memcpy(out + at, src, (size_t)len); out[at + len] = '\0'; }
      at += (uint64_t)len + 1;
    }
  return at;
}
