/*
 * fm_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the SYSTRAN/fuzzy-match `FuzzyMatch::match()` hot path
 * (suffix-array probe, n-gram registration, coverage filter, float edit distance,
 * cost-bound heap, top-N, contrastive rerank).  It exists so that the HIP path can be
 * checked bit-for-bit against the reference's algorithm; it is NOT a product path and
 * nothing under fuzzy-match_amd/ may include, link or call it.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).  The reference itself is unbuildable in this image (needs Boost,
 * OpenNMT Tokenizer, ICU glue), so the oracle is pinned against the reference's own
 * known-answer tests and fixtures instead: test/test.cc:223-262,337-632 and
 * test/data/{tm1.fmi,tm1.old.fmi,test-tm1} — see tests/test_oracle_golden.py.
 */
#ifndef FM_ORACLE_H
#define FM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fmo_index fmo_index;

/* include/fuzzy/costs.hh:7-29 + the match() argument list (include/fuzzy/fuzzy_match.hh:72-84) */
typedef struct fmo_params {
  float    fuzzy;
  uint32_t nmatch;          /* number_of_matches, 0 = no limit */
  int32_t  no_perfect;
  int32_t  min_subseq_length;
  float    min_subseq_ratio;
  float    idf_penalty;     /* vocab_idf_penalty */
  float    insert_cost, delete_cost, replace_cost;
  float    contrast;        /* contrastive_factor */
  int32_t  reduce;          /* 0 = MEAN, 1 = MAX (fuzzy_match.hh:15) */
  int32_t  contrast_buffer; /* -1 => nmatch */
} fmo_params;

/* FuzzyMatch::Match minus the host-only members (fuzzy_match.hh:32-46) */
typedef struct fmo_rec {
  float    score;
  float    penalty;
  int32_t  max_subseq;
  uint32_t s_id;
} fmo_rec;

/* SURVEY §8(d) algorithmic-byte accounting, counted on the reference algorithm's logical reads */
typedef struct fmo_counters {
  uint64_t equal_range_calls;
  uint64_t compare_steps;    /* start_by() calls */
  uint64_t probe_bytes;      /* sum over steps of 6 + 4 + 4 + 4*c_t */
  uint64_t suffix_visits;    /* suffixes walked by register_suffix_range_match */
  uint64_t suffix_kept;      /* passed the length filter */
  uint64_t reg_bytes;        /* 2*visits + 4*kept */
  uint64_t candidates;       /* entries of get_longest_matches() */
  uint64_t cand_bytes;       /* sum 4 + 4*(s_len+1) */
  uint64_t coverage_tokens;  /* tokens scanned by count_covered_words */
  uint64_t dp_calls;
  uint64_t dp_cells;
  uint64_t out_bytes;        /* 16 * nmatch per query */
  uint64_t queries;
  uint64_t queries_with_match;
} fmo_counters;

/* ---- index construction (SuffixArray / SuffixArrayIndex build side) ---- */
fmo_index* fmo_index_create(uint64_t max_tokens_in_pattern);
void       fmo_index_free(fmo_index*);
/* SuffixArray::add_sentence (src/suffix_array.cc:9-27); real_tokstring / itoks may be NULL/0
 * (then the sentence's real tokens are taken to be identical to its normalised tokens). */
int        fmo_index_add(fmo_index*, const uint32_t* ids, uint32_t n,
                         const char* real_tokstring,
                         uint32_t n_itoks, const uint64_t* itok_idx, const char* const* itok_str);
/* bulk add of id-only sentences: CSR offsets (n_sent+1) */
int        fmo_index_add_bulk(fmo_index*, const uint32_t* off, const uint32_t* toks, uint32_t n_sent);
/* VocabIndexer read side: vocab size and per-word sentence frequency (vocab_indexer.cc:73-90) */
int        fmo_index_set_vocab(fmo_index*, uint32_t vocab_size, const uint32_t* sfreq);
/* SuffixArray::sort (src/suffix_array.cc:58-102) with comp (:235-251); nthreads only speeds it up */
int        fmo_index_sort(fmo_index*, int nthreads);
/* adopt already-sorted arrays (what SuffixArray::load leaves in memory, suffix_array.hxx:55-91) */
int        fmo_index_set_arrays(fmo_index*,
                                const uint32_t* suf_sid, const uint16_t* suf_pos, uint64_t n_suf,
                                const uint32_t* sentence_buffer, uint64_t n_buf,
                                const uint32_t* sentence_pos, uint64_t n_sent,
                                const uint32_t* qva, uint64_t n_qva);
/* VocabIndexer::sfreq of an index adopted with set_arrays, counted by the oracle itself from the sentence buffer */
int        fmo_index_count_vocab(fmo_index*, uint32_t vocab_size);
/* attach real tokens / itoks to an existing sentence (for indexes adopted with set_arrays) */
int        fmo_index_set_real(fmo_index*, uint32_t s_id, const char* real_tokstring,
                              uint32_t n_itoks, const uint64_t* itok_idx, const char* const* itok_str);

/* ... for sentences [first_sid, first_sid + n), none with penalty tokens: sentence i's tab-joined real tokens = the NUL-terminated
 * string at packed + off[i] */
int        fmo_index_set_real_bulk(fmo_index*, uint32_t first_sid, uint32_t n, const char* packed, const uint64_t* off);

uint64_t        fmo_index_num_suffixes(const fmo_index*);
uint64_t        fmo_index_num_sentences(const fmo_index*);
uint64_t        fmo_index_buffer_size(const fmo_index*);
uint64_t        fmo_index_qva_size(const fmo_index*);
const uint32_t* fmo_index_suf_sid(const fmo_index*);
const uint16_t* fmo_index_suf_pos(const fmo_index*);
const uint32_t* fmo_index_buffer(const fmo_index*);
const uint32_t* fmo_index_sentence_pos(const fmo_index*);
const uint32_t* fmo_index_qva(const fmo_index*);

/* SuffixArray::equal_range (src/suffix_array.cc:105-212); out[0..1] = first, second */
void fmo_equal_range(const fmo_index*, const uint32_t* ngram, uint64_t length,
                     uint64_t min, uint64_t max, uint64_t out[2], fmo_counters* ctr);

/* FuzzyMatch::match(Sentence, Tokens, …) (src/fuzzy_match.cc:435-681) on vocabulary ids.
 * p_real_tokstring / itoks describe `real` (NULL => same as the pattern's normalised tokens).
 * Writes at most `cap` matches; returns the number the reference would have returned
 * (may exceed cap), or a negative value on argument errors. */
int64_t fmo_match(const fmo_index*, const fmo_params*,
                  const uint32_t* p_wids, uint32_t p,
                  const char* p_real_tokstring,
                  uint32_t n_itoks, const uint64_t* itok_idx, const char* const* itok_str,
                  fmo_rec* out, uint64_t cap, fmo_counters* ctr);

/* fmo_match for id-only patterns when the caller's `matches` vector already holds n_prior entries (token sequences as
 * CSR): the contrastive rerank penalises against them first (src/fuzzy_match.cc:622-650).  prm->nmatch = NEW matches. */
int64_t fmo_match_prior(const fmo_index*, const fmo_params*, const uint32_t* p_wids, uint32_t p,
                        const uint32_t* prior_off, const uint32_t* prior_tok, uint32_t n_prior, fmo_rec* out, uint64_t cap);
/* the same with the prior entries' s_ids: the rerank's memo is keyed on (candidate s_id, memory s_id), src/fuzzy_match.cc:627,
 * 637-650 — a memory entry whose id an earlier one carried reuses that one's cost */
int64_t fmo_match_prior_ids(const fmo_index*, const fmo_params*, const uint32_t* p_wids, uint32_t p,
                            const uint32_t* prior_off, const uint32_t* prior_tok, const uint32_t* prior_sid, uint32_t n_prior,
                            fmo_rec* out, uint64_t cap);

/* batch of id-only patterns over `nthreads` workers sharing one const index, results in
 * input order (the CLI's -N pool, cli/src/FuzzyMatch-cli.cc:112-193).  out is nq*cap
 * records, out_count nq entries (clamped to cap).  ctr (optional) receives the summed counters. */
int fmo_match_batch(const fmo_index*, const fmo_params*,
                    const uint32_t* q_off, const uint32_t* q_tok, uint64_t nq,
                    int nthreads, fmo_rec* out, uint32_t* out_count, uint64_t cap,
                    fmo_counters* ctr);

/* fmo_match_batch with every pattern's `real` Sentence: pattern q's tab-joined real tokens = the NUL-terminated string at
 * real_packed + real_off[q] (no penalty tokens); both NULL = fmo_match_batch */
int fmo_match_batch_real(const fmo_index*, const fmo_params*,
                         const uint32_t* q_off, const uint32_t* q_tok, uint64_t nq,
                         const char* real_packed, const uint64_t* real_off,
                         int nthreads, fmo_rec* out, uint32_t* out_count, uint64_t cap,
                         fmo_counters* ctr);

/* FuzzyMatch::subsequence (src/fuzzy_match.cc:250-365) after tokenisation, on vocabulary ids.  Returns 1 and fills
 * *out (score, max_subseq = length of the sub-n-gram, s_id; penalty 0) and *out_position (start of the sub-n-gram in
 * the pattern: the reference appends its detokenised text to Match::id) when a match was found, 0 when not. */
int fmo_subsequence(const fmo_index*, const uint32_t* p_wids, uint32_t p, const char* p_real_tokstring,
                    uint32_t n_itoks, const uint64_t* itok_idx, const char* const* itok_str,
                    uint32_t number_of_matches, int no_perfect, int min_subseq_length, float min_subseq_ratio,
                    int idf_weighting, fmo_rec* out, uint32_t* out_position);

/* pieces exported for unit tests */
float fmo_edit_distance_ids(const uint32_t* s1, int n1, const uint32_t* s2, int n2,
                            float ins, float del, float rep, float diff_word, float max_fuzzyness);
int   fmo_edit_distance_char(const char* s1, int n1, const char* s2, int n2);
float fmo_normalizer(uint64_t p, uint64_t s, float ins, float del, float rep);
float fmo_score_from_cost(float cost);

#ifdef __cplusplus
}
#endif
#endif
