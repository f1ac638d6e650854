/*
 * fmx.h — C ABI of the MI355X-native fuzzy-match lookup engine (libfmx.so).
 *
 * The reference (SYSTRAN/fuzzy-match) has no FFI layer: its boundary is the C++ class
 * fuzzy::FuzzyMatch in libFuzzyMatch.so.  This header is the C-ABI a binding for that class
 * would sit on; each entry point cites the reference interface it replaces (paths relative to
 * the reference checkout).  The C++ facade fuzzy-match_amd/include/fuzzy/fuzzy_match.hh
 * re-creates the reference's class on top of it; INTEGRATION.md shows the bindings.
 *
 * Conventions: plain pointers and sizes, no exceptions across the boundary, every function
 * returns FMX_OK (0) or a negative fmx_status; fmx_last_error() gives the message for the calling
 * thread.  Index handles are immutable once fmx_index_finalize() has run and may be shared
 * between threads; result buffers are caller-owned.  There is no CPU fallback: every match entry
 * point fails with FMX_E_NO_DEVICE when no gfx950 device is usable.
 */
#ifndef FMX_H
#define FMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum fmx_status {
  FMX_OK = 0,
  FMX_E_INVALID = -1,     /* bad argument */
  FMX_E_IO = -2,          /* file missing / unreadable / corrupt (import throws in the reference) */
  FMX_E_NO_DEVICE = -3,   /* no usable GPU, or HIP call failed */
  FMX_E_NOMEM = -4,
  FMX_E_STATE = -5,       /* index not sorted / not finalized / not on the device */
  FMX_E_UNSUPPORTED = -6  /* a documented limit of this build was hit (the message names it) */
} fmx_status;

/* fmx_index_add_tm: the sentence was ignored exactly like the reference ignores it (empty, or longer than
 * max_tokens_in_pattern; src/suffix_array_index.cc:16).  Distinct from every fmx_status. */
#define FMX_TM_SKIPPED (-100)

typedef struct fmx_index fmx_index;

/* Arguments of FuzzyMatch::match() — include/fuzzy/fuzzy_match.hh:72-84, EditCosts
 * include/fuzzy/costs.hh:7-29, ContrastReduce fuzzy_match.hh:15 (0 = MEAN, 1 = MAX). */
typedef struct fmx_params {
  float    fuzzy;
  uint32_t nmatch;            /* number_of_matches; 0 = no limit */
  int32_t  no_perfect;
  int32_t  min_subseq_length; /* reference defaults: 3 (Sentence/string API), 2 (Tokens API) */
  float    min_subseq_ratio;  /* 0.3 / 0 */
  float    idf_penalty;       /* vocab_idf_penalty */
  float    insert_cost, delete_cost, replace_cost;
  float    contrast;          /* contrastive_factor */
  int32_t  reduce;
  int32_t  contrast_buffer;   /* -1 => nmatch (src/fuzzy_match.cc:451-452) */
} fmx_params;

/* FuzzyMatch::Match (include/fuzzy/fuzzy_match.hh:32-46) without the host-side members; the facade
 * fills id / length / s from the host copy of the index. */
typedef struct fmx_match {
  float    score;
  float    penalty;
  int32_t  max_subseq;
  uint32_t s_id;
} fmx_match;

/* A batch of patterns as CSR over vocabulary ids (what VocabIndexer::getIndex returns,
 * src/vocab_indexer.cc:52-71).  q_real / q_itok carry the `real` Sentence argument of
 * match(const Sentence&, const Tokens&, ...) and may be NULL (real tokens = normalised tokens, no
 * penalty tokens):
 *   q_real[q_off[q] + j]      interned id of the j-th real token (fmx_index_lookup_real)
 *   q_itok[q_off[q] + q + j]  interned id of the penalty string in front of real token j,
 *                             j in [0, p] (fmx_index_intern_itok; 0 = none) — nq + T entries. */
typedef struct fmx_batch {
  uint64_t        nq;
  const uint32_t* q_off;  /* nq + 1 offsets into q_tok */
  const uint32_t* q_tok;
  const uint32_t* q_real;
  const uint32_t* q_itok;
  uint32_t        max_len; /* longest pattern of the batch if known, else 0 (computed on the device) */
} fmx_batch;

/* per-call measurements (all optional reading) */
typedef struct fmx_stats {
  double   ms_total;        /* wall time of the call on the host */
  double   ms_h2d, ms_d2h;
  double   ms_kernel[16];   /* HIP-event time per pipeline stage, summed over sub-batches; see fmx_stage_name */
  uint64_t launches[16];
  uint64_t n_hits;          /* suffix-array hits visited by the expand stage */
  uint64_t n_kept;          /* hits that survived the length / dominance filters */
  uint64_t n_candidates;    /* distinct (query, sentence) candidates */
  uint64_t n_dp;            /* edit-distance evaluations */
  uint64_t n_retries;       /* sub-batches re-run with a larger workspace */
  uint64_t n_big;           /* queries whose survivor set needed the 1024-thread sorter */
  uint64_t n_phase_b;       /* hits that passed the length + signature tests (tokens were read) */
  uint64_t max_kept;        /* largest survivor set of a single query */
  uint64_t n_heavy_hits;    /* of n_hits: hits of long ranges, tested 64 per lane through the signature bit planes */
  uint64_t n_heavy_pass;    /* of those: hits that passed the plane test and were re-tested exactly */
  uint64_t pinned;          /* 1 if the host buffers of fmx_match_batch were pinned (copies overlapped compute) */
  uint64_t plane_words;     /* 8-byte bit-plane words k_expand_h read (its byte model: 8 x this) */
  uint64_t probe_bytes;     /* option "probe_count": bytes k_probe's loads requested (its byte model) */
  uint64_t probe_lines;     /* option "probe_count": distinct 128-byte lines those loads touched, per lane */
  uint64_t n_light_hits;    /* of n_hits: hits of "light" queries, walked by k_expand_b itself (no phase A, no queue) */
  uint64_t n_small;         /* queries answered by the one-launch path for micro-batches (k_match_small) */
  uint64_t n_rerank;        /* sentence-pair edit distances of the flattened contrastive rerank (k_cround) */
} fmx_stats;

const char* fmx_last_error(void);
const char* fmx_version(void);
const char* fmx_stage_name(int stage);

/* ---- devices --------------------------------------------------------------------------- */
int fmx_device_count(void);                 /* number of usable gfx950 devices, 0 if none */
/* Pinned host memory for batch buffers (q_off / q_tok / out / out_count ...): with these fmx_match_batch
 * overlaps the PCIe copies of one pass with the kernels of its neighbours.  Pageable buffers work too. */
void* fmx_host_alloc(size_t bytes);
void  fmx_host_free(void*);

/* ---- index lifetime ---------------------------------------------------------------------
 * fuzzy::FuzzyMatch(int pt, size_t max_tokens_in_pattern) — include/fuzzy/fuzzy_match.hh:48-49 */
fmx_index* fmx_index_create(int32_t pt, uint64_t max_tokens_in_pattern);
void       fmx_index_free(fmx_index*);
/* import_binarized_fuzzy_matcher — include/fuzzy/fuzzy_matcher_binarization.hh:9,
 * src/fuzzy_matcher_binarization.cc:32-51 (the reference throws; here NULL + error text) */
fmx_index* fmx_index_load(const char* fmi_path, char* err, size_t errlen);
/* Adopt the members of the reference's SuffixArray as they are — an index built and SORTED elsewhere (include/fuzzy/
 * suffix_array.hh:50-61: _suffixes as parallel sentence-id / 1-based position arrays, _sentence_buffer = [len, tok x len, 0]
 * per sentence, _sentence_pos, _quickVocabAccess) plus VocabIndexer::sfreq (include/fuzzy/vocab_indexer.hh:40).
 * quick_vocab_access and sfreq may be NULL (recomputed).  The arrays are copied and validated (structure, and the
 * SuffixArray::comp order of adjacent suffixes, src/suffix_array.cc:235-251); the result is a sorted id-only index:
 * fmx_index_finalize + fmx_index_to_device make it searchable, fmx_index_save writes it as .fmi.  NULL + error text on
 * failure (SURVEY §8b lists this entry point next to fmx_index_load). */
fmx_index* fmx_index_from_arrays(int32_t pt, uint64_t max_tokens_in_pattern, const uint32_t* suf_sid, const uint16_t* suf_pos,
                                 uint64_t n_suffixes, const uint32_t* sentence_buffer, uint64_t buffer_size,
                                 const uint32_t* sentence_pos, uint64_t n_sentences, const uint32_t* quick_vocab_access,
                                 uint64_t n_qva, const uint32_t* sfreq, uint32_t vocab_size, char* err, size_t errlen);
/* export_binarized_fuzzy_matcher — fuzzy_matcher_binarization.hh:8 */
int        fmx_index_save(const fmx_index*, const char* fmi_path);

/* FuzzyMatch::add_tm(id, Sentence real, Tokens norm, sort=false) — include/fuzzy/fuzzy_match.hh:52,
 * src/suffix_array_index.cc:10-30.  real_tokstring = real tokens joined by '\t' (NULL: same as norm,
 * the Tokens overload fuzzy_match.hh:51).  Returns the sentence id (>= 0), FMX_TM_SKIPPED when the reference
 * would have ignored the sentence, or a negative fmx_status. */
int64_t fmx_index_add_tm(fmx_index*, const char* id, const char* const* norm, uint32_t n_norm,
                         const char* real_tokstring, uint32_t n_itoks, const uint64_t* itok_idx,
                         const char* const* itok_str);
/* Bulk variant on vocabulary ids (>= 2) for corpora that never had string forms (benchmarks). */
int fmx_index_add_ids(fmx_index*, const uint32_t* off, const uint32_t* toks, uint64_t n_sent,
                      uint32_t vocab_size);
/* FuzzyMatch::sort() — include/fuzzy/fuzzy_match.hh:56, src/suffix_array.cc:58-102.
 * device < 0: CPU build with `nthreads` workers; device >= 0: suffix sort on that GPU.
 * Both produce exactly the reference's suffix order. */
int fmx_index_sort(fmx_index*, int device, int nthreads);
/* derive the SoA arrays (after sort or load) and freeze the index */
int fmx_index_finalize(fmx_index*);
/* copy the frozen index into the HBM of `device` (a HIP ordinal of a gfx950 part) */
int fmx_index_to_device(fmx_index*, int device);
/* Replicate the frozen index on several devices (index replicated, queries sharded).  fmx_match_batch
 * and fmx_match_one then cut every batch into contiguous query blocks, one per device, run them concurrently and
 * write each block's records straight into the caller's buffers (single process, host gather; the RCCL gather
 * of the one-process-per-GPU deployment lives in fuzzy-match_amd/shard.py).  Devices already holding the index
 * are kept.  fmx_set_devices is the name SURVEY §8b gives this entry point. */
int fmx_index_to_devices(fmx_index*, const int* devices, int n_devices);
int fmx_set_devices(fmx_index*, const int* devices, int n_devices);
/* Thaw a finalized index (drops the device copies and the derived arrays) so that add_tm can be called again —
 * the reference allows add_tm after sort() (include/fuzzy/fuzzy_match.hh:51-56). */
int fmx_index_reopen(fmx_index*);

/* ---- index queries (host) ---------------------------------------------------------------- */
/* VocabIndexer::getIndex — src/vocab_indexer.cc:52-71 (unknown -> 1) */
int fmx_vocab_lookup(const fmx_index*, const char* const* toks, uint64_t n, uint32_t* ids);
/* the same for the token stream of a whole batch, spread over `nthreads` host threads (no std::string per token: the per-query
 * host prep of match(), src/fuzzy_match.cc:468-470, at batch scale) */
int fmx_vocab_lookup_mt(const fmx_index*, const char* const* toks, uint64_t n, uint32_t* ids, int nthreads);
/* attach spellings to an index built from ids (fmx_index_add_ids): spellings[k] = the word of id k for k in [2, n), NULL = none */
int fmx_index_set_forms(fmx_index*, const char* const* spellings, uint64_t n);
uint32_t fmx_index_lookup_real(const fmx_index*, const char* s, size_t len);
uint32_t fmx_index_intern_itok(fmx_index*, const char* s, size_t len);
/* SuffixArrayIndex::id — src/suffix_array_index.cc:66-70; "" for id-only corpora */
const char* fmx_index_sentence_id(const fmx_index*, uint32_t s_id);

typedef enum fmx_array {
  FMX_A_SUF_SID = 0,   /* u32  x n_suffixes   SuffixView::sentence_id, suffix-array order */
  FMX_A_SUF_POS = 1,   /* u16  x n_suffixes   SuffixView::subsentence_pos (1-based) */
  FMX_A_BUFFER = 2,    /* u32                 _sentence_buffer */
  FMX_A_SENT_POS = 3,  /* u32  x n_sentences  _sentence_pos */
  FMX_A_QVA = 4,       /* u32  x vocab+1      _quickVocabAccess */
  FMX_A_SFREQ = 5,     /* u32  x vocab        VocabIndexer::sfreq */
  FMX_A_SUF_REC = 6,   /* 8 B  x n_suffixes   {u32 off; u16 rem; u16 slen} (derived) */
  FMX_A_IDF = 7,       /* f32  x vocab        (derived) */
  FMX_A_REAL_IDS = 8,  /* u32  per buffer slot, empty when real tokens == normalised tokens */
  FMX_A_ITOK_IDS = 9,  /* u32  per buffer slot */
  FMX_A_ITOK_DIST = 10,/* u16  n_itok x n_itok */
  FMX_A_SUF_SIG = 11,  /* 8 B x n_suffixes    min(sentence length, 255) << 56 | 56-bit word signature of the sentence (derived) */
  FMX_A_SUF_KEY = 12,  /* 32 B x n_suffixes   the eight tokens after the suffix's first token (derived) */
  FMX_A_GRAM = 13,     /* 16 B x 2^k          trigram -> suffix range hash table {w1, w2, lo, cnt} (derived) */
  FMX_A_GRAM5 = 14,    /* 128 B x n_buckets   5-gram buckets: three entries {w1..w4, lo, cnt, tokens 5..8 of suffix lo} + directory {16-bit tags, entries in use, overflow flag, lcp per entry} (derived) */
  FMX_A_SUF_B = 15,    /* 16 B x n_suffixes   {u32 off; u16 rem; u16 slen; u32 previous token; u32 sentence id} (derived) */
  FMX_A_GRAM7 = 16     /* 16 B x 2^k          7-gram table of the heavy 5-grams {w5, w6, lo, cnt} (derived) */
} fmx_array;
/* borrowed pointer to a host array of the index, valid until the index changes or is freed */
int fmx_index_array(const fmx_index*, int which, const void** data, uint64_t* count);
/* The "(derived)" per-suffix arrays and n-gram tables (FMX_A_SUF_REC, _SUF_SIG, _SUF_KEY, _GRAM, _GRAM5, _GRAM7, _SUF_B) are built
 * on the device by fmx_index_to_device; fmx_index_array builds their host twins on request (tests, CPU emulation).
 * fmx_index_device_array copies one of them out of the HBM of replica `replica` as it is used by the kernels:
 * dst == NULL only reports the element count. */
int fmx_index_device_array(const fmx_index*, int replica, int which, void* dst, uint64_t capacity_bytes, uint64_t* count);

typedef struct fmx_index_info {
  uint64_t n_sentences, n_suffixes, buffer_size, vocab_size, max_tokens_in_pattern;
  int32_t  pt, sorted, finalized, on_device, device, real_differs;
  uint32_t n_itok;
  float    idf_max;
  uint64_t device_bytes;
} fmx_index_info;
int fmx_index_get_info(const fmx_index*, fmx_index_info*);

/* ---- the hot path ------------------------------------------------------------------------
 * FuzzyMatch::match(const Sentence&, const Tokens&, ...) — include/fuzzy/fuzzy_match.hh:72-84,
 * src/fuzzy_match.cc:435-681, for a whole batch of patterns.
 * out: nq * cap records, out_count: nq entries (matches written per query, <= cap),
 * out_total (optional): nq entries, the number of matches the reference would return (> cap means
 * truncated; only possible when nmatch == 0 or nmatch > cap).  Results are in input order and,
 * per query, in the reference's order. */
int fmx_match_batch(fmx_index*, const fmx_params*, const fmx_batch* host_batch,
                    fmx_match* out, uint32_t* out_count, uint32_t* out_total, uint32_t cap,
                    fmx_stats* stats);
/* fmx_match_batch that also leaves the records (and counts) in device memory of the index's first device, for a
 * collective that follows — the RCCL gather of the one-process-per-GPU deployment (fuzzy-match_amd/shard.py). */
int fmx_match_batch_keep(fmx_index*, const fmx_params*, const fmx_batch* host_batch,
                         fmx_match* out, uint32_t* out_count, uint32_t* out_total, uint32_t cap,
                         fmx_match* d_keep, uint32_t* d_keep_count, fmx_stats* stats);
/* fmx_match_batch with the patterns as token SPELLINGS instead of vocabulary ids: q_off[nq] NUL-terminated, non-empty
 * spellings back to back in `text` (n_bytes bytes; pinned memory for asynchronous copies), q_off[0] == 0.  The per-pattern
 * host prep of match() — VocabIndexer::getIndex per token, src/fuzzy_match.cc:468-470, src/vocab_indexer.cc:52-60 — runs on
 * the device (k_vocab_lookup: token starts by a scan over the bytes, one look-up per start in the device copy of the spelling
 * table; unknown spellings are VOCAB_UNK).  FMX_E_INVALID when a text does not hold exactly q_off[nq] spellings.
 * Indexes whose real tokens differ from their normalised tokens (the reference CLI's default -p nbr,tag,cas: a word's real
 * token is its case feature, a number's its digits, src/fuzzy_match.cc:123-192) get the `real` Sentence of
 * match(const Sentence&, const Tokens&, ...) the same way:
 *   real_text   q_off[nq] spellings, the real token of every pattern token (what _tokenize_and_normalize put into `real`);
 *               interned on the device against the index's real-token table (a spelling the index never saw differs from
 *               every indexed one, like on the host).  NULL: the pattern tokens' own spellings — match(const Tokens&, ...),
 *               whose real sentence is the pattern itself (src/fuzzy_match.cc:402-416).  An empty real token cannot be
 *               written as a spelling: such patterns go through fmx_match_batch.
 *   itok_slot / itok_id   the batch's penalty tokens (Sentence::set_itok) as n_itoks pairs: slot = q_off[q] + q + j for the
 *               string in front of real token j of pattern q (the layout of fmx_batch::q_itok), id = fmx_index_intern_itok of
 *               the string (interning stays on the host: a new string grows the distance table there).
 * All three are ignored by an index without real tokens.  The struct has no size field: ZERO-INITIALISE it (`fmx_text_batch tb = {0};`)
 * so that members a caller does not know about read as "absent"; ids the index has not interned and slots beyond the batch are
 * refused (FMX_E_INVALID) before anything reaches the device. */
typedef struct fmx_text_batch {
  uint64_t        nq;
  const uint32_t* q_off;    /* nq + 1 offsets in tokens */
  const char*     text;
  uint64_t        n_bytes;
  const char*     real_text;      /* optional, see above */
  uint64_t        n_real_bytes;
  uint64_t        n_itoks;        /* optional */
  const uint64_t* itok_slot;
  const uint32_t* itok_id;
} fmx_text_batch;
int fmx_match_batch_text(fmx_index*, const fmx_params*, const fmx_text_batch*, fmx_match* out, uint32_t* out_count,
                         uint32_t* out_total, uint32_t cap, fmx_stats* stats);
/* fmx_match_batch_text that also leaves the records (and counts) in device memory, like fmx_match_batch_keep */
int fmx_match_batch_text_keep(fmx_index*, const fmx_params*, const fmx_text_batch*, fmx_match* out, uint32_t* out_count,
                              uint32_t* out_total, uint32_t cap, fmx_match* d_keep, uint32_t* d_keep_count, fmx_stats* stats);
/* The two halves of fmx_match_batch_text, like fmx_match_batch_device_begin / _end below: _begin enqueues the copies in, the id
 * lookup and every pass of the call and returns; _end waits, brings the records home and reports what _text would have
 * reported.  ONE host thread keeps several batches in flight this way (each in its own context: at most "contexts" pending, a
 * further _begin is refused with FMX_E_INVALID), the batches' copies and kernels overlap on the GPU.  The batch's buffers and
 * the output buffers must stay valid (and pinned, for the copies to be asynchronous) until _end; _end on the thread that
 * called _begin, once per handle. */
typedef struct fmx_pending fmx_pending;
int fmx_match_batch_text_begin(fmx_index*, const fmx_params*, const fmx_text_batch*, fmx_match* out, uint32_t* out_count,
                               uint32_t* out_total, uint32_t cap, fmx_match* d_keep /* optional, like _keep */, uint32_t* d_keep_count,
                               fmx_pending** pending);
int fmx_match_batch_text_end(fmx_index*, fmx_pending* pending);
/* fmx_match_batch for PRE-TOKENISED LINES as they sit in a file: `text` = n_bytes bytes, tokens separated by blanks (' ', '\t',
 * '\r'), one pattern per line, EVERY line ended by '\n' (n_lines of them; an empty line is an empty pattern: no match).  The host
 * does not look at a token: the bytes cross PCIe as they are (pinned memory for asynchronous copies), the device cuts them into
 * patterns and tokens, looks the vocabulary ids up (k_lines_count / k_lines_emit / k_vocab_lookup) and matches — the reference
 * CLI's read -> tokens -> getIndex -> match() per line (cli/src/FuzzyMatch-cli.cc:112-193, src/fuzzy_match.cc:468-470) for input
 * that is tokenised already.  On an index with real tokens the pattern's own spellings are its real tokens
 * (match(const Tokens&, ...)).  Records: like fmx_match_batch, one row of `cap` per line.  FMX_E_INVALID when the text does not
 * hold exactly n_lines line feeds. */
int fmx_match_batch_lines(fmx_index*, const fmx_params*, const char* text, uint64_t n_bytes, uint64_t n_lines,
                          fmx_match* out, uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_stats* stats);
/* RAW text lines: FuzzyMatch::match(const std::string& pattern, ...) (include/fuzzy/fuzzy_match.hh:86-95, src/fuzzy_match.cc:393-412:
 * _tokenize_and_normalize, then match(Sentence, Tokens, ...)) for a block of '\n'-terminated UTF-8 lines, the reference CLI's
 * only input (cli/src/FuzzyMatch-cli.cc:219-233).  Lines made of ASCII letters, digits, punctuation and blanks are tokenised
 * on the device (the aggressive tokenizer is a byte-class machine on that alphabet) when the index's penalty tokens are among
 * tag / nbr / cas; every other line is tokenised by the host front-end (fmx_text_tokenize's code: NFC, scripts, placeholders)
 * and answered with its real tokens and penalty tokens like fmx_match_batch would.  Records: one row of `cap` per line, in
 * input order. */
int fmx_match_batch_raw_lines(fmx_index*, const fmx_params*, const char* text, uint64_t n_bytes, uint64_t n_lines,
                              fmx_match* out, uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_stats* stats);
/* Same with every buffer already in the HBM of the index's device; `stream` is a hipStream_t
 * (NULL = a stream of the library after a device-wide synchronisation).  total_tokens = q_off[nq] when q_off[0] == 0, else
 * 0 (with it and max_len set, a batch of one pass needs no device read up front; the value is CHECKED on the device against
 * q_off[0] == 0 and q_off[nq]: a call whose total_tokens disagrees is inert on the device (no kernel walks a token) and returns
 * FMX_E_INVALID — from _end for the two-halves form).  Work is enqueued on `stream`; the call
 * synchronises that stream before returning.  stats != NULL: every pass is synchronised and timed per stage (HIP events);
 * stats == NULL: the passes run back to back; concurrent calls overlap on the GPU (with the option "device_turns" set, calls
 * of 262 144 queries or more are chained like concurrent fmx_match_batch calls, see "turns"). */
int fmx_match_batch_device(fmx_index*, const fmx_params*, const fmx_batch* device_batch,
                           uint64_t total_tokens, fmx_match* d_out, uint32_t* d_out_count,
                           uint32_t* d_out_total, uint32_t cap, void* stream, fmx_stats* stats);

/* The two halves of fmx_match_batch_device(stats = NULL): _begin enqueues every launch of the call and returns, _end waits
 * for it (and re-runs a pass whose workspace overflowed: first calls on an index only).  ONE host thread can thus keep
 * several batches in flight — each in its own context ("contexts" option: at most that many pending) — and the library
 * overlaps them on the GPU: the kernels of one call that are bound by HBM run beside those of the others that are bound by
 * instruction issue (six calls in flight over eight hardware queues measured best; the library asks for the queues with
 * GPU_MAX_HW_QUEUES=8 when it is loaded, which only works before the process's first HIP call).  stream = NULL: a stream of the call's context; the caller's inputs
 * must be complete.  The output buffers are final after _end.  Call _end on the thread that called _begin, once per handle. */
int fmx_match_batch_device_begin(fmx_index*, const fmx_params*, const fmx_batch* device_batch, uint64_t total_tokens,
                                 fmx_match* d_out, uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap, void* stream,
                                 fmx_pending** pending);
int fmx_match_batch_device_end(fmx_index*, fmx_pending* pending);

/* FuzzyMatch::match for ONE pattern, as the reference's callers issue it from N threads
 * (cli/src/FuzzyMatch-cli.cc:139-147,254-257).  Thread-safe and blocking; concurrent callers are coalesced into
 * micro-batches behind the scenes (flushed when the GPU is free, or after "coalesce_us" at the latest), so the
 * launch overhead of a pass is shared.  q_real: p entries or NULL; q_itok: p + 1 entries or NULL. */
int fmx_match_one(fmx_index*, const fmx_params*, const uint32_t* wids, uint32_t p, const uint32_t* q_real,
                  const uint32_t* q_itok, fmx_match* out, uint32_t cap, uint32_t* out_count, uint32_t* out_total);

/* fmx_match_one when the caller's `matches` vector already holds entries and contrastive_factor > 0: the reference's
 * rerank loop penalises every new candidate against ALL of `matches` — the entries that were there before the call
 * first, in their order, then its own picks (src/fuzzy_match.cc:622-650, "for (Match &match_memory : matches)").
 * prior_off / prior_tok: the token sequences (Match::s, Match::length) of the n_prior entries as CSR.  params->nmatch is
 * the number of NEW matches wanted (the facade passes number_of_matches - matches.size()).  Not coalesced. */
int fmx_match_one_prior(fmx_index*, const fmx_params*, const uint32_t* wids, uint32_t p, const uint32_t* q_real,
                        const uint32_t* q_itok, const uint32_t* prior_off, const uint32_t* prior_tok, uint32_t n_prior,
                        fmx_match* out, uint32_t cap, uint32_t* out_count, uint32_t* out_total);
/* The same with the entries' Match::s_id (n_prior ids).  The reference memoises the edit cost of a (candidate, entry of `matches`)
 * pair under the two s_ids (src/fuzzy_match.cc:627, 637-650): an entry of `matches` whose id an earlier entry carries — matches
 * of another translation memory, or a new match whose id a prior entry happens to have — is scored through the earlier entry's
 * cost whatever its own tokens are.  With the ids this entry point reproduces that; fmx_match_one_prior (no ids) treats every
 * prior entry as an id of its own. */
int fmx_match_one_prior_ids(fmx_index*, const fmx_params*, const uint32_t* wids, uint32_t p, const uint32_t* q_real,
                            const uint32_t* q_itok, const uint32_t* prior_off, const uint32_t* prior_tok,
                            const uint32_t* prior_sid, uint32_t n_prior, fmx_match* out, uint32_t cap, uint32_t* out_count,
                            uint32_t* out_total);

/* FuzzyMatch::subsequence(...) — include/fuzzy/fuzzy_match.hh:99-105, src/fuzzy_match.cc:250-365, for a batch of
 * tokenised patterns: the sub-n-gram of the pattern with the largest weight (its length, or its summed idf when
 * idf_weighting) that occurs in an indexed sentence, and the best of up to number_of_matches sentences containing
 * it.  Per query: out_found (0 / 1), one record (score, max_subseq = length of the sub-n-gram, s_id) and the start
 * of the sub-n-gram in the pattern (the reference appends its text to Match::id). */
int fmx_subsequence_batch(fmx_index*, const fmx_batch* host_batch, uint32_t number_of_matches, int32_t no_perfect,
                          int32_t min_subseq_length, float min_subseq_ratio, int32_t idf_weighting,
                          fmx_match* out, uint32_t* out_found, uint32_t* out_position);

/* ---- text front-end ------------------------------------------------------------------------
 * FuzzyMatch::_tokenize_and_normalize — src/fuzzy_match.cc:108-193: tokenise a sentence and split it into the
 * normalised tokens (`pattern`), the real tokens and the penalty strings (`real`) according to the penalty-token mask
 * pt.  The tokenizer is a restricted restatement of the one the reference configures (src/fuzzy_match.cc:70-91,
 * OpenNMT Tokenizer, aggressive mode) — see fuzzy-match_amd/csrc/fmx_text.hh for what it covers.  The sentence is first
 * NFC-normalised by ICU like the reference does (src/fuzzy_match.cc:43-58), through the libicuuc of the system found at
 * run time (without one: a warning on stderr once, and the input is taken as NFC).  The handle owns every string it returns. */
typedef struct fmx_text fmx_text;
fmx_text*   fmx_text_tokenize(int32_t pt, const char* sentence, size_t len);
/* the general tokenizer alone: ASCII sentences otherwise take a byte-class fast path with the same output (tests compare the two) */
fmx_text*   fmx_text_tokenize_general(int32_t pt, const char* sentence, size_t len);
void        fmx_text_free(fmx_text*);
uint32_t    fmx_text_num_tokens(const fmx_text*);               /* normalised tokens */
const char* fmx_text_norm(const fmx_text*, uint32_t i);
const char* fmx_text_real(const fmx_text*, uint32_t i);
uint32_t    fmx_text_num_itoks(const fmx_text*);                /* penalty strings, ascending real-token index */
uint64_t    fmx_text_itok_index(const fmx_text*, uint32_t k);
const char* fmx_text_itok(const fmx_text*, uint32_t k);
uint32_t    fmx_text_num_raw(const fmx_text*);                  /* tokenizer output (`tokens`, src/fuzzy_match.cc:113) */
const char* fmx_text_raw(const fmx_text*, uint32_t i);
const char* fmx_text_feature(const fmx_text*, uint32_t i);      /* case feature of raw token i (`features[0][i]`; "N" without pt_cas) */
uint32_t    fmx_text_map(const fmx_text*, uint32_t k);          /* map_tokens[k], k in [0, num_tokens] */
/* detokenised raw tokens [map[from], map[to]) — what subsequence() appends to Match::id (src/fuzzy_match.cc:341-348) */
const char* fmx_text_detokenize(fmx_text*, uint32_t from, uint32_t to);
/* FuzzyMatch::add_tm(id, sentence) — src/fuzzy_match.cc:213-224: FMX_TM_SKIPPED for an empty segment (the reference
 * warns on stderr and returns false) */
int64_t fmx_index_add_text(fmx_index*, const char* id, const char* sentence, size_t len);
/* n sentences at once — FuzzyMatch::add_tm(id, sentence) per line like the reference CLI's import loop
 * (cli/src/FuzzyMatch-cli.cc:32-79): tokenised on every usable CPU, then added in order by one thread (vocabulary ids and sentence
 * ids as n calls of fmx_index_add_text leave them).  ids (optional, as a whole or per entry) NUL-terminated; sids_out (optional):
 * the sentence id of each line or FMX_TM_SKIPPED.  Returns the number of sentences added, or a negative status. */
int64_t fmx_index_add_text_bulk(fmx_index*, const char* const* ids, const char* const* sentences, const size_t* lens, uint64_t n,
                                int64_t* sids_out);

/* tuning knobs (all have sane defaults):
 *   "contexts"          concurrent calls served at once per device, each with its own workspace (default 2)
 *   "heavy_min"         suffix ranges at least this long are tested through the bit planes (default 1024, 0 = never)
 *   "coalesce_us"       longest time fmx_match_one waits for company (default 100; a lone caller never waits)
 *   "host_lanes"        host threads (each with its own workspace and streams) one fmx_match_batch call spreads a
 *                       device's queries over (default 1: measured no better than one lane with the pass schedule
 *                       below; concurrent callers overlap anyway, every call works in its own context)
 *   "probe_count"       measurement: k_probe also counts the bytes and lines it requests (fmx_stats.probe_*)
 *   "probe_split"       the probe as two kernels, head (one table line per start position) + tail (bisection of what is left, in
 *                       full waves) — default 1; 0 = one kernel (round 4's layout, same results)
 *   "max_host_active"   host-buffer calls (fmx_match_batch*, _text*, _lines*) that may have work on the GPU at once, whatever
 *                       "contexts" says (default 4; further synchronous callers wait on the host, further _begin calls are queued
 *                       in order); a caller that pipelines _begin / _end from ONE thread may raise it to 5 (bench.py does); more
 *                       than that is slower, not catastrophic (until round 5 seven and more crawled: a high-priority stream per call).
 *                       Such calls also start at least 0.5 ms apart per device (environment FMX_HOST_SPACING_US)
 *   "sub_queries"       queries per pipeline pass (default 1 048 576)
 *   "host_sub_queries"  pass size of fmx_match_batch (host buffers, default 1 048 576): the inputs of the next pass and
 *                       the records of the previous one cross PCIe while a pass computes
 *   "device_turns"      the same for the device-resident entry points only (default 0: they simply overlap; "turns" sets both)
 *   "turns"             how concurrent host-buffer calls of 32 768 queries or more share the GPU (smaller calls simply overlap): 2 (default)
 *                       = a call's first kernel waits until its predecessor's probe and long-range kernels are through, so its
 *                       HBM-bound probe runs beside the predecessor's issue-bound second half; 1 = it waits for the predecessor's
 *                       last kernel; 0 = calls simply overlap; 3 = turns are taken on the host
 *   "light_max"         queries with at most this many suffix-array hits, none of them in a long range, are walked by
 *                       one kernel without the phase-A filter (default 256, at most 1024; 0 = none)
 *   "real_hybrid"       indexes with real tokens under unit costs: 1 = the bit-parallel kernel answers the pairs in which no equal word
 *                       meets a different real token and leaves the others to the float kernel (default 0: the float kernel for all is
 *                       faster at C3, profiles/r06_real_hybrid_ab.md); same records either way
 *   "expand_order", "b_grid", "heavy_grid", "serial"   measurement / design-time switches of the expand stage
 *   "initial_segment"   initial capacity of the hit workspace (grows on demand; sizes are learned per index) */
int fmx_set_option(fmx_index*, const char* name, int64_t value);

#ifdef __cplusplus
}
#endif
#endif
