// fmx_core.hh — per-lane building blocks of the match() hot path, shared by every HIP kernel.
//
// Everything here is a plain inline function over raw pointers so that the exact code the
// kernels run per lane can also be compiled for the host by tests/emul (a test-only serial
// harness that checks the GPU decomposition against the oracle without a GPU).  No product
// path runs these on the CPU.
//
// Float discipline: every expression is parenthesised as in the reference and the library is
// built with -ffp-contract=off, so each add/mul/div rounds once, like the reference's x86-64
// SSE build (SURVEY §8a "exact-arithmetic spec").
#pragma once
#include <stdint.h>
#include <float.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FMX_HD __host__ __device__ __forceinline__
#else
#define FMX_HD inline
#endif

namespace fmx {

// match() arguments (include/fuzzy/fuzzy_match.hh:72-84) + per-index constants the kernels need.
struct Params {
  float    fuzzy;
  uint32_t nmatch;
  int32_t  no_perfect;
  int32_t  min_subseq_length;
  float    min_subseq_ratio;
  float    idf_penalty;
  float    ins, del, rep;
  float    contrast;
  int32_t  reduce;
  int32_t  cbuf;        // contrast_buffer after the "-1 => nmatch" rule (src/fuzzy_match.cc:451-452)
  float    idf_max;     // compute_max_idf_penalty (src/fuzzy_match.cc:367-370) or 0.01 when idf is off
  uint32_t max_tokens;  // max_tokens_in_pattern (suffix_array_index.hh:15)
};

// One suffix-array entry as laid out in HBM: everything a compare step or the length filter
// needs in one 8-byte load.  off = index into the token stream of the suffix's first token
// (= _sentence_pos[sid] + subsentence_pos), rem = tokens left in the sentence from there,
// slen = the sentence's length (the reference's _sentence_length cache, suffix_array.hh:61).
struct SufRec {
  uint32_t off;
  uint16_t rem;
  uint16_t slen;
};

// candidate / accepted-match record in the per-query segment
struct Cand {
  uint32_t sid;
  uint32_t longest;
  float    score;
  float    penalty;
};

constexpr uint32_t kNoMatchLen = 0xffffffffu;  // query skipped by the guards

// src/fuzzy_match.cc:455-467 — guards and min_subseq_length normalisation.  Returns the
// unsigned _min_seq_len handed to NGramMatches, or kNoMatchLen when match() returns false early.
FMX_HD uint32_t min_seq_len(uint32_t p, const Params& prm) {
  if (p > prm.max_tokens || p == 0) return kNoMatchLen;
  int ml = prm.min_subseq_length;
  if ((uint64_t)(int64_t)ml > (uint64_t)p) ml = (int)p;  // (std::size_t)(int) > size
  const int by_ratio = (int)(prm.min_subseq_ratio * (float)p);
  if (by_ratio > ml) ml = by_ratio;
  return (uint32_t)ml;
}

// length of n-gram whose suffix range seeds the candidates of one start position: the reference
// never registers unigrams unless p == 1 (src/fuzzy_match.cc:484-493,525,546).
FMX_HD uint32_t seed_len(uint32_t p, uint32_t msl) {
  if (p == 1) return 1;
  return msl < 2 ? 2u : msl;
}

// Costs::get_normalizer — include/fuzzy/costs.hh:33-47
FMX_HD float normalizer(uint32_t p, uint32_t s, float ins, float del, float rep) {
  if (ins == 0.f && del == 0.f && rep == 0.f) return 1.f;
  if (ins + del <= rep) return (ins * (float)p) + (del * (float)s);
  if (p <= s) return ((rep - del) * (float)p) + (del * (float)s);
  return ((rep - ins) * (float)s) + (ins * (float)p);
}

// NGramMatches::theoretical_rejection — src/ngram_matches.cc:31-39
FMX_HD bool reject_len(uint32_t p, uint32_t s, const Params& prm) {
  float d = (float)p - (float)s;
  d = d < 0.f ? -d : d;
  const float rc = (p >= s) ? prm.ins : prm.del;
  const float bound = 1.f - ((rc * d) / normalizer(p, s, prm.ins, prm.del, prm.rep));
  return ((double)bound + 0.000005) < (double)prm.fuzzy;
}

// NGramMatches::theoretical_rejection_cover — src/ngram_matches.cc:41-59
FMX_HD bool reject_cover(uint32_t p, uint32_t s, uint32_t cover, const Params& prm) {
  float bound;
  const float norm = normalizer(p, s, prm.ins, prm.del, prm.rep);
  if (prm.ins + prm.del < prm.rep) {
    bound = 1.f - (((prm.ins * ((float)s - (float)cover)) + (prm.del * ((float)p - (float)cover))) / norm);
  } else {
    const float rc = (p > s) ? prm.ins : prm.del;
    const float mn = (p > s) ? (float)s : (float)p;
    const float mx = (p > s) ? (float)p : (float)s;
    bound = 1.f - (((prm.rep * (mn - (float)cover)) + (rc * (mx - mn))) / norm);
  }
  return ((double)bound + 0.000005) < (double)prm.fuzzy;
}

// score quantisation — src/fuzzy_match.cc:598,651
FMX_HD float score_from_cost(float cost) {
  const float t = 10000.f - (cost * 100.f);
  return (float)((double)(int)t / 10000.0);
}

// SuffixArray::start_by (src/suffix_array.cc:263-273) for a suffix whose first token is already
// known to equal ngram[0] (it lives in that token's _quickVocabAccess bucket): compare tokens
// [from, m).  -1: suffix sorts before every suffix starting with the n-gram (smaller token, or
// the sentence ends first); 0: starts with it; +1: sorts after.
FMX_HD int start_by_from(const uint32_t* __restrict__ tok, SufRec r, const uint32_t* __restrict__ ngram,
                         uint32_t from, uint32_t m) {
  for (uint32_t k = from; k < m; k++) {
    if (k >= r.rem) return -1;
    const uint32_t a = tok[r.off + k], b = ngram[k];
    if (a < b) return -1;
    if (a > b) return 1;
  }
  return 0;
}

// Measurement only (option "probe_count"): bytes and 128-byte lines the probe's loads request, per lane.  It is the
// byte model of THIS algorithm, next to SURVEY §8d's model of the reference's equal_range steps.
struct ProbeTally {
  uint32_t bytes;
  uint32_t lines;
};
FMX_HD void tally_add(ProbeTally* t, uint32_t bytes, uint32_t lines = 1) {
  if (t) { t->bytes += bytes; t->lines += lines; }
}

// Half-open range of suffixes starting with ngram[0..m) — the value SuffixArray::equal_range
// (src/suffix_array.cc:105-212) converges to, computed as a plain lower bound followed by a
// galloping upper bound (ranges of long n-grams are short).  m >= 1.
FMX_HD void mgram_range(const SufRec* __restrict__ suf, const uint32_t* __restrict__ tok,
                        const uint32_t* __restrict__ qva, uint32_t n_qva,
                        const uint32_t* __restrict__ ngram, uint32_t m, uint32_t& out_lo, uint32_t& out_hi,
                        ProbeTally* tally = nullptr) {
  out_lo = out_hi = 0;
  const uint32_t w0 = ngram[0];
  if (w0 + 1 >= n_qva) return;
  uint32_t lo = qva[w0], hi = qva[w0 + 1];
  tally_add(tally, 8);
  if (m == 1 || lo >= hi) { out_lo = lo; out_hi = hi; return; }
  // lower bound: first i with start_by >= 0
  uint32_t a = lo, b = hi;
  while (a < b) {
    const uint32_t mid = a + ((b - a) >> 1);
    if ((tally_add(tally, 8 + 4 * (m - 1), 2), start_by_from(tok, suf[mid], ngram, 1, m)) < 0) a = mid + 1; else b = mid;
  }
  const uint32_t first = a;
  if (first >= hi || (tally_add(tally, 8 + 4 * (m - 1), 2), start_by_from(tok, suf[first], ngram, 1, m)) != 0) { out_lo = out_hi = first; return; }
  // upper bound: gallop from first, then bisect; invariant start_by(a) == 0, start_by(b) > 0 or b == hi
  uint32_t step = 1;
  a = first;
  b = hi;
  while ((unsigned long long)a + step < hi) {
    if ((tally_add(tally, 8 + 4 * (m - 1), 2), start_by_from(tok, suf[a + step], ngram, 1, m)) == 0) { a += step; step <<= 1; }
    else { b = a + step; break; }
  }
  while (b - a > 1) {
    const uint32_t mid = a + ((b - a) >> 1);
    if ((tally_add(tally, 8 + 4 * (m - 1), 2), start_by_from(tok, suf[mid], ngram, 1, m)) == 0) a = mid; else b = mid;
  }
  out_lo = first;
  out_hi = b;
}

// The eight tokens that follow a suffix's first token, stored next to the suffix (0 past the end of the
// sentence): a compare step of the binary search is one 32-byte load instead of record -> tokens for seed
// n-grams of up to nine tokens (patterns of up to ~30 tokens at the default min_subseq_ratio).
constexpr uint32_t kKeyTokens = 8;
struct __attribute__((aligned(32))) SufKey {
  uint32_t t[kKeyTokens];
};

// start_by for the keyed layout: tokens [1, min(m, 9)) from the key, the rest (rare: the key already
// narrowed the range to a handful of suffixes) from the token stream.  Requires ngram[k] != 0 for the
// keyed positions, because a key token 0 means "sentence ended" (id 0 is also the empty word; callers
// fall back to mgram_range when the pattern contains it).
FMX_HD int start_by_keyed(const SufKey* __restrict__ key, const SufRec* __restrict__ suf, const uint32_t* __restrict__ tok,
                          uint32_t i, const uint32_t* __restrict__ ngram, uint32_t m) {
  const SufKey kk = key[i];
  const uint32_t lim = m - 1 < kKeyTokens ? m - 1 : kKeyTokens;
  int r = 0;
#pragma unroll
  for (uint32_t k = 0; k < kKeyTokens; k++) {
    if (k < lim && r == 0) {
      const uint32_t a = kk.t[k], b = ngram[k + 1];
      r = a < b ? -1 : (a > b ? 1 : 0);  // -1: smaller token, or the sentence ended (0)
    }
  }
  if (r != 0 || m - 1 <= kKeyTokens) return r;
  return start_by_from(tok, suf[i], ngram, kKeyTokens + 1, m);
}

// mgram_range over the keyed layout; same result.
FMX_HD void mgram_range_keyed(const SufKey* __restrict__ key, const SufRec* __restrict__ suf,
                              const uint32_t* __restrict__ tok, const uint32_t* __restrict__ qva, uint32_t n_qva,
                              const uint32_t* __restrict__ ngram, uint32_t m, uint32_t& out_lo, uint32_t& out_hi,
                              ProbeTally* tally = nullptr) {
  for (uint32_t k = 1; k < m && k <= kKeyTokens; k++)
    if (ngram[k] == 0) { mgram_range(suf, tok, qva, n_qva, ngram, m, out_lo, out_hi, tally); return; }
  out_lo = out_hi = 0;
  const uint32_t w0 = ngram[0];
  if (w0 + 1 >= n_qva) return;
  uint32_t lo = qva[w0], hi = qva[w0 + 1];
  tally_add(tally, 8);
  if (m == 1 || lo >= hi) { out_lo = lo; out_hi = hi; return; }
  uint32_t a = lo, b = hi;
  while (a < b) {
    const uint32_t mid = a + ((b - a) >> 1);
    if ((tally_add(tally, 32), start_by_keyed(key, suf, tok, mid, ngram, m)) < 0) a = mid + 1; else b = mid;
  }
  const uint32_t first = a;
  if (first >= hi || (tally_add(tally, 32), start_by_keyed(key, suf, tok, first, ngram, m)) != 0) { out_lo = out_hi = first; return; }
  uint32_t step = 1;
  a = first;
  b = hi;
  while ((unsigned long long)a + step < hi) {
    if ((tally_add(tally, 32), start_by_keyed(key, suf, tok, a + step, ngram, m)) == 0) { a += step; step <<= 1; }
    else { b = a + step; break; }
  }
  while (b - a > 1) {
    const uint32_t mid = a + ((b - a) >> 1);
    if ((tally_add(tally, 32), start_by_keyed(key, suf, tok, mid, ngram, m)) == 0) a = mid; else b = mid;
  }
  out_lo = first;
  out_hi = b;
}

// Trigram table: (w0, w1, w2) -> the suffix range of that trigram, open addressing, 16-byte entries.
// w0 is implied by the range (it lies inside w0's _quickVocabAccess bucket).  One or two random reads
// replace the ~20 dependent compare steps of the search through a large first-word bucket; the remaining
// tokens of the seed n-gram are resolved by bisecting the (small) trigram range.
struct GramEntry {
  uint32_t w1, w2, lo, cnt;
};
constexpr uint32_t kGramEmpty = 0xffffffffu;

FMX_HD uint32_t gram_hash(uint32_t w0, uint32_t w1, uint32_t w2) {
  uint32_t h = w0 * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + w1 * 0x85EBCA6Bu;
  h = (h ^ (h >> 13)) + w2 * 0xC2B2AE35u;
  return h ^ (h >> 16);
}

// 5-gram buckets (round 4): EVERY distinct 5-gram of the index, one 128-byte line per probe.  A seed n-gram of five tokens
// or more (patterns of 17 tokens and more at the default min_subseq_ratio: 97 % of the probed starts at C3) goes straight
// to the bucket of its first five tokens instead of trigram entry -> 5-gram entry -> keys, and the bucket is the unit of
// traffic: an entry carries what the next step needs —
//   * the five tokens (w0 implied by the _quickVocabAccess bucket `lo` lies in), the group's suffix range,
//   * the four tokens that follow the 5-gram in the group's FIRST suffix, and how many of them the WHOLE group shares
//     (`lcp`, 0..4: first and last suffix of a sorted group agree on exactly the tokens all of them agree on).
// A seed of up to 5 + lcp tokens is decided inside the bucket (the whole group, or nothing); a longer one is decided when it
// differs from the shared tokens (nothing), and only otherwise continues into the group's keys.  A bucket holds three
// entries and an eight-byte directory — a 16-bit tag per entry (a second hash of the 5-gram) and the bucket's state —, and
// a lane reads the DIRECTORY first: most probed starts name a 5-gram the index does not hold (the pattern was edited
// there) and are through after that one 8-byte load; a tag that matches is followed by the 40 bytes of its entry, out of
// the line the directory just brought in.  (Loading the whole 128 bytes per lane — eight 16-byte loads, each touching 64
// different lines per wave — halved the HBM traffic of the probe and made it SLOWER, 1.78 -> 2.04 ms at C3: the kernel had
// become bound by load instructions, not by lines.)  The table has about one bucket per distinct 5-gram (kG5FillPer16), so a
// bucket overflows into its successors rarely (flag `over`) and an absent 5-gram costs one line.
struct G5Entry {
  uint32_t w1, w2, w3, w4;  // tokens 1..4 of the 5-gram
  uint32_t lo, cnt;         // its suffix range [lo, lo + cnt)
  uint32_t t[4];            // tokens 5..8 of suffix lo (0 past the end of its sentence)
};
struct __attribute__((aligned(128))) G5Bucket {
  G5Entry  e[3];
  uint32_t dir_a;  // tag of entry 0 | tag of entry 1 << 16
  uint32_t dir_b;  // tag of entry 2 | state << 16: bits 0-1 entries in use, bit 2 `over` (an entry whose home is this bucket, or one
                   // that passed through, lives in a later bucket), bits 4 + 3 k .. 6 + 3 k the lcp of entry k
};
constexpr uint32_t kG5PerBucket = 3;
#ifndef FMX_G5_FILL
#define FMX_G5_FILL 16
#endif
constexpr uint32_t kG5FillPer16 = FMX_G5_FILL;  // distinct 5-grams per 16 buckets (16 = one per bucket: 1.9 % of the buckets overflow)
FMX_HD uint32_t g5_used(uint32_t dir_b) { return (dir_b >> 16) & 3u; }
FMX_HD uint32_t g5_over(uint32_t dir_b) { return (dir_b >> 18) & 1u; }
FMX_HD uint32_t g5_lcp_of(uint32_t dir_b, uint32_t k) { return (dir_b >> (20 + 3 * k)) & 7u; }
FMX_HD uint32_t g5_tag_of(uint32_t dir_a, uint32_t dir_b, uint32_t k) { return k == 0 ? (dir_a & 0xffffu) : (k == 1 ? (dir_a >> 16) : (dir_b & 0xffffu)); }
FMX_HD uint32_t gram5_hash(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4) {
  uint32_t h = w0 * 0x9E3779B1u;
  h = (h ^ (h >> 15)) + w1 * 0x85EBCA6Bu;
  h = (h ^ (h >> 13)) + w2 * 0xC2B2AE35u;
  h = (h ^ (h >> 16)) + w3 * 0x27D4EB2Fu;
  h = (h ^ (h >> 15)) + w4 * 0x165667B1u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15;
  return h;
}
// the entry's tag: 16 bits of a SECOND hash of the five tokens (entries of one bucket share most bits of the first)
FMX_HD uint32_t gram5_tag(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4) {
  uint32_t h = w4 * 0x9E3779B1u;
  h = (h ^ (h >> 13)) + w3 * 0xC2B2AE35u;
  h = (h ^ (h >> 15)) + w2 * 0x27D4EB2Fu;
  h = (h ^ (h >> 16)) + w1 * 0x165667B1u;
  h = (h ^ (h >> 13)) + w0 * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0x846CA68Bu;
  return h >> 16;
}
FMX_HD uint32_t g5_bucket_of(uint32_t h, uint32_t n_buckets) { return (uint32_t)(((unsigned long long)h * n_buckets) >> 32); }

// Second level under the 5-gram buckets: the distinct 7-grams of every 5-gram whose range holds more than kDeepGramMin
// suffixes (a repeated phrase), keyed by (start of the 5-gram range, w5, w6), 16-byte entries, open addressing.  It caps the
// length of the dependent-read chain of a probe: without it one lane that has to bisect a 100 000-suffix range keeps its
// whole wave waiting.
#ifndef FMX_DEEP_MIN
#define FMX_DEEP_MIN 8
#endif
constexpr uint32_t kDeepGramMin = FMX_DEEP_MIN;

struct GramTables {
  const GramEntry* g3;   // trigrams (seeds of three and four tokens)
  const G5Bucket*  g5;   // 5-gram buckets (seeds of five tokens and more)
  const GramEntry* g7;   // 7-grams of the heavy 5-grams
  uint32_t         mask3, n_b5, mask7;
};

// [lo, hi) -> the sub-range whose suffixes compare equal under cmp (cmp(i) < 0: before, 0: inside, > 0: after)
template <class Cmp>
FMX_HD void bisect_equal_range(uint32_t lo, uint32_t hi, Cmp cmp, uint32_t& out_lo, uint32_t& out_hi) {
  uint32_t a = lo, b = hi;
  while (a < b) {
    const uint32_t mid = a + ((b - a) >> 1);
    if (cmp(mid) < 0) a = mid + 1; else b = mid;
  }
  const uint32_t first = a;
  if (first >= hi || cmp(first) != 0) { out_lo = out_hi = first; return; }
  uint32_t step = 1;
  a = first;
  b = hi;
  while ((unsigned long long)a + step < hi) {
    if (cmp(a + step) == 0) { a += step; step <<= 1; }
    else { b = a + step; break; }
  }
  while (b - a > 1) {
    const uint32_t mid = a + ((b - a) >> 1);
    if (cmp(mid) == 0) a = mid; else b = mid;
  }
  out_lo = first;
  out_hi = b;
}

// Key tokens [from-1, lim) against the n-gram held in registers (ng[k] = ngram[k] for k < m, anything
// beyond): -1 / 0 / 1 like start_by.  Fully unrolled, no memory access.
FMX_HD int key_cmp(const SufKey& kk, const uint32_t (&ng)[kKeyTokens + 1], uint32_t from, uint32_t lim) {
  int r = 0;
#pragma unroll
  for (uint32_t k = 0; k < kKeyTokens; k++) {
    const uint32_t a = kk.t[k], b = ng[k + 1];
    const int c = a < b ? -1 : (a > b ? 1 : 0);
    if (k + 1 >= from && k < lim && r == 0) r = c;
  }
  return r;
}

// mgram_range through the n-gram tables; same result as mgram_range / mgram_range_keyed.
//
// The probe is a chain of dependent random reads, and each costs the memory system a 128-byte line whatever it requests, so
// the chain is kept short and every line is made to count: the first-word bucket bounds and the table line are requested
// together, the n-gram lives in registers, seeds of five tokens and more are answered by ONE line of the 5-gram buckets
// whenever the group's shared tokens decide them (see G5Bucket), and what is left is bisected over the group's keys with the
// lower and the upper end of the equal range in lockstep (two independent key loads per level).
//
// The function comes in two halves (round 5) so that k_probe can run them as two kernels — the HEAD (one table line: most
// starts end there) with a lane per start position, the TAIL (7-gram entry, bisection: 17 % of the probed starts at C3, three to
// four dependent levels each) with a lane per start position that needs it, in full waves:
//   mgram_range_head  -> kHeadDone: out_lo / out_hi are final; else the range [out_lo, out_hi) of the table entry and `from`,
//                        the number of leading tokens every suffix of that range is known to share with the seed
//                        (kHeadKeyed: the tables cannot be used — id 0 in a keyed position — the tail runs the keyed search)
//   mgram_range_tail  -> the equal range inside it
constexpr uint32_t kHeadDone = 0, kHeadKeyed = 0xffu;

// ng[k] = ngram[k] for k < min(m, 9), 0 beyond
FMX_HD uint32_t mgram_range_head(const GramTables& gt, const uint32_t* __restrict__ qva, uint32_t n_qva,
                                 const uint32_t (&ng)[kKeyTokens + 1], uint32_t m, uint32_t& out_lo, uint32_t& out_hi,
                                 ProbeTally* tally = nullptr) {
  bool usable = gt.g3 != nullptr && m >= 3 && n_qva >= 2;
#pragma unroll
  for (uint32_t k = 1; k <= kKeyTokens; k++) usable = usable && (k >= m || ng[k] != 0);
  out_lo = out_hi = 0;
  if (!usable) return kHeadKeyed;
  const uint32_t w0 = ng[0];
  const bool w_ok = w0 + 1 < n_qva;
  const uint32_t wi = w_ok ? w0 : 0u;
  uint32_t lo, hi;
  if (m >= 5 && gt.g5 != nullptr) {
    // ---- seeds of five tokens and more: the bucket of the first five ----
    uint32_t b = g5_bucket_of(gram5_hash(w0, ng[1], ng[2], ng[3], ng[4]), gt.n_b5);
    const uint32_t tag = gram5_tag(w0, ng[1], ng[2], ng[3], ng[4]);
    const uint32_t blo = qva[wi], bhi = qva[wi + 1];  // (requested together with the bucket's directory)
    tally_add(tally, 8);
    uint32_t lcp = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    bool found = false;
    for (uint32_t hops = 0;; hops++) {
      const G5Bucket* bk = gt.g5 + b;
      const uint32_t dir_a = bk->dir_a, dir_b = bk->dir_b;  // one 8-byte load: most starts end here
      tally_add(tally, 8);
      const uint32_t used = g5_used(dir_b);
#pragma unroll
      for (uint32_t k = 0; k < kG5PerBucket; k++) {
        if (!found && k < used && g5_tag_of(dir_a, dir_b, k) == tag) {
          const G5Entry e = bk->e[k];  // same line as the directory
          tally_add(tally, 40, 0);
          if (e.w1 == ng[1] && e.w2 == ng[2] && e.w3 == ng[3] && e.w4 == ng[4] && e.lo >= blo && e.lo < bhi) {
            found = true; lo = e.lo; hi = e.lo + e.cnt; lcp = g5_lcp_of(dir_b, k);
            t0 = e.t[0]; t1 = e.t[1]; t2 = e.t[2]; t3 = e.t[3];
          }
        }
      }
      if (found || g5_over(dir_b) == 0u || hops >= gt.n_b5) break;
      b = b + 1 == gt.n_b5 ? 0u : b + 1;
    }
    if (!found || !w_ok || blo >= bhi) return kHeadDone;  // no suffix starts with these five tokens
    if (m == 5) { out_lo = lo; out_hi = hi; return kHeadDone; }
    // the tokens the whole group shares after the 5-gram (tokens 5 .. 5 + lcp - 1 = those of its first suffix)
    const uint32_t need = m - 5, sh = need < lcp ? need : lcp;
    bool differ = false;
    differ = differ || (sh > 0 && t0 != ng[5]);
    differ = differ || (sh > 1 && t1 != ng[6]);
    differ = differ || (sh > 2 && t2 != ng[7]);
    differ = differ || (sh > 3 && t3 != ng[8]);
    if (differ) return kHeadDone;                                        // every suffix of the group continues otherwise
    if (need <= lcp) { out_lo = lo; out_hi = hi; return kHeadDone; }      // every suffix of the group continues like the seed
    out_lo = lo; out_hi = hi;
    return 5 + lcp;
  }
  // ---- seeds of three and four tokens: the trigram table ----
  uint32_t slot = gram_hash(w0, ng[1], ng[2]) & gt.mask3;
  GramEntry e = gt.g3[slot];  // requested together with the bucket bounds
  const uint32_t blo = qva[wi], bhi = qva[wi + 1];
  tally_add(tally, 16); tally_add(tally, 8);
  if (!w_ok || blo >= bhi) return kHeadDone;
  for (;;) {
    if (e.lo == kGramEmpty) return kHeadDone;  // no suffix starts with this trigram
    if (e.w1 == ng[1] && e.w2 == ng[2] && e.lo >= blo && e.lo < bhi) { lo = e.lo; hi = e.lo + e.cnt; break; }
    slot = (slot + 1) & gt.mask3;
    e = gt.g3[slot];
    tally_add(tally, 16, 0);  // linear probing: the next slot is almost always in the same line
  }
  out_lo = lo; out_hi = hi;
  if (m == 3) return kHeadDone;
  return 3;
}

// [lo, hi) and `from` as mgram_range_head left them (from != kHeadDone)
FMX_HD void mgram_range_tail(const GramTables& gt, const SufKey* __restrict__ key, const SufRec* __restrict__ suf,
                             const uint32_t* __restrict__ tok, const uint32_t* __restrict__ qva, uint32_t n_qva,
                             const uint32_t* __restrict__ ngram, const uint32_t (&ng)[kKeyTokens + 1], uint32_t m, uint32_t from,
                             uint32_t lo, uint32_t hi, uint32_t& out_lo, uint32_t& out_hi, ProbeTally* tally = nullptr) {
  if (from == kHeadKeyed) { mgram_range_keyed(key, suf, tok, qva, n_qva, ngram, m, out_lo, out_hi, tally); return; }
  out_lo = out_hi = 0;
  // heavy 5-gram groups: the 7-gram entry (unless the shared tokens already cover w5, w6: the entry would name the whole group)
  if (from >= 5 && from < 7 && m >= 7 && gt.g7 != nullptr && hi - lo > kDeepGramMin) {
    uint32_t slot = gram_hash(lo, ng[5], ng[6]) & gt.mask7;
    for (;;) {
      const GramEntry e7 = gt.g7[slot];
      tally_add(tally, 16);
      if (e7.lo == kGramEmpty) return;  // no suffix of this 5-gram continues with (w5, w6)
      if (e7.w1 == ng[5] && e7.w2 == ng[6] && e7.lo >= lo && e7.lo < hi) { lo = e7.lo; hi = e7.lo + e7.cnt; break; }
      slot = (slot + 1) & gt.mask7;
    }
    if (m == 7) { out_lo = lo; out_hi = hi; return; }
    from = 7;
  }
  // equal range of the remaining tokens inside [lo, hi): a1 -> first suffix >= n-gram, a2 -> first suffix > n-gram
  const uint32_t lim = m - 1 < kKeyTokens ? m - 1 : kKeyTokens;
  const bool deep = m - 1 > kKeyTokens;  // tokens beyond the key come from the token stream
  uint32_t a1 = lo, b1 = hi, a2 = lo, b2 = hi;
  while (a1 < b1 || a2 < b2) {
    const bool act1 = a1 < b1, act2 = a2 < b2;
    uint32_t m1 = a1 + ((b1 - a1) >> 1), m2 = a2 + ((b2 - a2) >> 1);
    if (!act1) m1 = m2;
    if (!act2) m2 = m1;
    const SufKey k1 = key[m1];
    const SufKey k2 = key[m2];
    tally_add(tally, 32); if (m2 != m1) tally_add(tally, 32);
    SufRec r1, r2;
    r1.off = 0; r1.rem = 0; r1.slen = 0; r2 = r1;
    if (deep) { r1 = suf[m1]; r2 = suf[m2]; tally_add(tally, 8); if (m2 != m1) tally_add(tally, 8); }  // requested with the keys: a tie costs one more level, not two
    int c1 = key_cmp(k1, ng, from, lim);
    int c2 = key_cmp(k2, ng, from, lim);
    if (deep) {
      if (c1 == 0) { c1 = start_by_from(tok, r1, ngram, kKeyTokens + 1, m); tally_add(tally, 4 * (m - kKeyTokens - 1)); }
      if (c2 == 0) { if (m2 != m1) tally_add(tally, 4 * (m - kKeyTokens - 1)); c2 = m2 == m1 ? c1 : start_by_from(tok, r2, ngram, kKeyTokens + 1, m); }
    }
    if (act1) { if (c1 < 0) a1 = m1 + 1; else b1 = m1; }
    if (act2) { if (c2 <= 0) a2 = m2 + 1; else b2 = m2; }
  }
  out_lo = a1;
  out_hi = a2 > a1 ? a2 : a1;
}

// head + tail in one go (the host twin, the one-launch kernel for micro-batches, subsequence())
FMX_HD void mgram_range_hashed(const GramTables& gt, const SufKey* __restrict__ key,
                               const SufRec* __restrict__ suf, const uint32_t* __restrict__ tok,
                               const uint32_t* __restrict__ qva, uint32_t n_qva, const uint32_t* __restrict__ ngram,
                               uint32_t m, uint32_t& out_lo, uint32_t& out_hi, ProbeTally* tally = nullptr) {
  uint32_t ng[kKeyTokens + 1];
  tally_add(tally, 4 * (m < kKeyTokens + 1 ? m : kKeyTokens + 1));  // the seed n-gram itself
#pragma unroll
  for (uint32_t k = 0; k <= kKeyTokens; k++) ng[k] = k < m ? ngram[k] : 0u;
  uint32_t lo, hi;
  const uint32_t from = mgram_range_head(gt, qva, n_qva, ng, m, lo, hi, tally);
  if (from == kHeadDone) { out_lo = lo; out_hi = hi; return; }
  mgram_range_tail(gt, key, suf, tok, qva, n_qva, ngram, ng, m, from, lo, hi, out_lo, out_hi, tally);
}

// Longest common prefix of a suffix (already known to share `known` tokens) with pat[0..maxlen).
// This is the match_length the reference's narrowing loop ends up registering for the suffix
// (src/fuzzy_match.cc:500-550).
FMX_HD uint32_t lcp_extend(const uint32_t* __restrict__ tok, SufRec r, const uint32_t* __restrict__ pat,
                           uint32_t known, uint32_t maxlen) {
  uint32_t lim = r.rem < maxlen ? r.rem : maxlen;
  uint32_t k = known;
  while (k < lim && tok[r.off + k] == pat[k]) k++;
  return k;
}

// A hit of start `it` on a suffix is dominated when the preceding tokens also agree: start it-1
// then reaches the same sentence with a strictly longer match and the same length filter, so
// dropping the hit cannot change max-per-sentence (NGramMatches keeps only the max,
// src/ngram_matches.cc:79-81).  r.slen - r.rem = tokens before the suffix in its sentence.
FMX_HD bool dominated_hit(const uint32_t* __restrict__ tok, SufRec r, const uint32_t* __restrict__ pat_at_it,
                          uint32_t it) {
  return it > 0 && r.rem < r.slen && tok[r.off - 1] == pat_at_it[-1];
}

// Phase-B record: SufRec plus the token in front of the suffix and the sentence id, so that one 16-byte
// load decides the dominance test and names the survivor (two dependent random reads fewer per queued hit).
struct SufRecB {
  uint32_t off;
  uint16_t rem;
  uint16_t slen;
  uint32_t prev;  // token before the suffix in its sentence (unspecified when the suffix starts the sentence)
  uint32_t sid;
};

FMX_HD bool dominated_hit_b(const SufRecB& r, const uint32_t* __restrict__ pat_at_it, uint32_t it) {
  return it > 0 && r.rem < r.slen && r.prev == pat_at_it[-1];
}

// 48-bit word signature of a sentence: bit sig_bit(w) is set for every word w of the sentence.  For a
// pattern, the number of positions whose word's bit is set is an upper bound of
// PatternCoverage::count_covered_words, and theoretical_rejection_cover is monotone in the cover
// (for non-negative costs), so a candidate rejected with the upper bound is rejected by the
// reference too: the exact token scan only runs for the few that pass.
// Packed word: [ sentence length : 8, 255 = "255 tokens or more" | signature : 56 ].  The signature's false
// positive rate falls quickly with the number of buckets (queued hits at C3: 40 buckets 2.5x, 48 buckets
// 1x), so the length gets only the bits the length filter needs; clamped lengths are decided in phase B.
constexpr uint32_t kSigBits = 56;
constexpr uint32_t kSigLenClamp = 255;
FMX_HD uint32_t sig_bit(uint32_t w) { return (uint32_t)(((unsigned long long)(w * 0x9E3779B1u) * kSigBits) >> 32); }
FMX_HD unsigned long long sig_pack(uint32_t slen, unsigned long long sig) {
  return ((unsigned long long)(slen < kSigLenClamp ? slen : kSigLenClamp) << kSigBits) | sig;
}
// Bit planes (k_expand_h) use twice the buckets: bucket sig_bit(w) is split in two by one more hash bit, so a sentence's
// 56-bit signature is the OR of the two halves of its 112-bit one and suf_sig / k_expand_a stay as they are.  At ~25
// words per sentence a bucket is set with probability 0.20 instead of 0.36: the chance that a dozen absent pattern words
// all look present — what lets a hit through the plane test only to fail the exact coverage test in k_expand_b — falls
// by two orders of magnitude, for ~10 % more plane rows per pattern (distinct buckets of its words: 22.5 instead of 20).
constexpr uint32_t kSigBits2 = 2 * kSigBits;
FMX_HD uint32_t sig_half(uint32_t w) { return (w * 0x85EBCA6Bu) >> 31; }
FMX_HD uint32_t sig_bit2(uint32_t w) { return sig_bit(w) + kSigBits * sig_half(w); }
FMX_HD uint32_t sig_slen(unsigned long long packed) { return (uint32_t)(packed >> kSigBits); }  // clamped at 255
FMX_HD unsigned long long sig_bits(unsigned long long packed) { return packed & ((1ull << kSigBits) - 1ull); }

FMX_HD uint32_t sig_cover_upper(unsigned long long sig, const uint8_t* __restrict__ hbit, uint32_t p) {
  uint32_t c = 0;
  for (uint32_t j = 0; j < p; j++) c += (uint32_t)((sig >> hbit[j]) & 1ull);
  return c;
}

// smallest cover that passes theoretical_rejection_cover for a sentence of s tokens (the bound is
// monotone in the cover for non-negative costs), 0xffff when the length filter
// (theoretical_rejection) or every cover up to p rejects.
FMX_HD uint32_t cover_threshold(uint32_t p, uint32_t s, const Params& prm) {
  if (reject_len(p, s, prm) || reject_cover(p, s, p, prm)) return 0xffffu;
  uint32_t lo = 0, hi = p;  // invariant: hi passes
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (reject_cover(p, s, mid, prm)) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// PatternCoverage::count_covered_words (src/pattern_coverage.cc:15-28): the sum of counts of the
// distinct pattern words present in the sentence == number of pattern positions whose word
// occurs in the sentence.
FMX_HD uint32_t cover_count(const uint32_t* __restrict__ sent, uint32_t s_len, const uint32_t* __restrict__ pat,
                            uint32_t p) {
  uint32_t c = 0;
  for (uint32_t j = 0; j < p; j++) {
    const uint32_t w = pat[j];
    bool found = false;
    for (uint32_t k = 0; k < s_len && !found; k++) found = sent[k] == w;
    c += found ? 1u : 0u;
  }
  return c;
}

FMX_HD float fmin2(float a, float b) { return b < a ? b : a; }  // std::min(a, b)

// Optional per-candidate side inputs of the full _edit_distance (src/edit_distance.cc:5-77).
struct DpSide {
  const uint32_t* s_real;   // sentence real-token ids (bit31 = first byte in "LUMC" or empty), per token slot
  const uint32_t* p_real;   // pattern real-token ids, per pattern position
  const uint32_t* s_itok;   // sentence itok ids, slot i in [0, n1]
  const uint32_t* p_itok;   // pattern itok ids, slot j in [0, n2]
  const uint16_t* itok_dist;  // (n_itok x n_itok) byte-Levenshtein table, id 0 = no itok
  uint32_t        n_itok;
  const float*    p_idf;    // idf_penalty[j] per pattern position (nullptr when idf is off)
  float           idf_weight;
};

// Float edit distance between an indexed sentence (rows, n1) and the pattern (columns, n2).
// Returns `final` = arr[n1][n2]; M = max(max_i rowmin_i, final) is what the reference's
// row-min early exit compares against its bound (src/edit_distance.cc:73-74,
// src/fuzzy_match.cc:595): the candidate is rejected iff M > bound.  If a row minimum exceeds
// `stop_above` the walk stops early and M (> stop_above) is all that is meaningful.
// Row is any accessor with float& operator[](uint32_t j), j in [0, n2].
template <bool FULL, class Row>
FMX_HD float edit_distance(const uint32_t* __restrict__ s1, uint32_t n1, const uint32_t* __restrict__ s2,
                           uint32_t n2, float ins, float del, float rep, float dw, const DpSide& side,
                           float stop_above, Row row, float& M_out) {
  const float del_dw = del * dw, ins_dw = ins * dw, rep_dw = rep * dw;
  const bool  idf = FULL && side.p_idf != nullptr && side.idf_weight != 0.f;
  const bool  itk = FULL && side.itok_dist != nullptr;
  auto T = [&](uint32_t i, uint32_t j) -> float {
    if (!itk) return 0.f;
    const uint32_t a = side.s_itok ? side.s_itok[i] : 0u, b = side.p_itok ? side.p_itok[j] : 0u;
    return (float)side.itok_dist[a * side.n_itok + b];
  };
  // row 0
  float v = FULL ? T(n1, n2) : 0.f;  // arr[0][0]: trailing penalty tokens
  row[0] = v;
  for (uint32_t j = 1; j <= n2; j++) {
    v = (v + ins_dw);
    if (FULL) {
      if (itk) v = v + (float)side.itok_dist[side.p_itok ? side.p_itok[j] : 0u];  // sn2[j]
      if (idf) v = v + (side.p_idf[j - 1] * side.idf_weight);
    }
    row[j] = v;
  }
  float M = 0.f;
  float final_v = row[n2];
  for (uint32_t i = 1; i <= n1; i++) {
    const uint32_t w = s1[i - 1];
    float diag = row[0];
    float left = diag + del_dw;
    if (FULL && itk) left = left + (float)side.itok_dist[(side.s_itok ? side.s_itok[i] : 0u) * side.n_itok];  // sn1[i]
    row[0] = left;
    float mn = FLT_MAX;
    for (uint32_t j = 1; j <= n2; j++) {
      const float up = row[j];
      float pen = 0.f;
      if (idf) pen = side.p_idf[j - 1] * side.idf_weight;
      float diff = 0.f;
      if (w != s2[j - 1]) {
        diff = FULL ? (rep_dw + pen) : rep_dw;
      } else if (FULL && side.s_real != nullptr) {
        const uint32_t r1 = side.s_real[i - 1], r2 = side.p_real[j - 1];
        if ((r1 & 0x7fffffffu) != r2) diff = (r1 >> 31) ? (rep * 1.f) : (rep * 2.f);
      }
      float a, b, c;
      if (FULL) {
        a = (up + del_dw) + T(i - 1, j);
        b = ((left + ins_dw) + T(i, j - 1)) + pen;
        c = (diag + diff) + T(i - 1, j - 1);
      } else {
        a = up + del_dw;
        b = left + ins_dw;
        c = diag + diff;
      }
      const float d = fmin2(fmin2(a, b), c);
      row[j] = d;
      mn = fmin2(mn, d);
      diag = up;
      left = d;
    }
    if (mn > M) M = mn;
    final_v = left;
    if (mn > stop_above) { M_out = M; return mn; }
  }
  if (n1 == 0) final_v = row[n2];
  if (final_v > M) M = final_v;
  M_out = M;
  return final_v;
}

// lowest_costs (std::priority_queue<float>, src/fuzzy_match.cc:567-602) as an ascending array:
// top() = last element.
struct CostHeap {
  float*   v;
  uint32_t n;
  FMX_HD float top() const { return v[n - 1]; }
  FMX_HD void  push(float c) {
    uint32_t i = n++;
    while (i > 0 && v[i - 1] > c) { v[i] = v[i - 1]; i--; }
    v[i] = c;
  }
  FMX_HD void pop() { if (n) n--; }
};

// ---------------------------------------------------------------------------------------------
// Unit costs (insert = delete = replace = c >= 0, no idf / real-token / penalty-token terms): the float DP
// of _edit_distance (src/edit_distance.cc:5-77) only ever adds the same increment inc = c * dw or 0, so a
// cell whose integer edit distance is k holds exactly S_k = fl(fl(fl(inc + inc) + inc) ...), k additions
// (induction over the recurrence: every predecessor holds S_a, S_a + inc = S_{a+1}, min(S_x, S_y) =
// S_min(x,y) because S is non-decreasing).  The last cell therefore is S_k with k the plain Levenshtein
// distance, and M = max(row minima, final) = final (row minima never decrease for non-negative costs).
// k comes from Hyyro's bit-parallel recurrence for the global distance (pattern positions = bits; the
// "<< 1 | 1" sets D[i][0] = i): ~20 word operations per sentence token plus the match mask, instead of
// seven float operations per DP cell.  tests/emul checks S_k against the float DP on every candidate.
// ---------------------------------------------------------------------------------------------
struct LevState {
  unsigned long long pv, mv;
  uint32_t score;
};
FMX_HD void lev_init(LevState& st, uint32_t m) { st.pv = ~0ull; st.mv = 0ull; st.score = m; }
// eq: bit j set iff pattern token j equals the sentence token of this row; m = pattern length (1..64)
FMX_HD void lev_step(LevState& st, unsigned long long eq, uint32_t m) {
  const unsigned long long xv = eq | st.mv;
  const unsigned long long xh = (((eq & st.pv) + st.pv) ^ st.pv) | eq;
  unsigned long long ph = st.mv | ~(xh | st.pv);
  unsigned long long mh = st.pv & xh;
  st.score += (uint32_t)((ph >> (m - 1)) & 1ull) - (uint32_t)((mh >> (m - 1)) & 1ull);
  ph = (ph << 1) | 1ull;
  mh = mh << 1;
  st.pv = mh | ~(xv | ph);
  st.mv = ph & xv;
}
FMX_HD void lev_step32(uint32_t& pv, uint32_t& mv, uint32_t& score, uint32_t eq, uint32_t m) {  // m <= 32
  const uint32_t xv = eq | mv;
  const uint32_t xh = (((eq & pv) + pv) ^ pv) | eq;
  uint32_t ph = mv | ~(xh | pv);
  uint32_t mh = pv & xh;
  score += ((ph >> (m - 1)) & 1u) - ((mh >> (m - 1)) & 1u);
  ph = (ph << 1) | 1u;
  mh = mh << 1;
  pv = mh | ~(xv | ph);
  mv = ph & xv;
}
FMX_HD float unit_cost_sum(uint32_t k, float inc) {
  float a = 0.f;
  for (uint32_t i = 0; i < k; i++) a = a + inc;
  return a;
}
// plain-loop version (host / emulation; the kernels keep the pattern in registers)
FMX_HD uint32_t lev_distance(const uint32_t* __restrict__ sent, uint32_t n1, const uint32_t* __restrict__ pat, uint32_t n2) {
  LevState st;
  lev_init(st, n2);
  for (uint32_t i = 0; i < n1; i++) {
    unsigned long long eq = 0;
    for (uint32_t j = 0; j < n2; j++) eq |= (unsigned long long)(sent[i] == pat[j]) << j;
    lev_step(st, eq, n2);
  }
  return st.score;
}

// One turn of the candidate loop (src/fuzzy_match.cc:587-610) given the candidate's DP outcome.
// Returns true when the candidate becomes a Match (score >= fuzzy); score_out is its score.
// `score` = score_from_cost(final_cost), computed by the caller (the kernels evaluate it for 64 candidates in
// parallel before the sequential walk: the double-precision division is most of a turn's instructions).
FMX_HD bool replay_candidate_scored(CostHeap& heap, const Params& prm, uint32_t p, uint32_t s_len, float final_cost,
                                    float M, float score) {
  const float ub = heap.top();
  if ((prm.no_perfect && final_cost == 0.f && s_len == p) || M > ub) return false;
  heap.push(final_cost);
  if (score < prm.fuzzy || (prm.cbuf > 0 && heap.n > (uint32_t)prm.cbuf)) heap.pop();
  return score >= prm.fuzzy;
}
FMX_HD bool replay_candidate(CostHeap& heap, const Params& prm, uint32_t p, uint32_t s_len, float final_cost,
                             float M, float& score_out) {
  score_out = score_from_cost(final_cost);
  return replay_candidate_scored(heap, prm, p, s_len, final_cost, M, score_out);
}

// order-preserving float -> uint32 (ascending)
FMX_HD uint32_t float_key(float f) {
  union { float f; uint32_t u; } x;
  x.f = f;
  return (x.u & 0x80000000u) ? ~x.u : (x.u | 0x80000000u);
}

}  // namespace fmx
