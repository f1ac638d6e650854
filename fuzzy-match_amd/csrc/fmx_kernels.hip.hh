// fmx_kernels.hip.hh — the HIP kernels of the match() hot path for gfx950 (wave64).
//
// Stage map (DESIGN.md §4 has the table with what bounds each kernel):
//   k_setup, k_tidf, k_cmin, k_phist / k_prank   A   guards, min_subseq_length, token -> query map, idf per position, the
//                 threshold table of the length / coverage filters, pattern-length ranks
//   k_probe       C,D one lane per (query, start): range of the seed n-gram in the suffix array (n-gram hash tables + keys)
//   k_caps_scan, k_seg_scan, k_scan_lb   work-item lists and segment offsets (single-pass look-back scans, capacities fused in)
//   k_expand_b<true>   D,E,F,G light queries (<= 256 hits, short ranges): ranges walked, dominance + length filter, match
//                 length, exact coverage, max per sentence and candidate order, one wave per query
//   k_expand_h / k_expand_a + k_expand_b<false>   the other queries: 64 hits per lane through the signature bit planes /
//                 hit by hit with the signature bound, then the same token work on the queued hits
//   k_cand_route, k_cand, k_cand_mid, k_cand_big   E,F survivor sets of more than 64: max per sentence, (longest desc, s_id asc)
//   k_classify, k_cmap   dense candidate lists by pattern-length class
//   k_dp_reg<P, IDF, UNIT>, k_dp<FULL, ROWS_IN_HBM>   I   one lane per (query, candidate): bit-parallel / register / LDS-row DP
//   k_finish_pair, k_finish<SORT_CAP, LIST>   H   replay of the lowest_costs bound in reference order, top-N
//   k_cacc, k_camap, k_cround<P>, k_contrast<MODE>   J   contrastive rerank (flattened rounds; per-query kernel for the tail
//                 and for calls that bring prior matches)
// All integer/compare/float-scalar work: no MFMA anywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "fmx_core.hh"

namespace fmx {

struct DevIndex {
  const SufRec*   suf;
  const unsigned long long* suf_sig;  // sentence length (8 bits, clamped) << 56 | 56-bit word signature
  const SufKey*   suf_key;    // eight tokens after the first, per suffix
  GramTables      gram;       // trigram table + 5-gram table of the heavy trigrams (null if absent)
  const SufRecB*  suf_b;      // phase-B record: SufRec + previous token + sentence id
  const uint32_t* tok;
  const uint32_t* sent_pos;
  const uint32_t* qva;
  const float*    idf;
  const uint32_t* real;       // nullptr unless real tokens differ from normalised tokens
  const uint32_t* itok;
  const uint16_t* itok_dist;
  uint32_t        n_itok;
  uint32_t        n_qva;
  uint32_t        n_sent;
  uint32_t        max_slen;
  uint64_t        n_suf;
  // Bit planes of suf_sig (built on the device at upload): planes[k * n_blk + blk], bit i = bit k of the signature of
  // suffix 64 * blk + i (rows 0..55 = the signature buckets).  Plane-major, so a wave that walks 64 consecutive blocks
  // reads 512 contiguous bytes per row and only the rows its pattern uses.
  const unsigned long long* planes;
  // Thermometer planes of the stored (clamped) sentence length: thermo[(s - 1) * n_blk + blk], bit i = "stored length
  // of suffix 64 * blk + i >= s", for s = 1 .. n_thermo (= the longest stored length, at most 255).  A length interval
  // [a, b] of 64 suffixes is T[a] & ~T[b + 1]: two loads instead of a bit-sliced compare.
  const unsigned long long* thermo;
  uint64_t        n_blk;
  uint32_t        n_thermo;
};

struct DevBatch {
  const uint32_t* q_off;   // absolute offsets (nq_total + 1)
  const uint32_t* q_tok;
  const uint32_t* q_real;
  const uint32_t* q_itok;
  uint32_t        q_begin;  // first query of this sub-batch
  uint32_t        nq;       // queries in this sub-batch
  uint32_t        t_begin;  // q_off[q_begin]
  uint32_t        nt;       // tokens in this sub-batch
  // contrastive rerank against matches the caller's vector already holds (src/fuzzy_match.cc:622-650): their token
  // sequences (Match::s, Match::length) as CSR, shared by every query of the batch; n_prior == 0: none
  const uint32_t* prior_off;
  const uint32_t* prior_tok;
  uint32_t        n_prior;
  const uint32_t* prior_sid;  // their Match::s_id (the memo of the rerank is keyed on ids), or null
};

struct QHit;

struct DevWork {
  uint32_t* q_m;        // nq   seed n-gram length or kNoMatchLen
  uint32_t* t_q;        // nt   token -> local query
  uint32_t* t_lo;       // nt   range start per (query, start)
  uint32_t* t_cnt;      // nt   range size
  unsigned long long* q_hits;  // nq   hits per query (every range of the query)
  unsigned long long* q_hits_a;  // nq  hits of the ranges k_expand_a walks (sum of t_cnt: the long ranges go to k_expand_h)
  unsigned long long* q_cap;   // nq+1 spare scan arrays; q_cap + nq + 1 = q_capl
  unsigned long long* q_qseg;  // nq+1 queue offsets into qb: exclusive scan of the capacities (an upper bound of what phase A queues)
  unsigned long long* q_seg;   // nq+1 segment offsets into seg / c_qi / keys ...: light queries first (q_segl), then the exclusive scan of the
                               //      hits phase A really queued for the others (k_seg_scan merges the two)
  unsigned long long* q_capl;  // nq+1 scan input beside q_cap: segment size of a "light" query (its hits; 0 = not light)
  unsigned long long* q_segl;  // nq+1 its exclusive scan: segment offsets of the light queries (known before phase A runs)
  uint32_t* q_scap;     // nq   segment capacity per query (k_seg_scan)
  uint32_t* q_kept;     // nq
  uint32_t* q_cand;     // nq
  uint32_t* q_acc;      // nq
  uint32_t* q_pb;       // nq   hits queued by k_expand_a for k_expand_b
  unsigned long long* q_best;  // 3 x nq  contrastive rerank: the bids of three consecutive rounds (k_cround)
  uint32_t* q_rank;     // nq   position of the query in pattern-length order (DP lists are built in that order)
  uint32_t* p_hist;     // 2 x kPBins: queries per pattern length, then the placement cursors
  const uint16_t* cmin_tab;  // (p_max + 1) x kCminTab: per (pattern length, stored sentence length) the phase-A threshold
  unsigned long long* c_in;    // 3 x (nq+1) candidates per query, one array per length class (scan input)
  unsigned long long* c_off;   // 3 x (nq+1) exclusive scans: dense per-class candidate lists
  unsigned long long* c_qi;    // seg_cap  (query << 32 | candidate index), dense per length class
  float*    t_idf;      // nt   idf_penalty per pattern position (src/fuzzy_match.cc:372-390)
  Cand*     seg;        // capacity seg_cap
  QHit*     qb;      // capacity qb_cap: per-query queues between the two expand phases
  float*    heap;       // capacity 2 x seg_cap (lowest_costs spill, two slots per segment entry)
  unsigned long long  seg_cap;
  unsigned long long  qb_cap;
  unsigned long long* counters;  // [0]=hits [1]=kept [2]=cands [3]=dp [4]=max survivors of a query whose
                                 // segment overflowed [5]=extra work items [6]=segment space needed
                                 // [7]=queries queued for k_cand_big [8]=phase-B hits [9]=max kept
                                 // [10],[11]=queries queued for k_contrast<1>, <2> [12]=extra k_expand_b items
                                 // [21]=queue space needed (when qb is too small)
                                 // [13]=queries queued for k_cand_mid [14]=queries queued for the large k_finish
                                 // [15]=work items of k_expand_h [16]=hits walked through the bit planes
                                 // [17]=of those, hits that passed the plane test (re-tested exactly)
                                 // [18]=8-byte plane words read by k_expand_h [19],[20]=bytes / lines requested by k_probe<true>
                                 // [22]=hits of the light queries (walked by k_expand_b<true>) [23]=queries queued for k_cand
                                 // [24],[25]=segment / queue entries the pass used (always written: the host sizes later workspaces by them)
                                 // [26]=passes whose token bounds disagree with q_off (k_setup; the call fails)
                                 // [27]=sentence-pair edit distances of the flattened contrastive rerank (k_cround)
                                 // [28..31]=queries in k_cgrp_route's lists of 8 / 16 / 32 / 64 accepted matches
};
constexpr int kStatSlots = 64;     // k_expand_h's three statistics are summed into 64 slots each (kStatBase + 3 x 64): 8192 workgroups
                                   // adding to three single words at the end of the kernel queued for ~0.1 ms; k_stats folds them into [16..18]
constexpr int kStatBase = 32;
constexpr int kTailLists = 256;    // k_probe queues start positions for k_probe_tail in this many lists (a single fill counter would see
                                   // one returning atomic per wave, 400 k per step at C3: they serialise — measured 4 ms)
constexpr int kCtrProbeTail = kStatBase + 3 * kStatSlots;   // counters[kCtrProbeTail + l] = entries of list l
constexpr int kCtrDpDefer = kCtrProbeTail + kTailLists;   // (query, candidate) pairs with penalty tokens handed from k_dp_reg<.., REAL> to k_dp<true, *>
                                                          // ([28..31] are k_cgrp_route's four list lengths)
constexpr int kNumCounters = kCtrDpDefer + 8;

constexpr int kWave = 64;
constexpr uint32_t kLdsSortMax = 1024;   // per-wave LDS sort capacity (keys)
constexpr uint32_t kHeapLdsMax = 62;     // lowest_costs kept in LDS up to this contrast_buffer

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t n = __shfl_up(v, d, 64);
    if (lane >= (uint32_t)d) v += n;
  }
  return v;
}

__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ unsigned long long uni(unsigned long long x) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)x), hi = __builtin_amdgcn_readfirstlane((uint32_t)(x >> 32));
  return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_setup — src/fuzzy_match.cc:450-467
// ---------------------------------------------------------------------------------------------
__global__ void k_setup(DevIndex ix, DevBatch b, DevWork w, Params prm) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > b.nq) return;
  w.q_hits[q] = 0; w.q_hits_a[q] = 0;  // k_probe's per-query accumulators (nq + 1 entries)
  if (q == b.nq) {
    // the token bounds the host enqueued this pass with (the caller's total_tokens, when it gave one) against q_off itself
    if (b.q_off[b.q_begin] != b.t_begin || b.q_off[b.q_begin + b.nq] != b.t_begin + b.nt) atomicAdd(&w.counters[26], 1ull);
    return;
  }
  w.q_pb[q] = 0;
  const uint32_t o0 = b.q_off[b.q_begin + q], o1 = b.q_off[b.q_begin + q + 1];
  const uint32_t p = o1 - o0;
  // (a query whose tokens lie outside the bounds the pass was enqueued with — a wrong total_tokens — is skipped: nothing is
  // indexed out of range; the q == nq thread above makes the call fail)
  const bool inside = o0 >= b.t_begin && o1 >= o0 && o1 - b.t_begin <= b.nt;
  const uint32_t msl = inside ? min_seq_len(p, prm) : kNoMatchLen;
  w.q_m[q] = msl == kNoMatchLen ? kNoMatchLen : msl;
  // The token -> query map, coalesced (round 5): the wave's 64 queries own a contiguous stretch of tokens when they are all inside
  // the pass (they are, unless the call's offsets are wrong); lane l writes tokens l, l + 64, ... of the stretch and finds their
  // query by bisecting the 64 start offsets held across the lanes.  (One thread writing its query's tokens one by one: 64 separate
  // 4-byte stores per instruction, 70 us per million queries.)
  const unsigned long long act = __ballot(true);   // a prefix of the wave: lanes beyond the pass and the q == nq thread have left
  const uint32_t n_act = (uint32_t)__popcll(act), lane = lane_id();
  if (__ballot(inside) == act) {
    const uint32_t s0 = __shfl(o0, 0, 64), s1 = __shfl(o1, (int)n_act - 1, 64);   // (o1 of a lane is o0 of the next: one stretch)
    for (uint32_t base = s0; base < s1; base += n_act) {   // (wave-uniform trip count: every lane takes part in the shuffles)
      const uint32_t t = base + lane;
      // the LAST query that starts at or before t owns it (offsets ascend; an empty query shares its successor's offset)
      int lo = 0, hi = (int)n_act - 1;
#pragma unroll
      for (int step = 0; step < 6; step++) {
        const int mid = (lo + hi + 1) >> 1;
        const uint32_t om = __shfl(o0, mid, 64);
        if (om <= t) lo = mid; else hi = mid - 1;
      }
      if (t < s1) w.t_q[t - b.t_begin] = (q - lane) + (uint32_t)lo;
    }
  } else if (inside) {
    for (uint32_t k = 0; k < p; k++) w.t_q[o0 - b.t_begin + k] = q;
  }
}

// A pass whose token bounds disagree with q_off (k_setup counted it: the caller's total_tokens was wrong) is inert on the device:
// the per-token kernels below would otherwise walk b.nt tokens the caller's buffers may not hold, through t_q entries no query
// wrote.  With k_probe inert every query has zero hits and nothing downstream indexes a token; the host fails the call
// (FMX_E_INVALID) when it reads the counter.
__device__ __forceinline__ bool pass_refused(const DevWork& w) { return w.counters[26] != 0ull; }

// compute_idf_penalty — src/fuzzy_match.cc:381-387 (UNK and out-of-vocabulary ids -> 0), one thread per
// pattern token of the pass
__global__ void k_tidf(DevIndex ix, DevBatch b, DevWork w) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= b.nt || pass_refused(w)) return;
  const uint32_t wd = b.q_tok[b.t_begin + t];
  w.t_idf[t] = (wd + 1 < ix.n_qva && wd != 1u) ? ix.idf[wd] : 0.f;
}

// ---------------------------------------------------------------------------------------------
// k_probe — one lane per (query, start position): the suffix range the reference's narrowing
// equal_range calls converge to for the seed n-gram (src/suffix_array.cc:105-212,
// src/fuzzy_match.cc:495-551).  Each compare step is one 8-byte SufRec load + token loads.
// ---------------------------------------------------------------------------------------------
// A long suffix range handed to k_expand_h: `chunk` selects kHeavyChunkBlocks blocks of 64 suffixes of it.
struct HeavyItem {
  uint32_t t;      // token index of the (query, start position) inside the pass
  uint32_t lo;     // the range [lo, lo + cnt)
  uint32_t cnt;
  uint32_t chunk;
};
constexpr uint32_t kPlaneRows = kSigBits2;      // bit planes of the 112-bucket signature: rows 0..55 = first half, 56..111 = second half of each bucket
constexpr uint32_t kPlaneMaxP = 63;             // six-level bit-sliced counters: patterns of up to 63 tokens
constexpr uint32_t kHeavyMinDefault = 1024;     // ranges at least this long stream through the bit planes
constexpr uint32_t kLightMaxDefault = 256;      // queries with at most this many hits (all in short ranges) skip phase A: k_expand_b<true>
constexpr uint32_t kHeavyChunkBlocks = 256;     // blocks of 64 suffixes per k_expand_h work item (16384 hits)
constexpr uint32_t kHeavyGrid = 32768;  // (8192 until round 3: alone the kernel takes 1.05 instead of 1.22 ms with the finer distribution; no difference for six overlapped calls)

// segmented inclusive scan of (value per lane) over runs of equal `q` in a wave
__device__ __forceinline__ unsigned long long seg_scan_u64(unsigned long long v, uint32_t q) {
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t oq = __shfl_up(q, d, 64);
    const uint32_t olo = __shfl_up((uint32_t)v, d, 64), ohi = __shfl_up((uint32_t)(v >> 32), d, 64);
    if (lane >= (uint32_t)d && oq == q) v += ((unsigned long long)ohi << 32) | olo;
  }
  return v;
}

// COUNT (measurement runs): also tallies the bytes / lines the lane's loads request into counters[19], [20].
#ifndef FMX_PROBE_BLOCK
#define FMX_PROBE_BLOCK 64
#endif
constexpr uint32_t kProbeBlock = FMX_PROBE_BLOCK;  // threads per workgroup of k_probe (waves are independent: no LDS, no barrier)

// A start position whose range the head of the probe could not decide (fmx_core.hh: mgram_range_head): handed to k_probe_tail.
// The table range it is to search waits in t_lo / t_cnt of the start position.
struct ProbeItem {
  uint32_t t;        // token index of the (query, start position) inside the pass
  uint32_t from_m;   // `from` of mgram_range_head | seed length << 8
};
// list l holds the entries of the workgroups with blockIdx % kTailLists == l, kProbeBlock slots each: room for every token
__host__ __device__ inline uint32_t probe_tail_list_cap(uint32_t nt) {
  const uint32_t n_blocks = (nt + kProbeBlock - 1) / kProbeBlock;
  return kProbeBlock * ((n_blocks + kTailLists - 1) / kTailLists);
}

// what both probe kernels do with a lane's range: t_lo / t_cnt, the work items of the long ranges, the per-query hit totals
__device__ __forceinline__ void probe_emit(const DevBatch& b, const DevWork& w, uint32_t t, uint32_t q, uint32_t p, bool mine, uint32_t lo, uint32_t hi,
                                           uint32_t heavy_min, HeavyItem* items_h, uint32_t item_h_cap) {
  const uint32_t cnt = mine ? hi - lo : 0u;
  // Long ranges (a repeated phrase: 0.1 % of the ranges hold most of the hits) are not walked hit by hit: they become
  // work items of k_expand_h, which tests 64 suffixes per lane through the signature bit planes.
  const bool heavy = cnt >= heavy_min && p <= kPlaneMaxP;
  if (mine) {
    w.t_lo[t] = lo;
    w.t_cnt[t] = heavy ? 0u : cnt;
  }
  if (heavy) {
    const uint32_t blk0 = lo >> 6, blk1 = (hi - 1) >> 6;
    const uint32_t n_items = (blk1 - blk0) / kHeavyChunkBlocks + 1;
    const unsigned long long base = atomicAdd(&w.counters[15], (unsigned long long)n_items);
    for (uint32_t k = 0; k < n_items; k++)
      if (base + k < item_h_cap) { HeavyItem h; h.t = t; h.lo = lo; h.cnt = cnt; h.chunk = k; items_h[base + k] = h; }
  }
  // Per-query hit totals (q_hits / q_hits_a, zeroed by k_setup): a query's start positions are consecutive lanes, so a
  // segmented scan over the wave leaves one atomic per (wave, query) instead of one per start position.
  if (__ballot(cnt != 0) == 0ull) return;
  const uint32_t lane = lane_id();
  const unsigned long long sum = seg_scan_u64(cnt, q);
  const unsigned long long sum_a = seg_scan_u64(heavy ? 0u : cnt, q);
  const uint32_t nq = __shfl_down(q, 1, 64);
  const bool tail = lane == 63 || nq != q;
  if (tail && sum != 0) atomicAdd(&w.q_hits[q], sum);
  if (tail && sum_a != 0) atomicAdd(&w.q_hits_a[q], sum_a);
}

// SPLIT: the lane runs the head of the range function only — one line of the n-gram tables, which decides five starts out of six —
// and queues what is left for k_probe_tail (round 5: with everything in one kernel every wave ran the bisection loop, three to
// four dependent levels of key loads and compares, for the one lane in eight that needed it).
template <bool COUNT, bool SPLIT>
__global__ void __launch_bounds__(kProbeBlock) k_probe(DevIndex ix, DevBatch b, DevWork w, uint32_t heavy_min, HeavyItem* items_h,
                                              uint32_t item_h_cap, ProbeItem* tail_items) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (pass_refused(w)) return;
  const bool in = t < b.nt;
  const uint32_t q = in ? w.t_q[t] : 0xffffffffu;
  uint32_t lo = 0, hi = 0, p = 0, m = 0, from = kHeadDone;
  ProbeTally tally;
  tally.bytes = 16 + 12;  // t_q, q_m, two offsets; t_lo / t_cnt written, the hit totals
  tally.lines = 0;
  if (in) {
    const uint32_t msl = w.q_m[q];
    const uint32_t o0 = b.q_off[b.q_begin + q] - b.t_begin;
    p = b.q_off[b.q_begin + q + 1] - b.t_begin - o0;
    const uint32_t it = t - o0;
    if (msl != kNoMatchLen) {
      m = seed_len(p, msl);
      // register_suffix_range_match ignores match_length < _min_seq_len (src/ngram_matches.cc:65-66);
      // with p == 1 the only registered length is 1.
      const bool reachable = (p == 1) ? (msl <= 1) : true;
      if (reachable && it + m <= p) {
        const uint32_t* ngram = b.q_tok + b.t_begin + t;
        if constexpr (SPLIT) {
          uint32_t ng[kKeyTokens + 1];
          tally_add(COUNT ? &tally : nullptr, 4 * (m < kKeyTokens + 1 ? m : kKeyTokens + 1));  // the seed n-gram itself
#pragma unroll
          for (uint32_t k = 0; k <= kKeyTokens; k++) ng[k] = k < m ? ngram[k] : 0u;
          from = mgram_range_head(ix.gram, ix.qva, ix.n_qva, ng, m, lo, hi, COUNT ? &tally : nullptr);
        } else {
          mgram_range_hashed(ix.gram, ix.suf_key, ix.suf, ix.tok, ix.qva, ix.n_qva, ngram, m, lo, hi, COUNT ? &tally : nullptr);
        }
      }
    }
  }
  const bool defer = SPLIT && from != kHeadDone;
  if (COUNT) {
    if (defer) tally.bytes += 16;  // the list entry and the range left for the tail
    unsigned long long tb = in ? tally.bytes : 0u, tl = in ? tally.lines : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { tb += __shfl_xor(tb, d, 64); tl += __shfl_xor(tl, d, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&w.counters[19], tb); atomicAdd(&w.counters[20], tl); }
  }
  if constexpr (SPLIT) {
    const unsigned long long dm = __ballot(defer);
    if (dm) {
      const uint32_t lane = lane_id();
      unsigned long long base = 0;
      const uint32_t list = blockIdx.x % (uint32_t)kTailLists;
      if (lane == 0) base = atomicAdd(&w.counters[kCtrProbeTail + list], (unsigned long long)__popcll(dm));
      base = uni(base) + (unsigned long long)list * probe_tail_list_cap(b.nt);
      if (defer) {
        ProbeItem pi; pi.t = t; pi.from_m = from | (m << 8);
        tail_items[base + __popcll(dm & ((1ull << lane) - 1ull))] = pi;
        w.t_lo[t] = lo; w.t_cnt[t] = hi - lo;  // the range the tail is to search
      }
    }
  }
  probe_emit(b, w, t, q, p, in && !defer, lo, hi, heavy_min, items_h, item_h_cap);
}

// The tail of the probe: one lane per queued start position (grid-stride: the host does not know how many there are), full waves.
constexpr uint32_t kProbeTailPer = 32;  // workgroups per list (at most)
template <bool COUNT>
__global__ void __launch_bounds__(kWave) k_probe_tail(DevIndex ix, DevBatch b, DevWork w, uint32_t heavy_min, HeavyItem* items_h,
                                                      uint32_t item_h_cap, const ProbeItem* tail_items) {
  const uint32_t list = blockIdx.x % (uint32_t)kTailLists;
  const unsigned long long n = w.counters[kCtrProbeTail + list];
  tail_items += (unsigned long long)list * probe_tail_list_cap(b.nt);
  const uint32_t per = gridDim.x / (uint32_t)kTailLists;  // workgroups per list: the host shrinks the grid with the pass
  for (unsigned long long i0 = (unsigned long long)(blockIdx.x / kTailLists) * kWave; i0 < n; i0 += (unsigned long long)per * kWave) {
    const unsigned long long i = i0 + lane_id();
    const bool in = i < n;
    uint32_t t = 0, q = 0xffffffffu, p = 0, lo = 0, hi = 0;
    ProbeTally tally;
    tally.bytes = in ? 16 + 12 + 4 * (kKeyTokens + 1) : 0;  // list entry, the range, t_q and two offsets, the seed n-gram again
    tally.lines = 0;
    if (in) {
      const ProbeItem pi = tail_items[i];
      t = pi.t;
      const uint32_t from = pi.from_m & 0xffu, m = pi.from_m >> 8;
      q = w.t_q[t];
      p = b.q_off[b.q_begin + q + 1] - b.q_off[b.q_begin + q];
      const uint32_t* ngram = b.q_tok + b.t_begin + t;
      uint32_t ng[kKeyTokens + 1];
#pragma unroll
      for (uint32_t k = 0; k <= kKeyTokens; k++) ng[k] = k < m ? ngram[k] : 0u;
      const uint32_t r_lo = w.t_lo[t], r_hi = r_lo + w.t_cnt[t];
      mgram_range_tail(ix.gram, ix.suf_key, ix.suf, ix.tok, ix.qva, ix.n_qva, ngram, ng, m, from, r_lo, r_hi, lo, hi, COUNT ? &tally : nullptr);
    }
    if (COUNT) {
      unsigned long long tb = tally.bytes, tl = tally.lines;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { tb += __shfl_xor(tb, d, 64); tl += __shfl_xor(tl, d, 64); }
      if (lane_id() == 0) { atomicAdd(&w.counters[19], tb); atomicAdd(&w.counters[20], tl); }
    }
    probe_emit(b, w, t, q, p, in, lo, hi, heavy_min, items_h, item_h_cap);
  }
}

// Signature bit planes and length thermometer planes of the suffix array (index build, once per upload): one wave per
// block of 64 suffixes.  sent_sig2 = the two halves of each sentence's 112-bucket signature (k_dv_sent_sig).
__global__ void __launch_bounds__(kWave) k_build_planes(const unsigned long long* __restrict__ suf_sig, const SufRecB* __restrict__ suf_b,
                                                        const unsigned long long* __restrict__ sent_sig2, unsigned long long n_suf,
                                                        unsigned long long n_blk, unsigned long long* __restrict__ planes,
                                                        unsigned long long* __restrict__ thermo, uint32_t n_thermo) {
  const unsigned long long blk = blockIdx.x;
  const uint32_t lane = threadIdx.x & 63;
  const unsigned long long i = blk * 64 + lane;
  const unsigned long long pk = i < n_suf ? suf_sig[i] : 0ull;
#pragma unroll
  for (uint32_t half = 0; half < 2; half++) {
    const unsigned long long sg = i < n_suf ? sent_sig2[2ull * suf_b[i].sid + half] : 0ull;
    unsigned long long mine = 0;  // 64 x 64 bit transpose of the block: row k = bit k of the 64 words
#pragma unroll
    for (uint32_t k = 0; k < kSigBits; k++) {
      const unsigned long long m = __ballot((sg >> k) & 1ull);
      if (lane == k) mine = m;
    }
    if (lane < kSigBits) planes[(unsigned long long)(half * kSigBits + lane) * n_blk + blk] = mine;
  }
  const uint32_t len = i < n_suf ? sig_slen(pk) : 0u;
  for (uint32_t s0 = 1; s0 <= n_thermo; s0 += 64) {
    unsigned long long row = 0;
    for (uint32_t k = 0; k < 64 && s0 + k <= n_thermo; k++) {
      const unsigned long long m = __ballot(len >= s0 + k);
      if (lane == k) row = m;
    }
    if (s0 + lane <= n_thermo) thermo[(unsigned long long)(s0 + lane - 1) * n_blk + blk] = row;
  }
}

// ---------------------------------------------------------------------------------------------
// exclusive scans of per-query counts -> segment offsets
// ---------------------------------------------------------------------------------------------
#ifndef FMX_SCAN_THREADS
#define FMX_SCAN_THREADS 1024  // (round 5: 256 and 512 measured no different — the scans are not what a chained call waits for)
#endif
constexpr int kScanThreads = FMX_SCAN_THREADS;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;

__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long* total) {
  __shared__ unsigned long long s_wave[kScanThreads / 64];
  const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
  unsigned long long x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned long long n = __shfl_up(x, d, 64);
    if (lane >= (uint32_t)d) x += n;
  }
  if (lane == 63) s_wave[wv] = x;
  __syncthreads();
  if (wv == 0) {
    unsigned long long y = lane < kScanThreads / 64 ? s_wave[lane] : 0ull;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned long long n = __shfl_up(y, d, 64);
      if (lane >= (uint32_t)d) y += n;
    }
    if (lane < kScanThreads / 64) s_wave[lane] = y;
  }
  __syncthreads();
  const unsigned long long base = wv ? s_wave[wv - 1] : 0ull;
  *total = s_wave[kScanThreads / 64 - 1];
  __syncthreads();
  return base + x - v;
}

// Single-pass exclusive scan with decoupled look-back (Merrill & Garland 2016) of NARR arrays of n elements laid out
// `stride` apart: out[i] = sum of in[0 .. i), so with a zero last element out[n - 1] is the total.  One launch instead
// of the reduce / scan-of-sums / add triple (the pipeline runs five scans per pass; between the large kernels every
// launch is ~10 us of empty GPU).  Tiles are numbered by a ticket, so every predecessor a tile waits for is already
// running.  state[a * kLbMaxTiles + tile] = flag << 62 | value: 1 = the tile's own sum, 2 = the inclusive prefix up to
// and including the tile; the host zeroes state and ticket once per pass (one memset with the counters).
constexpr uint32_t kLbMaxTiles = 1024u * (1024u / kScanThreads);  // tiles per scan: sub-batches of up to 8 M queries
constexpr int kLbMaxArrays = 3;
constexpr unsigned long long kLbValMask = (1ull << 62) - 1ull;

// The tile's part of the look-back scan: given every thread's sums of its kScanItems consecutive elements (NARR arrays),
// returns in ex[] the exclusive prefix of the WHOLE array in front of the thread's first element.  Whole block, once per kernel.
template <int NARR>
__device__ __forceinline__ void scan_tile_lb(uint32_t tile, const unsigned long long (&sum)[NARR], unsigned long long (&ex)[NARR],
                                             unsigned long long* state) {
  __shared__ unsigned long long s_total[NARR], s_prefix[NARR];
#pragma unroll
  for (int a = 0; a < NARR; a++) {
    unsigned long long total;
    ex[a] = block_excl_scan(sum[a], &total);
    if (threadIdx.x == 0) s_total[a] = total;
  }
  __syncthreads();
  if (threadIdx.x < kWave) {  // the first wave publishes the tile's sums and looks back, 64 predecessors at a time
    const uint32_t lane = threadIdx.x;
#pragma unroll
    for (int a = 0; a < NARR; a++) {
      unsigned long long* st = state + (size_t)a * kLbMaxTiles;
      const unsigned long long total = s_total[a];
      unsigned long long run = 0;
      if (tile != 0) {
        if (lane == 0) __hip_atomic_store(&st[tile], (1ull << 62) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int hi = (int)tile - 1;  // nearest predecessor not yet accounted for
        for (;;) {
          const int t = hi - (int)lane;
          unsigned long long sv = 0;
          do {
            sv = t >= 0 ? __hip_atomic_load(&st[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (2ull << 62);
          } while (__ballot((sv >> 62) == 0ull) != 0ull);
          const unsigned long long pm = __ballot((sv >> 62) == 2ull);  // lanes that found an inclusive prefix (t < 0 counts as one of 0)
          const int first_p = pm ? __ffsll((long long)pm) - 1 : kWave;
          unsigned long long part = (int)lane <= first_p ? (sv & kLbValMask) : 0ull;
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) part += __shfl_xor(part, d, 64);
          run += part;
          if (pm) break;
          hi -= kWave;
        }
      }
      if (lane == 0) {
        __hip_atomic_store(&st[tile], (2ull << 62) | ((run + total) & kLbValMask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_prefix[a] = run;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int a = 0; a < NARR; a++) ex[a] += s_prefix[a];
}

__device__ __forceinline__ uint32_t scan_ticket_take(uint32_t* ticket) {
  __shared__ uint32_t s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  __syncthreads();
  return s_tile;
}

template <int NARR>
__global__ void __launch_bounds__(kScanThreads) k_scan_lb(const unsigned long long* in, unsigned long long* out, size_t stride,
                                                         uint32_t n, unsigned long long* state, uint32_t* ticket) {
  const uint32_t tile = scan_ticket_take(ticket);
  const uint32_t base = tile * kScanTile + threadIdx.x * kScanItems;
  unsigned long long v[NARR][kScanItems], sum[NARR], ex[NARR];
#pragma unroll
  for (int a = 0; a < NARR; a++) {
    sum[a] = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { v[a][k] = base + k < n ? in[a * stride + base + k] : 0ull; sum[a] += v[a][k]; }
  }
  scan_tile_lb<NARR>(tile, sum, ex, state);
#pragma unroll
  for (int a = 0; a < NARR; a++) {
    unsigned long long e = ex[a];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) { if (base + k < n) out[a * stride + base + k] = e; e += v[a][k]; }
  }
}

// ---------------------------------------------------------------------------------------------
// Work items of the expand stage: queries with more than kExpandChunk hits get several, so that several waves share
// the walk (k_caps_scan below lays them out).
// ---------------------------------------------------------------------------------------------
#ifndef FMX_EXPAND_CHUNK
#define FMX_EXPAND_CHUNK 16384
#endif
#ifndef FMX_UNROLL
#define FMX_UNROLL 4
#endif
constexpr uint32_t kExpandChunk = FMX_EXPAND_CHUNK;  // suffix hits per k_expand work item
constexpr uint32_t kExpandExtraGrid = 8192;

struct WorkItem {
  uint32_t q;
  uint32_t chunk;
};

// Appends work items to a list with ONE global atomic per block (same-address atomics retire one at a time: a million
// per-query atomics on one counter, or even one per wave, would dominate the kernel).  Whole block, blockDim.x a multiple of
// 64 and at most 1024; a thread holds kScanItems consecutive queries (q0 .. q0 + kScanItems - 1) with ni[k] work items each.
__device__ __forceinline__ void append_items_multi(unsigned long long* counter, WorkItem* items, uint32_t item_cap, uint32_t q0,
                                                   const uint32_t (&ni)[kScanItems], uint32_t first_chunk) {
  __shared__ uint32_t s_wave[16];
  __shared__ unsigned long long s_base;
  const uint32_t lane = lane_id(), wv = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) mine += ni[k];
  const uint32_t incl = wave_incl_scan(mine);
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (uint32_t k = 0; k < n_wv; k++) { const uint32_t c = s_wave[k]; s_wave[k] = tot; tot += c; }
    s_base = tot ? atomicAdd(counter, (unsigned long long)tot) : 0ull;
  }
  __syncthreads();
  unsigned long long at = s_base + s_wave[wv] + (incl - mine);
#pragma unroll
  for (int k = 0; k < kScanItems; k++)
    for (uint32_t j = 0; j < ni[k]; j++, at++)
      if (at < item_cap) { WorkItem it; it.q = q0 + (uint32_t)k; it.chunk = first_chunk + j; items[at] = it; }
}

// k_caps_scan — per query: queue capacity (input of the scan), "light" or not, and the work items of phase A; the scan behind
// it in the SAME launch (a pass is a chain of dependent launches, and beside another call's large kernels every one of them
// waits for wave slots): a thread computes the capacities of its kScanItems queries, the tile scans them (capacity -> q_qseg,
// light segment size -> q_segl, look-back over the tiles) and appends the work items.
// light_max != 0: queries whose hits all lie in short ranges, at most light_max of them, and whose pattern fits one wave
// are "light" — k_expand_b<true> walks their ranges itself (no phase A, no queue: two thirds of the hits of a short
// range pass the signature test anyway, so phase A was a per-query instruction stream for nothing) and their segments
// (q_capl -> q_segl) are sized by their hits.  Everything else goes through k_expand_a / k_expand_h as a work-item list
// (chunk 0 included).  light_max == 0: every query is walked by block q of k_expand_a, the list holds chunks >= 1.
__global__ void __launch_bounds__(kScanThreads) k_caps_scan(DevBatch b, DevWork w, WorkItem* items, uint32_t item_cap, uint32_t s_max,
                                                           uint32_t heavy_div, uint32_t light_max, unsigned long long* state,
                                                           uint32_t* ticket) {
  const uint32_t tile = scan_ticket_take(ticket);
  const uint32_t base = tile * kScanTile + threadIdx.x * kScanItems;
  unsigned long long v[2][kScanItems], sum[2] = {0ull, 0ull}, ex[2];
  uint32_t ni[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t q = base + k;
    v[0][k] = 0; v[1][k] = 0; ni[k] = 0;
    if (q == b.nq) w.q_capl[q] = 0;
    if (q < b.nq) {
      const unsigned long long all_hits = w.q_hits[q];  // summed by k_probe
      const unsigned long long hits = w.q_hits_a[q];    // the hits of the short ranges
      bool light = false;
      if (light_max != 0 && hits != 0 && hits == all_hits && hits <= light_max)
        light = b.q_off[b.q_begin + q + 1] - b.q_off[b.q_begin + q] <= (uint32_t)kWave;
      // Queue capacity of the query: every hit of the short ranges may survive phase A (near-duplicates: most
      // do), of the long ranges only a few per cent do — a heavy_div-th of them is reserved, and a query that needs more
      // is reported by k_seg_scan (the host re-runs the pass with a smaller divisor, down to 1 = every hit).
      const unsigned long long want = light ? 0ull : hits + (all_hits - hits + heavy_div - 1) / heavy_div;
      v[0][k] = want < s_max ? want : (unsigned long long)s_max;
      v[1][k] = light ? hits : 0ull;
      w.q_capl[q] = v[1][k];
      w.q_kept[q] = 0;
      if (!light && hits != 0) {
        const uint32_t n_chunks = (uint32_t)((hits + kExpandChunk - 1) / kExpandChunk);
        ni[k] = light_max != 0 ? n_chunks : n_chunks - 1;
      }
    }
    sum[0] += v[0][k]; sum[1] += v[1][k];
  }
  scan_tile_lb<2>(tile, sum, ex, state);
  unsigned long long e0 = ex[0], e1 = ex[1];
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t q = base + k;
    if (q <= b.nq) { w.q_qseg[q] = e0; w.q_segl[q] = e1; }
    e0 += v[0][k]; e1 += v[1][k];
  }
  append_items_multi(&w.counters[5], items, item_cap, base, ni, light_max != 0 ? 0u : 1u);
}

// PatternCoverage::count_covered_words (src/pattern_coverage.cc:15-28) with the pattern broadcast
// from LDS: each lane walks its own sentence once per 64 pattern positions and marks the positions
// whose word it sees; the marked count equals the reference's sum of per-word counts.
// A wave-uniform value loaded through the vector memory path, moved to scalar registers (frees VGPRs and
// lets address arithmetic on it run on the scalar unit).

// Ordering point for LDS traffic inside a single-wave workgroup: the LDS executes one wave's requests in
// program order, so only the compiler has to be kept from moving accesses across; no memory-counter wait and
// no s_barrier (outstanding global loads stay in flight, unlike __syncthreads()).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ uint32_t cover_lds(const uint32_t* __restrict__ sent, uint32_t slen,
                                              const uint32_t* s_pat, uint32_t p) {
  uint32_t cover = 0;
  for (uint32_t jb = 0; jb < p; jb += 64) {
    const uint32_t jn = p - jb < 64 ? p - jb : 64;
    uint32_t m0 = 0, m1 = 0;
    for (uint32_t k = 0; k < slen; k++) {
      const uint32_t t = sent[k];
      const uint32_t j0 = jn < 32 ? jn : 32;
      for (uint32_t j = 0; j < j0; j++) m0 |= (t == s_pat[jb + j]) ? (1u << j) : 0u;
      for (uint32_t j = 32; j < jn; j++) m1 |= (t == s_pat[jb + j]) ? (1u << (j - 32)) : 0u;
    }
    cover += __popc(m0) + __popc(m1);
  }
  return cover;
}

constexpr uint32_t kCminTab = 256;    // one entry per stored sentence length (8 bits, the last = "255 or longer")
constexpr uint32_t kGeLevels = 8;     // thresholds tmin .. tmin + 7 are told apart by the bit-sliced test (k_expand_h); higher ones count as tmin + 7
constexpr uint32_t kCminExt = 40;     // after the 256 thresholds: tmin, a spare, then per level two length intervals {first, last}
constexpr uint32_t kCminRow = kCminTab + kCminExt;
constexpr uint32_t kMaxPlanes = 8;    // bit planes of the per-bucket position counts (counts < 256)

// Per-work-item hash table of the pattern's words (p <= 64): word -> the pattern positions that hold it, as a bit mask.
// count_covered_words (src/pattern_coverage.cc:15-28: every distinct pattern word found in the sentence counts with the number of
// its positions) is then the popcount of the OR of the masks of the sentence's tokens: one LDS probe per token, nothing to weigh.
// (Until round 5 the table held a distinct-word index and the per-word counts went through seven ballot planes: 115 instructions
// per query to build, a weighted popcount per survivor.)
constexpr uint32_t kHashSlots = 128;
constexpr uint32_t kHashEmpty = 0xffffffffu;

struct PatHash {
  uint32_t key[kHashSlots];
  unsigned long long mask[kHashSlots];
};
// Patterns of up to 32 tokens (93 % of C3's): {word, mask} in one 8-byte slot, and a probe reads the home slot and its successor
// with ONE ds_read2_b64 and no loop — slot 128 mirrors slot 0, and a table in which some word sits further than one slot from
// home (pat_hash_build32 returns false: rare at a quarter full) is probed by the loop instead.
struct PatSlot { uint32_t key, mask; };
struct PatHash32 {
  PatSlot s[kHashSlots + 1];
};
union PatTable {
  PatHash h64;
  PatHash32 h32;
};

__device__ __forceinline__ uint32_t pat_hash_slot(uint32_t w) { return (w * 0x85EBCA6Bu) >> 25; }

// one wave, p <= 64
__device__ __forceinline__ void pat_hash_build(PatHash& ph, const uint32_t* s_pat, uint32_t p) {
  const uint32_t lane = lane_id();
  ph.key[lane] = kHashEmpty; ph.key[lane + 64] = kHashEmpty;
  ph.mask[lane] = 0ull; ph.mask[lane + 64] = 0ull;
  wave_lds_sync();
  if (lane < p) {
    const uint32_t wd = s_pat[lane];
    uint32_t h = pat_hash_slot(wd);
    for (;;) {
      const uint32_t old = atomicCAS(&ph.key[h], kHashEmpty, wd);
      if (old == kHashEmpty || old == wd) break;
      h = (h + 1) & (kHashSlots - 1);
    }
    atomicOr(&ph.mask[h], 1ull << lane);
  }
  wave_lds_sync();
}

// hash multipliers of the loop-free tables: a table with a word too far from its home slot is rebuilt with the next one
__device__ __forceinline__ uint32_t tab_mult(uint32_t k) {
  return k == 0 ? 0x85EBCA6Bu : (k == 1 ? 0xC2B2AE35u : (k == 2 ? 0x9E3779B1u : 0x27D4EB2Fu));
}
constexpr uint32_t kPatTries = 3;
// one wave, p <= 32.  Returns the multiplier under which every word sits in its home slot or the next one (a third of the tables
// of 25 distinct words fail that under ONE multiplier, 3 % under three), or 0 when none of kPatTries did.
__device__ __forceinline__ uint32_t pat_hash_build32(PatHash32& ph, const uint32_t* s_pat, uint32_t p) {
  const uint32_t lane = lane_id();
  for (uint32_t k = 0; k < kPatTries; k++) {
    const uint32_t mult = tab_mult(k);
    PatSlot e; e.key = kHashEmpty; e.mask = 0u;
    ph.s[lane] = e; ph.s[lane + 64] = e;
    wave_lds_sync();
    bool far = false;
    if (lane < p) {
      const uint32_t wd = s_pat[lane];
      uint32_t h = (wd * mult) >> 25, d = 0;
      for (;;) {
        const uint32_t old = atomicCAS(&ph.s[h].key, kHashEmpty, wd);
        if (old == kHashEmpty || old == wd) break;
        h = (h + 1) & (kHashSlots - 1); d++;
      }
      atomicOr(&ph.s[h].mask, 1u << lane);
      far = d > 1;
    }
    wave_lds_sync();
    if (__ballot(far) == 0ull) {
      if (lane == 0) ph.s[kHashSlots] = ph.s[0];
      wave_lds_sync();
      return mult;
    }
  }
  return 0u;
}

constexpr int kUnroll = FMX_UNROLL;  // hits per lane per phase-A iteration
constexpr uint32_t kQueue = 256;  // LDS ring of hits that passed the cheap tests (power of two >= 64 + 2 * 64)
#ifndef FMX_PHASE_B_CHUNK
#define FMX_PHASE_B_CHUNK 1024
#endif
constexpr uint32_t kPhaseBChunk = FMX_PHASE_B_CHUNK;  // queued hits per k_expand_b work item
// "no tabulated threshold": phase B evaluates the bound itself.  Table values are ordered so that phase A
// needs one compare per hit: thresholds (< 0x8000) <= kNeedExact <= the bound with the signature disabled
// (0x8000 | count) < 0xffff = rejected by the length filter.
constexpr uint32_t kNeedExact = 0x8000u;

// q_kept flag: phase B already reduced the query's survivors to candidates (k_cand has nothing to do)
constexpr uint32_t kKeptReduced = 0x80000000u;
__device__ __forceinline__ unsigned long long wave_sort64_upto(unsigned long long key, uint32_t n);

constexpr int kBUnroll = 2;  // queued hits per lane per phase-B block (their records are loaded together)
#ifndef FMX_LIGHT_WAVES
#define FMX_LIGHT_WAVES 7
#endif

// a queued hit that passed phase B's dominance test, waiting in LDS for the token work
struct __attribute__((aligned(16))) SurvB {
  uint32_t off;       // token offset of the suffix
  uint32_t rem_slen;  // tokens left | sentence length << 16
  uint32_t sid;
  uint32_t it_need;
};

struct __attribute__((aligned(8))) SigPair {  // two consecutive suf_sig words: one global_load_dwordx4
  unsigned long long a, b;
};

// ---------------------------------------------------------------------------------------------
// k_cmin — the phase-A threshold table.  It depends on the pattern length and the parameters only, so it is
// computed once per pass for every length (block = pattern length, thread = stored sentence length) instead
// of by every work item of k_expand_a (where the searches were a third of the kernel's instructions):
// 0xffff = rejected by the length filter (theoretical_rejection), 0 = clamped length, phase B decides,
// kNeedExact = no signature bound (negative costs), else the minimal cover that passes
// theoretical_rejection_cover.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_cmin(uint32_t max_slen, Params prm, int use_sig, uint16_t* tab) {
  __shared__ uint16_t s_c[kCminTab];
  const uint32_t p = blockIdx.x, sl = threadIdx.x;
  uint32_t c = 0xffffu;
  if (sl == kSigLenClamp) c = max_slen >= kSigLenClamp ? 0u : 0xffffu;
  else if (sl <= max_slen && !reject_len(p, sl, prm)) c = use_sig ? cover_threshold(p, sl, prm) : kNeedExact;
  uint16_t* row = tab + (size_t)p * kCminRow;
  row[sl] = (uint16_t)c;
  s_c[sl] = (uint16_t)c;
  __syncthreads();
  // Length intervals for the bit-sliced test of k_expand_h: tmin = the smallest threshold of any admissible stored
  // length, and for each level j (threshold tmin + j; j = 7 also stands for everything above) up to two intervals
  // [first, last] of stored lengths with that threshold.  Thresholds are constant below the pattern length and grow
  // above it, so two intervals per level are exact in practice; a third run is merged into the second (the lengths in
  // between are then tested against a threshold that is too low or should have been rejected by the length filter:
  // a superset, and every plane survivor is re-tested exactly).  Lengths the table rejects lie in no interval.
  if (threadIdx.x == 0) {
    uint16_t* ext = row + kCminTab;
    uint32_t tmin = 0xffffu;
    for (uint32_t l = 0; l < kSigLenClamp; l++) {
      const uint32_t v = s_c[l];
      if (v <= kPlaneMaxP && v < tmin) tmin = v;  // rejected (0xffff) and "no bound" (kNeedExact) are above kPlaneMaxP
    }
    ext[0] = (uint16_t)tmin;
    ext[1] = 0;
    uint32_t cnt[kGeLevels];
    for (uint32_t j = 0; j < kGeLevels; j++) {
      cnt[j] = 0;
      for (uint32_t r = 0; r < 2; r++) { ext[2 + 4 * j + 2 * r] = 1; ext[3 + 4 * j + 2 * r] = 0; }  // empty: first > last
    }
    for (uint32_t l = 0; l < kSigLenClamp && tmin != 0xffffu; l++) {
      const uint32_t v = s_c[l];
      if (v > kPlaneMaxP) continue;
      const uint32_t j = v - tmin < kGeLevels ? v - tmin : kGeLevels - 1;
      uint16_t* iv = ext + 2 + 4 * j;
      if (cnt[j] && iv[2 * (cnt[j] - 1) + 1] + 1u == l) iv[2 * (cnt[j] - 1) + 1] = (uint16_t)l;
      else if (cnt[j] < 2) { iv[2 * cnt[j]] = (uint16_t)l; iv[2 * cnt[j] + 1] = (uint16_t)l; cnt[j]++; }
      else iv[3] = (uint16_t)l;
    }
  }
}

// a hit that passed the cheap tests of phase A, queued for phase B
struct QHit {
  uint32_t i;        // suffix index
  uint32_t it_need;  // start position << 16 | minimal cover (or kNeedExact)
};

// ---------------------------------------------------------------------------------------------
// k_expand_a — phase A of the hit walk: one wave per work item (query, chunk of kExpandChunk hits).
// Walks the hits (the suffixes the reference's register_suffix_range_match visits,
// src/ngram_matches.cc:61-83) reading 8 bytes per hit — sentence length and 48-bit word signature —
// and decides most of them without touching a token:
//   s_cmin[s]   per work item, in LDS: minimal cover a sentence of s tokens needs to pass
//               theoretical_rejection_cover (0xffff = rejected by theoretical_rejection, the length
//               filter); monotone in the cover for non-negative costs
//   planes      bit planes of "pattern positions per signature bucket": an upper bound of the cover is
//               sum_k popcount(sig & plane_k) << k
// The ~2.5 % that pass are compacted through an LDS ring and appended, 64 at a time with one atomic, to
// the query's queue for k_expand_b.  The kernel is deliberately small (few registers, no token access):
// it is bound by HBM latency, so what matters is how many waves are resident.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kWave, 8) k_expand_a(DevIndex ix, DevBatch b, DevWork w, Params prm,
                                                    const WorkItem* items, uint32_t item_cap, int use_sig,
                                                    uint32_t first_q_blocks) {
  __shared__ __attribute__((aligned(8))) uint16_t s_cmin[kCminTab];  // per stored sentence length: minimal cover, 0xffff = rejected, 0 = phase B decides
  __shared__ uint32_t s_npref[kWave + 1];  // exclusive hit prefix of the non-empty start positions
  __shared__ uint32_t s_nlo[kWave];
  __shared__ uint32_t s_nit[kWave];
  __shared__ uint32_t s_bucket[kWave];
  __shared__ QHit s_q[kQueue];
  const uint32_t lane = lane_id();
  uint32_t q, chunk;
  uint32_t item_idx = 0;
  // blocks [0, first_q_blocks) walk chunk 0 of query blockIdx.x; the others loop over the work-item list
  const bool extra = blockIdx.x >= first_q_blocks;
  if (!extra) { q = blockIdx.x; chunk = 0; }
  else item_idx = blockIdx.x - first_q_blocks;
  for (;;) {
    if (extra) {
      unsigned long long n_items = w.counters[5];
      if (n_items > item_cap) n_items = item_cap;
      if (item_idx >= n_items) return;
      // wave-uniform by construction: keep everything derived from them in scalar registers
      q = __builtin_amdgcn_readfirstlane(items[item_idx].q);
      chunk = __builtin_amdgcn_readfirstlane(items[item_idx].chunk);
    }
    const uint32_t msl = uni(w.q_m[q]);
    const unsigned long long hits = msl == kNoMatchLen ? 0ull : uni(w.q_hits_a[q]);
    if (hits != 0) {
      const unsigned long long seg0 = uni(w.q_qseg[q]);
      const uint32_t cap = (uint32_t)(uni(w.q_qseg[q + 1]) - seg0);
      if (seg0 + cap > w.qb_cap) {  // queue space too small: host re-runs the sub-batch with a larger one
        if (lane == 0) atomicMax(&w.counters[21], w.q_qseg[b.nq]);
      } else {
        const uint32_t o0 = uni(b.q_off[b.q_begin + q]);
        const uint32_t p = uni(b.q_off[b.q_begin + q + 1]) - o0;
        const uint32_t* pat = b.q_tok + o0;
        const uint32_t t0 = o0 - b.t_begin;
        __syncthreads();
        // ---- per-item tables ----
        s_bucket[lane] = 0;
        __syncthreads();
        for (uint32_t j = lane; j < p; j += kWave) atomicAdd(&s_bucket[sig_bit(pat[j])], 1u);
        // this pattern length's row of the threshold table (k_cmin): 64 lanes x 8 bytes
        reinterpret_cast<unsigned long long*>(s_cmin)[lane] =
            reinterpret_cast<const unsigned long long*>(w.cmin_tab + (size_t)p * kCminRow)[lane];
        __syncthreads();
        unsigned long long planes[kMaxPlanes];
        uint32_t n_planes = 0;
        bool sig_off = !use_sig;
        {
          const uint32_t mycnt = s_bucket[lane];
          // a bucket with >= 2^kMaxPlanes pattern positions (p >= 256) would overflow the planes: the
          // signature bound is then replaced by its trivial value p (always passes)
          const bool overflow = __ballot(mycnt >> kMaxPlanes) != 0ull;
#pragma unroll
          for (uint32_t k = 0; k < kMaxPlanes; k++) {
            planes[k] = __ballot((mycnt >> k) & 1u);
            if (planes[k]) n_planes = k + 1;
          }
          if (overflow) { n_planes = 0; sig_off = true; }
          if (sig_off) {
#pragma unroll
            for (uint32_t k = 0; k < kMaxPlanes; k++) planes[k] = 0;
            n_planes = 0;
          }
        }
        const uint32_t ub_or = sig_off ? kNeedExact : 0u;  // signature bound disabled: every threshold passes, 0xffff still rejects
        QHit* qb = w.qb + seg0;
        const unsigned long long h_lo = (unsigned long long)chunk * kExpandChunk;
        const unsigned long long h_hi = h_lo + kExpandChunk < hits ? h_lo + kExpandChunk : hits;
        unsigned long long base_h = 0;  // hits of the start positions before c0
        uint32_t q_head = 0, q_tail = 0;  // ring of hits waiting to be flushed to the query's queue
        auto flush = [&](uint32_t n_take) {
          uint32_t base = 0;
          if (lane == 0) base = atomicAdd(&w.q_pb[q], n_take);
          base = __shfl(base, 0, 64);
          if (lane < n_take && base + lane < cap) qb[base + lane] = s_q[(q_head + lane) & (kQueue - 1)];
          q_head += n_take;
          __syncthreads();
        };
        for (uint32_t c0 = 0; c0 < p && base_h < h_hi; c0 += kWave) {
          const uint32_t it_l = c0 + lane;
          const uint32_t mycnt = it_l < p ? w.t_cnt[t0 + it_l] : 0u;
          const uint32_t incl = wave_incl_scan(mycnt);
          const uint32_t total = __shfl(incl, 63, 64);
          // compact the non-empty start positions: their prefixes are strictly increasing
          const unsigned long long nz = __ballot(mycnt != 0);
          const uint32_t n_nz = __popcll(nz);
          __syncthreads();
          if (mycnt != 0) {
            const uint32_t k = __popcll(nz & ((1ull << lane) - 1ull));
            s_npref[k] = incl - mycnt;
            s_nlo[k] = w.t_lo[t0 + it_l];
            s_nit[k] = it_l;
          }
          if (lane == 0) s_npref[n_nz] = total;
          __syncthreads();
          // hits [base_h, base_h + total) belong to this block of start positions; clip to the chunk
          const uint32_t a0 = h_lo > base_h ? (uint32_t)(h_lo - base_h) : 0u;  // both <= total (32 bits)
          const uint32_t a1 = h_hi - base_h < total ? (uint32_t)(h_hi - base_h) : total;
          uint32_t k = 0;  // this lane's current start position (index into the compacted list)
          // One hit, branch-free per lane: the tabulated threshold of the sentence length (0xffff = rejected
          // by the length filter, 0 = length outside the table: phase B evaluates everything) against the
          // upper bound of the cover from the word signature.  No token is touched for most hits.
          auto test = [&](bool valid, unsigned long long pk, uint32_t& cmin) -> bool {
            cmin = s_cmin[sig_slen(pk)];
            const unsigned long long sig = sig_bits(pk);
            uint32_t ub = (uint32_t)__popcll(sig & planes[0]) | ub_or;
            if (n_planes > 1) {
              ub += (uint32_t)__popcll(sig & planes[1]) << 1;
              if (n_planes > 2) {
#pragma unroll
                for (uint32_t kk2 = 2; kk2 < kMaxPlanes; kk2++)
                  if (kk2 < n_planes) ub += (uint32_t)__popcll(sig & planes[kk2]) << kk2;
              }
            }
            return valid & (ub >= cmin);  // cmin == 0xffff (length filter) exceeds every bound
          };
          // survivors of two hits per lane go to the LDS ring with one uniform branch
          auto push2 = [&](bool k0, uint32_t i0, uint32_t n0, bool k1, uint32_t i1, uint32_t n1) {
            const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
            if (m0 | m1) {
              const unsigned long long below = (1ull << lane) - 1ull;
              const uint32_t c0 = __popcll(m0);
              if (k0) { QHit e; e.i = i0; e.it_need = n0; s_q[(q_tail + __popcll(m0 & below)) & (kQueue - 1)] = e; }
              if (k1) { QHit e; e.i = i1; e.it_need = n1; s_q[(q_tail + c0 + __popcll(m1 & below)) & (kQueue - 1)] = e; }
              q_tail += c0 + __popcll(m1);
              if (q_tail - q_head >= kWave) {
                __syncthreads();
                do flush(kWave); while (q_tail - q_head >= kWave);
              }
            }
          };
          for (uint32_t hh = a0; hh < a1; hh += kWave * kUnroll) {
            // kUnroll hits per lane as kUnroll/2 pairs of consecutive hits: inside a range the pair is one
            // 16-byte load (8-byte loads reach only ~0.6 of the streaming rate), and all loads of the
            // iteration are in flight before the first is consumed
            const uint32_t blk_end = a1 - hh > kWave * kUnroll ? hh + kWave * kUnroll : a1;
            uint32_t k0 = __builtin_amdgcn_readfirstlane(k);
            while (s_npref[k0 + 1] <= (uint32_t)hh) k0++;
            if (s_npref[k0 + 1] >= blk_end) {
              // fast path (most hits: the long ranges): the whole block lies inside one start position's
              // range, so the suffix index is linear in the hit number and the start position is uniform
              k = k0;
              const uint32_t it_hi = s_nit[k0] << 16;
              const uint32_t i_base = s_nlo[k0] - s_npref[k0];
              if (blk_end - hh == kWave * kUnroll && n_planes <= 2) {
                // the common case inside a long range, stripped of every per-hit branch: full block, at most
                // two bit planes (no pattern word bucket holds four or more positions)
                const SigPair* src = reinterpret_cast<const SigPair*>(ix.suf_sig + (i_base + hh)) + lane;
                SigPair v[kUnroll / 2];
#pragma unroll
                for (int u = 0; u < kUnroll / 2; u++) v[u] = src[u * kWave];
                const uint32_t p0l = (uint32_t)planes[0], p0h = (uint32_t)(planes[0] >> 32);
                const uint32_t p1l = (uint32_t)planes[1], p1h = (uint32_t)(planes[1] >> 32);
                auto bound = [&](unsigned long long pk, uint32_t& cmin) -> bool {
                  cmin = s_cmin[sig_slen(pk)];
                  const uint32_t lo = (uint32_t)pk, hi = (uint32_t)(pk >> 32) & ((1u << (kSigBits - 32)) - 1u);
                  const uint32_t c0 = (uint32_t)__popc(lo & p0l) + (uint32_t)__popc(hi & p0h);
                  const uint32_t c1 = (uint32_t)__popc(lo & p1l) + (uint32_t)__popc(hi & p1h);
                  return ((c0 + (c1 << 1)) | ub_or) >= cmin;
                };
#pragma unroll
                for (int u = 0; u < kUnroll / 2; u++) {
                  const uint32_t i0 = i_base + hh + u * 2 * kWave + 2 * lane;
                  uint32_t c0, c1;
                  const bool keep0 = bound(v[u].a, c0);
                  const bool keep1 = bound(v[u].b, c1);
                  push2(keep0, i0, it_hi | c0, keep1, i0 + 1, it_hi | c1);
                }
                continue;
              }
              unsigned long long hpk[kUnroll];
#pragma unroll
              for (int u = 0; u < kUnroll / 2; u++) {
                const uint32_t h = (uint32_t)hh + u * 2 * kWave + 2 * lane;
                hpk[2 * u] = 0; hpk[2 * u + 1] = 0;
                if (h + 1 < blk_end) {
                  const SigPair v = *reinterpret_cast<const SigPair*>(ix.suf_sig + (i_base + h));
                  hpk[2 * u] = v.a; hpk[2 * u + 1] = v.b;
                } else if (h < blk_end) {
                  hpk[2 * u] = ix.suf_sig[i_base + h];
                }
              }
#pragma unroll
              for (int u = 0; u < kUnroll / 2; u++) {
                const uint32_t h = (uint32_t)hh + u * 2 * kWave + 2 * lane;
                uint32_t c0, c1;
                const bool keep0 = test(h < blk_end, hpk[2 * u], c0);
                const bool keep1 = test(h + 1 < blk_end, hpk[2 * u + 1], c1);
                push2(keep0, i_base + h, it_hi | c0, keep1, i_base + h + 1, it_hi | c1);
              }
              continue;
            }
            uint32_t hi_[kUnroll], hit_[kUnroll];
            bool hv[kUnroll], adj[kUnroll / 2];
            // a short block (the typical query: a dozen small ranges) only needs the first pair of hits per lane
            const bool two = blk_end - hh > 2 * kWave;
#pragma unroll
            for (int u = 0; u < kUnroll / 2; u++) {
              if (u >= 1 && !two) { hv[2 * u] = hv[2 * u + 1] = false; hi_[2 * u] = hi_[2 * u + 1] = 0; hit_[2 * u] = hit_[2 * u + 1] = 0; adj[u] = false; continue; }
              uint32_t kk[2] = {0, 0};
#pragma unroll
              for (int s2 = 0; s2 < 2; s2++) {
                const int e = 2 * u + s2;
                const uint32_t h = (uint32_t)hh + u * 2 * kWave + 2 * lane + s2;
                hv[e] = h < (uint32_t)a1;
                hi_[e] = 0; hit_[e] = 0;
                if (hv[e]) {
                  while (s_npref[k + 1] <= h) k++;  // amortised: each lane advances at most n_nz times per block
                  hit_[e] = s_nit[k];
                  hi_[e] = s_nlo[k] + (h - s_npref[k]);
                  kk[s2] = k;
                }
              }
              adj[u] = hv[2 * u + 1] && kk[0] == kk[1];
            }
            unsigned long long hpk[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll / 2; u++) {
              if (u >= 1 && !two) { hpk[2 * u] = hpk[2 * u + 1] = 0; continue; }
              if (adj[u]) {
                const SigPair v = *reinterpret_cast<const SigPair*>(ix.suf_sig + hi_[2 * u]);
                hpk[2 * u] = v.a; hpk[2 * u + 1] = v.b;
              } else {
                hpk[2 * u] = ix.suf_sig[hi_[2 * u]];
                hpk[2 * u + 1] = ix.suf_sig[hi_[2 * u + 1]];
              }
            }
#pragma unroll
            for (int u = 0; u < kUnroll / 2; u++) {
              if (u >= 1 && !two) continue;
              uint32_t c0, c1;
              const bool keep0 = test(hv[2 * u], hpk[2 * u], c0);
              const bool keep1 = test(hv[2 * u + 1], hpk[2 * u + 1], c1);
              push2(keep0, hi_[2 * u], (hit_[2 * u] << 16) | c0, keep1, hi_[2 * u + 1], (hit_[2 * u + 1] << 16) | c1);
            }
          }
          base_h += total;
        }
        if (q_tail != q_head) flush(q_tail - q_head);
      }
    }
    if (!extra) return;
    item_idx += gridDim.x - first_q_blocks;
  }
}

// ---------------------------------------------------------------------------------------------
// k_expand_h — phase A for the long suffix ranges.  98 % of all hits at C3 lie in the ~0.1 % of ranges with thousands
// of suffixes (a repeated phrase under a short pattern), and walking them hit by hit is an 8-byte-per-hit stream with
// ~20 VALU instructions per hit.  Here one LANE tests a whole block of 64 consecutive suffixes:
//   count   the signature bit planes (DevIndex::planes) give, per signature bucket, one 64-bit word = that bucket's bit
//           of the 64 suffixes, so the cover upper bound of all 64 is a bit-sliced sum (carry-save adders on 64-bit
//           words) of the planes of the pattern's buckets, weighted by positions per bucket
//   need    the threshold depends on the sentence length.  d = count - tmin is formed bit-sliced (tmin = the smallest
//           threshold of any admissible length, wave-uniform), decoded into the eight masks "d >= j", and each mask is
//           ANDed with the length intervals whose threshold is tmin + j — an interval [a, b] is T[a] & ~T[b + 1] on the
//           length thermometer planes (DevIndex::thermo): two loads, no compare
// Only the ~13-20 rows a pattern uses and ~10 thermometer rows are read: ~3 bytes per hit instead of 8, 64 lanes x 8 bytes
// contiguous per load, and ~5 VALU instructions per hit.  The survivors (a superset of phase A's, ~1 % of the hits) are
// re-tested exactly like k_expand_a does (same 8-byte word, same table), so the hits queued for k_expand_b are
// identical.  One wave per work item (range, 256 blocks).
// ---------------------------------------------------------------------------------------------
// carry-save adder on 32-bit words: s = x ^ y ^ z, c = majority(x, y, z).  The majority is a bit select — where x and y
// agree the carry is x, elsewhere it is z — i.e. one v_bfi_b32: three operations instead of five.  (The compiler rewrites
// the C expression of the select back into and / and / or, hence the asm; 32-bit halves because two asm results glued
// into an aligned 64-bit register pair cost copies and registers.)
__device__ __forceinline__ void csa32(uint32_t& s, uint32_t& c, uint32_t x, uint32_t y, uint32_t z) {
  const uint32_t u = x ^ y;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(c) : "v"(u), "v"(z), "v"(x));
  s = u ^ z;
}

// a[] = one 32-bit half of the six-level bit-sliced counters (bit i of a[l] = bit l of suffix i's count); adds eight
// one-bit planes with weight 2^J.  Carries beyond level 5 cannot occur (the total never exceeds the pattern length <= 63).
template <int J>
__device__ __forceinline__ void vadd8(uint32_t (&a)[6], const uint32_t (&x)[8]) {
  uint32_t s, c0, c1, c2, c3;
  csa32(s, c0, a[J], x[0], x[1]);
  csa32(s, c1, s, x[2], x[3]);
  csa32(s, c2, s, x[4], x[5]);
  csa32(s, c3, s, x[6], x[7]);
  a[J] = s;
  if constexpr (J + 1 < 6) {
    uint32_t d0, d1;
    csa32(s, d0, a[J + 1], c0, c1);
    csa32(s, d1, s, c2, c3);
    a[J + 1] = s;
    if constexpr (J + 2 < 6) {
      uint32_t e;
      csa32(s, e, a[J + 2], d0, d1);
      a[J + 2] = s;
#pragma unroll
      for (int l = J + 3; l < 6; l++) { const uint32_t t = a[l] & e; a[l] ^= e; e = t; }
    }
  }
}

// adds, with weight 2^J, the planes of the buckets in the (wave-uniform) mask m; col = this lane's column of the planes
// (lanes beyond the chunk point at its last block: no branch around the loads, the caller drops their result)
template <int J>
__device__ __forceinline__ void plane_level(uint32_t (&al)[6], uint32_t (&ah)[6], unsigned long long m, const unsigned long long* col,
                                            unsigned long long n_blk) {
  while (m) {
    unsigned long long x[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      x[u] = 0;
      if (m) {
        const uint32_t bk = (uint32_t)__builtin_ctzll(m);
        m &= m - 1ull;
        x[u] = col[(unsigned long long)bk * n_blk];
      }
    }
    // all eight loads are in flight before the first half is looked at
    uint32_t xl[8], xh[8];
#pragma unroll
    for (int u = 0; u < 8; u++) { xl[u] = (uint32_t)x[u]; xh[u] = (uint32_t)(x[u] >> 32); }
    vadd8<J>(al, xl);
    vadd8<J>(ah, xh);
  }
}

constexpr uint32_t kHeavyList = 2 * kWave;  // LDS list of plane-test survivors waiting for the exact test

// Registers of the two kernels below.  Until round 4 both ran at five waves per SIMD with ~270 / ~130 wave-uniform values spilled to
// VGPR lanes.  Most of those were not what the algorithm keeps per wave but what the compiler hoists: k_expand_h's bucket masks and
// sixteen length intervals are uniform for a whole work item, so every row offset and comparison derived from them was computed in
// front of the block loop and kept (94 VGPRs, 272 spilled SGPRs); k_expand_b's sorting network compares lane bits, so its 21
// direction masks were computed at the top of the kernel and kept for its length.  Now k_expand_h parks the masks and intervals in
// LDS and reads them (ds_read + readfirstlane) where a block iteration uses them — 58-60 VGPRs, which is what lets it run at EIGHT
// waves (0.95 -> 0.83 ms at C3; five waves with the old code: 1.05) — and the network takes the lane number through an opaque copy.
#ifndef FMX_HEAVY_WAVES
#define FMX_HEAVY_WAVES 8
#endif
#ifndef FMX_PHASEB_WAVES
#define FMX_PHASEB_WAVES 5
#endif
__global__ void __launch_bounds__(kWave, FMX_HEAVY_WAVES) k_expand_h(DevIndex ix, DevBatch b, DevWork w, Params prm, const HeavyItem* items,
                                                       uint32_t item_cap) {
  __shared__ __attribute__((aligned(8))) uint16_t s_cmin[kCminTab];
  __shared__ uint32_t s_bucket[2 * kWave];  // pattern positions per bucket of the 112-bucket signature: [0, 56) first halves, [64, 120) second halves
  __shared__ uint32_t s_list[kHeavyList];
  __shared__ QHit s_q[kQueue];
  // wave-uniform masks and intervals of the work item at hand, read where they are used (see above): positions per bucket as bit
  // planes over the buckets — s_pl = the 56 buckets of suf_sig (exact re-test), s_plh[h] = the two halves of each bucket (plane
  // test) — and the length intervals {first | last << 16} per level of this pattern length (k_cmin)
  __shared__ unsigned long long s_plh[2][6];
  __shared__ unsigned long long s_pl[6];
  __shared__ uint32_t s_iv[2 * kGeLevels];
  const uint32_t lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  unsigned long long n_items = w.counters[15];
  if (n_items > item_cap) n_items = item_cap;
  unsigned long long tot_hits = 0, tot_pass = 0, tot_words = 0;
  for (unsigned long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint32_t t = uni(items[item].t), r_lo = uni(items[item].lo), r_cnt = uni(items[item].cnt), chunk = uni(items[item].chunk);
    const uint32_t q = uni(w.t_q[t]);
    const unsigned long long seg0 = uni(w.q_qseg[q]);
    const uint32_t cap = (uint32_t)(uni(w.q_qseg[q + 1]) - seg0);
    if (seg0 + cap > w.qb_cap) continue;  // queue space too small: reported by k_seg_scan, the host re-runs the pass
    const uint32_t o0 = uni(b.q_off[b.q_begin + q]);
    const uint32_t p = uni(b.q_off[b.q_begin + q + 1]) - o0;
    const uint32_t* pat = b.q_tok + o0;
    const uint32_t it = t - (o0 - b.t_begin);
    const uint32_t it_hi = it << 16;
    wave_lds_sync();
    s_bucket[lane] = 0; s_bucket[kWave + lane] = 0;
    wave_lds_sync();
    for (uint32_t j = lane; j < p; j += kWave) { const uint32_t wd = pat[j]; atomicAdd(&s_bucket[sig_bit(wd) + kWave * sig_half(wd)], 1u); }
    reinterpret_cast<unsigned long long*>(s_cmin)[lane] =
        reinterpret_cast<const unsigned long long*>(w.cmin_tab + (size_t)p * kCminRow)[lane];
    wave_lds_sync();
    const uint32_t* ext = reinterpret_cast<const uint32_t*>(w.cmin_tab + (size_t)p * kCminRow + kCminTab);
    const uint32_t tmin = uni(ext[0]) & 0xffffu;
    const bool clamp_possible = s_cmin[kSigLenClamp] != 0xffffu;  // uniform: the index holds sentences of >= 255 tokens
    uint32_t stat_rows = clamp_possible ? 1u : 0u;  // rows read per block (statistics): the planes of the pattern's buckets ...
    {
      const uint32_t c0 = s_bucket[lane], c1 = s_bucket[kWave + lane];
#pragma unroll
      for (uint32_t k = 0; k < 6; k++) {
        const unsigned long long m0 = __ballot((c0 >> k) & 1u), m1 = __ballot((c1 >> k) & 1u), ms = __ballot(((c0 + c1) >> k) & 1u);
        if (lane == 0) { s_plh[0][k] = m0; s_plh[1][k] = m1; s_pl[k] = ms; }
        stat_rows += (uint32_t)__popcll(m0) + (uint32_t)__popcll(m1);
      }
    }
    if (lane < 2 * kGeLevels) s_iv[lane] = ext[1 + lane];
    wave_lds_sync();
    {  // ... and two thermometer rows per interval, one where it continues the previous
      uint32_t pe = 0xffffffffu;
#pragma unroll 1
      for (uint32_t k = 0; k < 2 * kGeLevels; k++) {
        const uint32_t ivk = s_iv[k], first = ivk & 0xffffu, last = ivk >> 16;
        if (first <= last) { stat_rows += first == pe ? 1u : 2u; pe = last + 1u; }
      }
    }
    QHit* qb = w.qb + seg0;
    uint32_t q_head = 0, q_tail = 0, n_list = 0;
    auto flush = [&](uint32_t n_take) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&w.q_pb[q], n_take);
      base = __shfl(base, 0, 64);
      if (lane < n_take && base + lane < cap) qb[base + lane] = s_q[(q_head + lane) & (kQueue - 1)];
      q_head += n_take;
      wave_lds_sync();
    };
    // exact phase-A test of one hit per lane (same word, same table, same bound as k_expand_a) and queueing
    auto exact = [&](bool valid, uint32_t i) {
      const unsigned long long pk = valid ? ix.suf_sig[i] : 0ull;
      const uint32_t cmin = s_cmin[sig_slen(pk)];
      const unsigned long long sig = sig_bits(pk);
      uint32_t ub = 0;
#pragma unroll
      for (uint32_t k = 0; k < 6; k++) ub += (uint32_t)__popcll(sig & s_pl[k]) << k;
      const bool keep = valid & (ub >= cmin);
      const unsigned long long m = __ballot(keep);
      if (m) {
        if (keep) { QHit e; e.i = i; e.it_need = it_hi | cmin; s_q[(q_tail + __popcll(m & below)) & (kQueue - 1)] = e; }
        q_tail += __popcll(m);
        if (q_tail - q_head >= kWave) { wave_lds_sync(); do flush(kWave); while (q_tail - q_head >= kWave); }
      }
    };
    const uint32_t r_hi = r_lo + r_cnt;  // exclusive
    const uint32_t blk_first = (r_lo >> 6) + chunk * kHeavyChunkBlocks;
    const uint32_t blk_last_r = (r_hi - 1) >> 6;
    const uint32_t blk_end = blk_first + kHeavyChunkBlocks - 1 < blk_last_r ? blk_first + kHeavyChunkBlocks : blk_last_r + 1;
    unsigned long long n_pass = 0;
    for (uint32_t b0 = blk_first; b0 < blk_end; b0 += kWave) {
      asm volatile("" ::: "memory");  // the reads of s_plh / s_iv below stay inside the iteration
      const uint32_t blk = b0 + lane;
      const bool valid = blk < blk_end;
      // lanes beyond the chunk load the chunk's last block instead of branching around every load; `surv` drops them
      const uint32_t blk_ld = valid ? blk : blk_end - 1u;
      // stored length >= s for the 64 suffixes of this lane's block (s wave-uniform)
      auto T = [&](uint32_t s_) -> unsigned long long {
        if (s_ == 0) return ~0ull;
        if (s_ > ix.n_thermo || !valid) return 0ull;
        return ix.thermo[(unsigned long long)(s_ - 1) * ix.n_blk + blk];
      };
      uint32_t al[6] = {0, 0, 0, 0, 0, 0}, ah[6] = {0, 0, 0, 0, 0, 0};
      const unsigned long long* col = ix.planes + blk_ld;
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const unsigned long long* ch = col + (unsigned long long)(h * kSigBits) * ix.n_blk;
        plane_level<0>(al, ah, uni(s_plh[h][0]), ch, ix.n_blk);
        plane_level<1>(al, ah, uni(s_plh[h][1]), ch, ix.n_blk);
        plane_level<2>(al, ah, uni(s_plh[h][2]), ch, ix.n_blk);
        plane_level<3>(al, ah, uni(s_plh[h][3]), ch, ix.n_blk);
        plane_level<4>(al, ah, uni(s_plh[h][4]), ch, ix.n_blk);
        plane_level<5>(al, ah, uni(s_plh[h][5]), ch, ix.n_blk);
      }
      unsigned long long a[6];
#pragma unroll
      for (int l = 0; l < 6; l++) a[l] = ((unsigned long long)ah[l] << 32) | al[l];
      // d = count - tmin, bit-sliced (tmin wave-uniform: each level is two operations); bor = borrow = count < tmin
      unsigned long long d[6], bor = 0;
#pragma unroll
      for (int l = 0; l < 6; l++) {
        if ((tmin >> l) & 1u) { d[l] = ~(a[l] ^ bor); bor = ~a[l] | bor; }
        else { d[l] = a[l] ^ bor; bor = ~a[l] & bor; }
      }
      const unsigned long long hi = d[3] | d[4] | d[5];
      unsigned long long ge[kGeLevels];  // ge[j]: count >= tmin + j
      ge[0] = ~0ull;
      ge[1] = hi | d[0] | d[1] | d[2];
      ge[2] = hi | d[1] | d[2];
      ge[3] = hi | d[2] | (d[1] & d[0]);
      ge[4] = hi | d[2];
      ge[5] = hi | (d[2] & (d[1] | d[0]));
      ge[6] = hi | (d[2] & d[1]);
      ge[7] = hi | (d[2] & d[1] & d[0]);
      unsigned long long pass = 0;
      // consecutive intervals usually touch (thresholds grow with the sentence length): the row that closed one opens the next
      uint32_t prev_end = 0xffffffffu;
      unsigned long long prev_t = 0;
#pragma unroll
      for (uint32_t k = 0; k < 2 * kGeLevels; k++) {
        const uint32_t ivk = uni(s_iv[k]);
        const uint32_t first = ivk & 0xffffu, last = ivk >> 16;
        if (first <= last) {  // last <= 254
          const unsigned long long tf = first == prev_end ? prev_t : T(first);
          const unsigned long long tl = T(last + 1u);
          pass |= tf & ~tl & ge[k >> 1];
          prev_end = last + 1u; prev_t = tl;
        }
      }
      pass &= ~bor;
      if (clamp_possible) pass |= T(kSigLenClamp);  // clamped lengths: phase B decides
      // the part of the block inside [r_lo, r_hi)
      const uint32_t lo_bit = blk == (r_lo >> 6) ? (r_lo & 63u) : 0u;
      const uint32_t hi_bit = blk == blk_last_r ? ((r_hi - 1) & 63u) + 1u : 64u;
      unsigned long long mask = hi_bit == 64 ? ~0ull : ((1ull << hi_bit) - 1ull);
      mask &= ~((1ull << lo_bit) - 1ull);
      unsigned long long surv = valid ? (pass & mask) : 0ull;
      n_pass += (unsigned long long)__popcll(surv);
      // survivors -> LDS list, 64 at a time through the exact test
      for (;;) {
        const bool has = surv != 0;
        const unsigned long long hm = __ballot(has);
        if (!hm) break;
        if (has) {
          const uint32_t bit = (uint32_t)__builtin_ctzll(surv);
          surv &= surv - 1ull;
          s_list[n_list + __popcll(hm & below)] = blk * 64u + bit;
        }
        n_list += __popcll(hm);
        if (n_list >= kWave) {
          wave_lds_sync();
          const uint32_t i = s_list[lane];
          const uint32_t rest = n_list - kWave;
          const uint32_t carry = lane < rest ? s_list[kWave + lane] : 0u;
          wave_lds_sync();
          if (lane < rest) s_list[lane] = carry;
          n_list = rest;
          exact(true, i);
        }
      }
    }
    if (n_list) {
      wave_lds_sync();
      const uint32_t i = lane < n_list ? s_list[lane] : 0u;
      exact(lane < n_list, i);
    }
    if (q_tail != q_head) { wave_lds_sync(); flush(q_tail - q_head); }
    // statistics, summed per workgroup (a shared atomic per work item would serialise the whole kernel on one address)
#pragma unroll
    for (int d2 = 32; d2 >= 1; d2 >>= 1) n_pass += __shfl_xor(n_pass, d2, 64);
    {
      const uint32_t h0 = blk_first == (r_lo >> 6) ? r_lo : blk_first * 64u;
      const uint32_t h1 = blk_end == blk_last_r + 1 ? r_hi : blk_end * 64u;
      tot_hits += h1 - h0;
      tot_pass += n_pass;
      tot_words += (unsigned long long)stat_rows * (blk_end - blk_first);
    }
  }
  if (lane == 0 && tot_hits) {
    const uint32_t slot = blockIdx.x & (kStatSlots - 1);
    atomicAdd(&w.counters[kStatBase + slot], tot_hits);
    atomicAdd(&w.counters[kStatBase + kStatSlots + slot], tot_pass);
    atomicAdd(&w.counters[kStatBase + 2 * kStatSlots + slot], tot_words);
  }
}

// a few microseconds of nothing: keeps the kernel behind it from reaching the GPU before one launched on another stream
__global__ void k_lag(const unsigned long long* p) {
  if (threadIdx.x == 0 && p[0] == ~0ull) __builtin_amdgcn_s_sleep(1);
}

// k_seg_scan — after phase A, in ONE launch (see k_caps_scan): the hits each query really queued become its segment size (the
// segments of everything downstream are sized by these, not by the loose bound the queues were sized by) and the work items
// of k_expand_b<false> per query (queries with more than kPhaseBChunk queued hits get several); both are scanned over the
// tiles; then segment base and capacity per query — the light queries' segments come first (q_segl, sized by their hits
// before phase A ran), the others follow in the order of this scan — and the work-item list, laid out by the same scan (no
// atomics, deterministic order).  q_seg[nq] = the segment space the pass needs.
__global__ void __launch_bounds__(kScanThreads) k_seg_scan(DevBatch b, DevWork w, WorkItem* items, uint32_t item_cap, uint32_t list0,
                                                          uint32_t first_chunk, unsigned long long* state, uint32_t* ticket) {
  const uint32_t tile = scan_ticket_take(ticket);
  const uint32_t base = tile * kScanTile + threadIdx.x * kScanItems;
  unsigned long long v[2][kScanItems], capl[kScanItems], sum[2] = {0ull, 0ull}, ex[2];
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t q = base + k;
    v[0][k] = 0; v[1][k] = 0; capl[k] = 0;
    if (q < b.nq) {
      capl[k] = w.q_capl[q];
      if (capl[k] != 0) {
        // light query: its segment lies in the light area (q_segl), k_expand_b<true> walks it
      } else if (w.q_qseg[q + 1] > w.qb_cap) {
        if (q + 1 == b.nq) atomicMax(&w.counters[21], w.q_qseg[b.nq]);  // host re-runs the pass with a larger queue space
      } else {
        const uint32_t cap = (uint32_t)(w.q_qseg[q + 1] - w.q_qseg[q]);
        const uint32_t n = w.q_pb[q];
        if (n > cap) atomicMax(&w.counters[4], (unsigned long long)n);  // queue overflow: retry with a larger capacity
        else {
          v[0][k] = n;
          if (n != 0) {
            const uint32_t n_chunks = (n + kPhaseBChunk - 1) / kPhaseBChunk;
            v[1][k] = list0 ? n_chunks : n_chunks - 1;
          }
        }
      }
    }
    sum[0] += v[0][k]; sum[1] += v[1][k];
  }
  scan_tile_lb<2>(tile, sum, ex, state);
  const unsigned long long totl = w.q_segl[b.nq];  // the light queries' segments come first
  unsigned long long e0 = ex[0], e1 = ex[1];
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t q = base + k;
    if (q == b.nq) {
      w.q_seg[q] = e0 + totl; w.counters[12] = e1;
      w.counters[24] = e0 + totl; w.counters[25] = w.q_qseg[q];
    } else if (q < b.nq) {
      const bool light = capl[k] != 0;
      w.q_scap[q] = light ? (uint32_t)capl[k] : (uint32_t)v[0][k];
      w.q_seg[q] = light ? w.q_segl[q] : totl + e0;
      for (unsigned long long i = e1; i < e1 + v[1][k] && i < item_cap; i++) { WorkItem it; it.q = q; it.chunk = first_chunk + (uint32_t)(i - e1); items[i] = it; }
    }
    e0 += v[0][k]; e1 += v[1][k];
  }
}

// ---------------------------------------------------------------------------------------------
// k_expand_b — phase B: one wave per (query, chunk of kPhaseBChunk queued hits), one hit per lane:
// dominance test (the previous start position reaches the same sentence with a longer match), match
// length, exact coverage test through a per-item hash set of the pattern's words, then ballot/prefix
// append of the survivors to the query's segment.  Dynamic LDS: pat[p_max].
// ---------------------------------------------------------------------------------------------
// LIGHT: the query's hits are not read from a queue but generated from its (short) suffix ranges — block q walks light
// query q (k_caps_scan); the length filter of phase A (the tabulated threshold of the sentence length) is applied here, the
// signature bound is skipped (the exact coverage test decides).  !LIGHT: blocks [0, first_q_blocks) take chunk 0 of
// query blockIdx.x, the others loop over the work-item list.
// (Round 4 measured the alternative the round-3 review proposed — leave the exact coverage test of patterns of up to 64 tokens to
// the bit-parallel edit-distance kernel, where count_covered_words is the popcount of the OR of the rows' match masks, and mark the
// candidates it rejects so that the replay skips them: bit-identical, the light kernel 1.41 -> 1.08 ms and k_expand_b<false>
// 0.82 -> 0.57 at C3, but the exact test removes a THIRD of the survivors (kept 8.3 M -> 11.9 M, candidates 6.7 M -> 10.1 M,
// queries with more than 64 survivors 41 -> 339 per step), and what follows pays more than was saved: candidate kernels
// 0.15 -> 0.48 ms, edit distances 0.58 -> 0.87, replay 0.46 -> 0.60 — one call at a time 179 -> 163 M patterns/s, six in flight
// 203 -> 196 M.  Removed; profiles/r04_c3_defer_cover_*.json keep the two lines.)
#ifdef FMX_LIGHT_STATS
__device__ unsigned long long g_light_stats[1024];  // design-time histograms of the light queries (tools/dbg/light_stats.py)
#endif
template <bool LIGHT>
__global__ void __launch_bounds__(kWave, LIGHT ? FMX_LIGHT_WAVES : FMX_PHASEB_WAVES) k_expand_b(DevIndex ix, DevBatch b, DevWork w, Params prm,
                                                       const WorkItem* items, uint32_t item_cap, uint32_t first_q_blocks,
                                                       uint32_t flat_max) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* s_pat = (uint32_t*)smem;
  __shared__ PatTable pt;
  PatHash& ph = pt.h64;
  // LIGHT: one block of 64 hits at a time (most light queries have no more), which keeps the kernel within 64 VGPRs and
  // 5 KB of LDS: eight waves per SIMD hide the two dependent loads in front of the token work
  constexpr int kBU = LIGHT ? 1 : kBUnroll;
  __shared__ SurvB s_surv[LIGHT ? kWave + 16 : kWave * (kBU + 1)];  // a carried rest of fewer than 16 + one block
  __shared__ unsigned long long s_out[kWave];  // survivors (sentence id << 16 | match length) while they fit one wave
  __shared__ uint32_t s_tb[kWave];             // ... where their sentence's tokens start (index into tok) and
  __shared__ uint16_t s_sl[kWave];             // ... its length: handed to the edit-distance kernels with the candidate
  // LIGHT: the non-empty start positions (exclusive hit prefix, range start, start position) and this pattern length's row
  // of the phase-A threshold table
  __shared__ uint32_t s_npref[LIGHT ? kWave : 1];
  __shared__ uint32_t s_nlo[LIGHT ? kWave : 1];
  __shared__ uint32_t s_nit[LIGHT ? kWave : 1];
  __shared__ __attribute__((aligned(8))) uint16_t s_cmin[LIGHT ? kCminTab : 4];
  const uint32_t lane0 = lane_id();
  uint32_t q, chunk;
  uint32_t item_idx = 0;
  const bool extra = !LIGHT && blockIdx.x >= first_q_blocks;
  if (!extra) { q = blockIdx.x + (LIGHT ? item_cap : 0u); chunk = 0; }  // (LIGHT: item_cap = first query of this launch's part)
  else item_idx = blockIdx.x - first_q_blocks;
  for (;;) {
    const uint32_t lane = lane0;
    if (extra) {
      unsigned long long n_items = w.counters[12];
      if (n_items > item_cap) n_items = item_cap;
      if (item_idx >= n_items) return;
      // wave-uniform by construction: keep everything derived from them in scalar registers
      q = __builtin_amdgcn_readfirstlane(items[item_idx].q);
      chunk = __builtin_amdgcn_readfirstlane(items[item_idx].chunk);
    }
    // The kernel is a chain of dependent memory latencies per query (most queries queue a few dozen hits),
    // so everything that can be requested together is: level 1 = all per-query scalars.
    const unsigned long long seg0 = uni(LIGHT ? w.q_segl[q] : w.q_seg[q]);
    const unsigned long long qseg0 = LIGHT ? 0ull : uni(w.q_qseg[q]);
    const uint32_t cap = LIGHT ? (uint32_t)(uni(w.q_segl[q + 1]) - seg0) : uni(w.q_scap[q]);
    const uint32_t n_q = LIGHT ? cap : uni(w.q_pb[q]);  // LIGHT: every hit of the query's ranges
    const uint32_t msl = uni(w.q_m[q]);
    const uint32_t o0 = uni(b.q_off[b.q_begin + q]);
    const uint32_t p = uni(b.q_off[b.q_begin + q + 1]) - o0;
    // LIGHT: queries of at most flat_max hits are walked by k_expand_lf (fmx_light.hip.hh), several to a wave
    const bool elsewhere = LIGHT && n_q <= flat_max;
    if (elsewhere) {
    } else if (n_q != 0 && seg0 + cap > w.seg_cap) {  // segments too small: host re-runs the pass
      if (lane == 0) atomicMax(&w.counters[6], LIGHT ? w.q_segl[b.nq] : w.q_seg[b.nq]);
    } else if (n_q != 0 && n_q <= cap) {
      const uint32_t m = seed_len(p, msl);
      const uint32_t* pat = b.q_tok + o0;
      const QHit* qb = w.qb + qseg0;
      Cand* seg = w.seg + seg0;
      const uint32_t e_lo = chunk * kPhaseBChunk;
      const uint32_t e_hi = e_lo + kPhaseBChunk < n_q ? e_lo + kPhaseBChunk : n_q;
      const bool sole = LIGHT || n_q <= kPhaseBChunk;  // this wave is the query's only phase-B worker: no atomics needed (a light query has at most light_max <= kPhaseBChunk hits)
      bool in_lds = sole;                     // ... and up to 64 survivors stay in LDS and are reduced right here
      uint32_t n_out = 0;
      QHit hq[kBU];
      SufRecB rb[kBU];
      bool v[kBU];
      if constexpr (LIGHT) {
        // the query's non-empty start positions, compacted: hit e belongs to the last one whose prefix is <= e
        const uint32_t t0 = o0 - b.t_begin;
        const uint32_t mycnt = lane < p ? w.t_cnt[t0 + lane] : 0u;  // p <= 64
        const uint32_t mylo = lane < p ? w.t_lo[t0 + lane] : 0u;
        reinterpret_cast<unsigned long long*>(s_cmin)[lane] =
            reinterpret_cast<const unsigned long long*>(w.cmin_tab + (size_t)p * kCminRow)[lane];
        const uint32_t incl = wave_incl_scan(mycnt);
        const unsigned long long nz = __ballot(mycnt != 0);
        if (mycnt != 0) {
          const uint32_t k = __popcll(nz & ((1ull << lane) - 1ull));
          s_npref[k] = incl - mycnt;
          s_nlo[k] = mylo;
          s_nit[k] = lane;
        }
        if (lane >= (uint32_t)__popcll(nz)) s_npref[lane] = 0xffffffffu;  // sentinels: the search below needs no bound test
        wave_lds_sync();
      }
      auto load_hits = [&](uint32_t e0) {
#pragma unroll
        for (int u = 0; u < kBU; u++) {
          const uint32_t e = e0 + u * kWave + lane;
          v[u] = e < e_hi;
          hq[u].i = 0; hq[u].it_need = 0;
          if constexpr (LIGHT) {
            if (v[u]) {
              uint32_t k = 0;
#pragma unroll
              for (uint32_t step = kWave / 2; step >= 1; step >>= 1)
                if (s_npref[k + step] <= e) k += step;  // k + step <= 63; entries past the last start hold the sentinel
              hq[u].i = s_nlo[k] + (e - s_npref[k]);
              hq[u].it_need = s_nit[k] << 16;
            }
          } else {
            if (v[u]) hq[u] = qb[e];
          }
        }
      };
      auto load_recs = [&]() {
#pragma unroll
        for (int u = 0; u < kBU; u++) {
          rb[u].off = 0; rb[u].rem = 0; rb[u].slen = 0; rb[u].prev = 0; rb[u].sid = 0;
          if (v[u]) rb[u] = ix.suf_b[hq[u].i];
        }
      };
      // level 2: the first block of queued hits together with the pattern; level 3: their 16-byte records,
      // in flight while the pattern's hash set is built (wave-level LDS ordering, no memory waits)
      load_hits(e_lo);
      wave_lds_sync();
      for (uint32_t j = lane; j < p; j += kWave) s_pat[j] = pat[j];
      load_recs();
      wave_lds_sync();
      uint32_t mult32 = 0;  // != 0: the pattern's table is the 32-bit one under this multiplier and needs no probe loop
      if (p <= 32) mult32 = uni(pat_hash_build32(pt.h32, s_pat, p));
      const bool near32 = mult32 != 0u;
      if (p <= kWave && !near32) pat_hash_build(ph, s_pat, p);  // (also: a 32-bit table with a word far from home under every multiplier)
      // Two stages per block of kWave * kBU queued hits.  Stage 1: the dominance test (about two thirds
      // of the queued hits fail it) compacts the survivors into LDS.  Stage 2: full waves of survivors do
      // the token work (match length + exact coverage), not diluted by dominated hits.
      uint32_t n_s = 0;
#ifdef FMX_LIGHT_STATS
      uint32_t ns_tot = 0;
#endif
      // survivors of the token work go to the LDS set (while they fit one wave) or to the query's segment
      auto append = [&](bool keep, uint32_t sid, uint32_t L, uint32_t tb, uint32_t sl) {
        const unsigned long long mask = __ballot(keep);
        if (mask) {
          const uint32_t cnt = __popcll(mask);
          const uint32_t rank = __popcll(mask & ((1ull << lane) - 1ull));
          if (in_lds && n_out + cnt > kWave) {  // does not fit one wave any more: continue in the segment
            if (lane < n_out) {
              Cand c;
              c.sid = (uint32_t)(s_out[lane] >> 16); c.longest = (uint32_t)(s_out[lane] & 0xffffu); c.score = 0.f; c.penalty = 0.f;
              seg[lane] = c;
            }
            in_lds = false;
          }
          if (in_lds) {
            if (keep) { s_out[n_out + rank] = ((unsigned long long)sid << 16) | (L & 0xffffu); s_tb[n_out + rank] = tb; s_sl[n_out + rank] = (uint16_t)sl; }
          } else {
            uint32_t base = n_out;
            if (!sole) {
              if (lane == 0) base = atomicAdd(&w.q_kept[q], cnt);
              base = __shfl(base, 0, 64);
            }
            if (keep && base + rank < cap) {
              Cand c;
              c.sid = sid; c.longest = L; c.score = 0.f; c.penalty = 0.f;
              seg[base + rank] = c;
            }
          }
          n_out += cnt;
        }
      };
      // patterns longer than one wave: one lane per survivor, pattern compared from LDS
      auto heavy = [&](uint32_t j0, bool active) {
        bool keep = active;
        uint32_t sid = 0, L = 0;
        if (keep) {
          const SurvB sv = s_surv[j0 + lane];
          const uint32_t it = sv.it_need >> 16, need = sv.it_need & 0xffffu;
          SufRec r;
          r.off = sv.off; r.rem = (uint16_t)(sv.rem_slen & 0xffffu); r.slen = (uint16_t)(sv.rem_slen >> 16);
          L = lcp_extend(ix.tok, r, s_pat + it, m, p - it);
          const uint32_t sl = r.slen;
          const uint32_t* sent = ix.tok + (r.off - (uint32_t)(r.slen - r.rem));
          const uint32_t cover = L < p ? cover_lds(sent, sl, s_pat, p) : p;
          if (need != kNeedExact && need != 0) keep = cover >= need;
          // need == 0: sentence length outside phase A's table, the length filter is still to be applied
          else keep = !reject_len(p, sl, prm) && !reject_cover(p, sl, cover, prm);
          sid = sv.sid;
        }
        const uint32_t sl_ = keep ? (s_surv[j0 + lane].rem_slen >> 16) : 0u;
        append(keep, sid, L, keep ? s_surv[j0 + lane].off - (sl_ - (s_surv[j0 + lane].rem_slen & 0xffffu)) : 0u, sl_);
      };
      // patterns of up to 64 tokens (hash set built): kSubLanes lanes per survivor share the token work, so
      // a typical query's dozen survivors fill the wave instead of a fifth of it.  The lanes of a group
      // compare / probe interleaved tokens; the match length is the first mismatch any of them sees, the
      // coverage the weighted popcount of the union of their "distinct pattern words seen" masks.
      constexpr uint32_t kSubLanes = 4, kSubSurv = kWave / kSubLanes;
      auto heavy4_t = [&](uint32_t j0, uint32_t n_act, auto tag) __attribute__((always_inline)) {
        constexpr int kTab = decltype(tag)::value;  // 0: 64-bit masks, probe loop; 2: 32-bit table, two slots, no loop
        const uint32_t sub = lane / kSubLanes, part = lane % kSubLanes;
        const bool act = sub < n_act;
        SurvB sv;
        sv.off = 0; sv.rem_slen = 0; sv.sid = 0; sv.it_need = 0;
        if (act) sv = s_surv[j0 + sub];
        const uint32_t it = sv.it_need >> 16, need = sv.it_need & 0xffffu;
        const uint32_t rem = sv.rem_slen & 0xffffu, sl = sv.rem_slen >> 16;
        // Everything the group needs from the sentence is requested at once — the first kLcpAhead steps of the match-length
        // walk (kSubLanes tokens per step) and the first kCovAhead x kSubLanes tokens of the coverage scan — so that a
        // typical survivor costs one memory latency here instead of one per step.
        constexpr int kLcpAhead = 4, kCovAhead = 8;
        const uint32_t lim = act ? (rem < p - it ? rem : p - it) : 0u;
        const uint32_t* sent = ix.tok + (sv.off - (sl - rem));
        uint32_t lt[kLcpAhead], ct[kCovAhead];
#pragma unroll
        for (int u = 0; u < kLcpAhead; u++) { const uint32_t k = m + u * kSubLanes + part; lt[u] = k < lim ? ix.tok[sv.off + k] : 0u; }
#pragma unroll
        for (int u = 0; u < kCovAhead; u++) { const uint32_t k = part + u * kSubLanes; ct[u] = (act && k < sl) ? sent[k] : kHashEmpty; }
        // match length: kSubLanes tokens per step, a position at or beyond the limit counts as a mismatch
        uint32_t L = m < lim ? m : lim;
        bool done = !act || m >= lim;
        uint32_t k0 = m;
#pragma unroll
        for (int u = 0; u < kLcpAhead; u++) {
          const uint32_t k = k0 + part;
          const bool mism = !done && (k >= lim || lt[u] != s_pat[it + k]);
          const uint32_t g = (uint32_t)(__ballot(mism) >> (sub * kSubLanes)) & ((1u << kSubLanes) - 1u);
          if (!done) {
            if (g) { L = k0 + (uint32_t)__ffs((int)g) - 1u; done = true; }
            else k0 += kSubLanes;
          }
        }
        while (__ballot(!done)) {  // longer matches: step by step
          const uint32_t k = k0 + part;
          const bool mism = !done && (k >= lim || ix.tok[sv.off + k] != s_pat[it + k]);
          const uint32_t g = (uint32_t)(__ballot(mism) >> (sub * kSubLanes)) & ((1u << kSubLanes) - 1u);
          if (!done) {
            if (g) { L = k0 + (uint32_t)__ffs((int)g) - 1u; done = true; }
            else k0 += kSubLanes;
          }
        }
        // coverage: pattern positions whose word is present in the sentence
        uint32_t cover = 0;
        if constexpr (kTab == 0) {
          unsigned long long seen = 0;
          if (act && L < p) {
            auto probe = [&](uint32_t t) {
              uint32_t h = pat_hash_slot(t);
              uint32_t key = ph.key[h];
              while (key != t && key != kHashEmpty) { h = (h + 1) & (kHashSlots - 1); key = ph.key[h]; }
              if (key == t) seen |= ph.mask[h];
            };
#pragma unroll
            for (int u = 0; u < kCovAhead; u++)
              if (part + u * kSubLanes < sl) probe(ct[u]);
            for (uint32_t kb = part + kCovAhead * kSubLanes; kb < sl; kb += 4 * kSubLanes) {  // sentences of more than 32 tokens
              uint32_t t[4];
#pragma unroll
              for (int u = 0; u < 4; u++) t[u] = kb + u * kSubLanes < sl ? sent[kb + u * kSubLanes] : kHashEmpty;
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if (kb + u * kSubLanes >= sl) break;
                probe(t[u]);
              }
            }
          }
#pragma unroll
          for (uint32_t d = 1; d < kSubLanes; d <<= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)seen, d, 64), hi = __shfl_xor((uint32_t)(seen >> 32), d, 64);
            seen |= ((unsigned long long)hi << 32) | lo;
          }
          cover = (uint32_t)__popcll(seen);
        } else {
          const PatHash32& p32 = pt.h32;
          uint32_t seen = 0;
          auto probe = [&](uint32_t t) {
            const uint32_t h = (t * mult32) >> 25;
            // padding tokens (kHashEmpty) meet an empty slot's zero mask or nothing: no guard needed
            const PatSlot a = p32.s[h], b2 = p32.s[h + 1];
            seen |= (a.key == t ? a.mask : 0u) | (b2.key == t ? b2.mask : 0u);
          };
#pragma unroll
          for (int u = 0; u < kCovAhead; u++) probe(ct[u]);
          if (__ballot(act && sl > kCovAhead * kSubLanes)) {  // sentences of more than 32 tokens
            for (uint32_t kb = part + kCovAhead * kSubLanes; kb < sl; kb += 4 * kSubLanes) {
              uint32_t t[4];
#pragma unroll
              for (int u = 0; u < 4; u++) t[u] = kb + u * kSubLanes < sl ? sent[kb + u * kSubLanes] : kHashEmpty;
#pragma unroll
              for (int u = 0; u < 4; u++) probe(t[u]);
            }
          }
#pragma unroll
          for (uint32_t d = 1; d < kSubLanes; d <<= 1) seen |= __shfl_xor(seen, d, 64);
          cover = (uint32_t)__popc(seen);
        }
        if (L >= p) cover = p;
        bool keep = act && part == 0;
        if (keep) {
          if (need != kNeedExact && need != 0) keep = cover >= need;
          // need == 0: sentence length outside phase A's table, the length filter is still to be applied
          else keep = !reject_len(p, sl, prm) && !reject_cover(p, sl, cover, prm);
        }
        append(keep, sv.sid, L, sv.off - (sl - rem), sl);
      };
      auto heavy4 = [&](uint32_t j0, uint32_t n_act) __attribute__((always_inline)) {
        if (near32) heavy4_t(j0, n_act, std::integral_constant<int, 2>{});
        else heavy4_t(j0, n_act, std::integral_constant<int, 0>{});
      };
      const uint32_t pass = p <= kWave ? kSubSurv : kWave;  // survivors per pass of the token work
      for (uint32_t e0 = e_lo; e0 < e_hi;) {
#pragma unroll
        for (int u = 0; u < kBU; u++) {
          const uint32_t it = hq[u].it_need >> 16;
          const bool dominated = it > 0 && rb[u].rem < rb[u].slen && rb[u].prev == s_pat[it ? it - 1 : 0];
          bool keep = v[u] && !dominated;
          uint32_t it_need = hq[u].it_need;
          if constexpr (LIGHT) {
            // phase A's length filter: the tabulated threshold of the (clamped) sentence length, 0xffff = rejected
            const uint32_t cmin = s_cmin[rb[u].slen < kSigLenClamp ? rb[u].slen : kSigLenClamp];
            keep = keep && cmin != 0xffffu;
            it_need |= cmin;
          }
          const unsigned long long mask = __ballot(keep);
          if (keep) {
            SurvB sv;
            sv.off = rb[u].off; sv.rem_slen = (uint32_t)rb[u].rem | ((uint32_t)rb[u].slen << 16);
            sv.sid = rb[u].sid; sv.it_need = it_need;
            s_surv[n_s + __popcll(mask & ((1ull << lane) - 1ull))] = sv;
          }
          n_s += __popcll(mask);
#ifdef FMX_LIGHT_STATS
          ns_tot += __popcll(mask);
#endif
        }
        e0 += kWave * kBU;
        const bool more = e0 < e_hi;
        if (more) { load_hits(e0); load_recs(); }  // next block's loads overlap this block's token work
        wave_lds_sync();
        uint32_t j0 = 0;
        if (p <= kWave) for (; j0 + pass <= n_s; j0 += pass) heavy4(j0, pass);
        else if constexpr (!LIGHT) for (; j0 + pass <= n_s; j0 += pass) heavy(j0, true);  // (a light query's pattern fits a wave)
        const uint32_t rest = n_s - j0;
        if (!more) {
          if (rest) { if (p <= kWave) heavy4(j0, rest); else if constexpr (!LIGHT) heavy(j0, lane < rest); }
          n_s = 0;
        } else {
          if (j0) {  // carry the incomplete wave of survivors to the front of the list
            SurvB tmp;
            if (lane < rest) tmp = s_surv[j0 + lane];
            wave_lds_sync();
            if (lane < rest) s_surv[lane] = tmp;
            wave_lds_sync();
          }
          n_s = rest;
        }
      }
      if (in_lds) {
        // what k_cand does for sets of up to 64 survivors: one candidate per sentence with its longest
        // match, ordered by (longest desc, sentence id asc)
        wave_lds_sync();
        // keys carry the survivor's slot in their low six bits (sentence id 32 | match length 16 | slot 6): after the two
        // sorts every candidate still knows where its sentence's token base and length sit in LDS
        unsigned long long key = lane < n_out ? ((s_out[lane] << 6) | lane) : ~0ull;
        if (n_out > 1) key = wave_sort64_upto(key, n_out);
        const uint32_t nlo = __shfl_down((uint32_t)key, 1, 64), nhi = __shfl_down((uint32_t)(key >> 32), 1, 64);
        const unsigned long long nxt = lane == 63 ? ~0ull : (((unsigned long long)nhi << 32) | nlo);
        const bool last = lane < n_out && (nxt >> 22) != (key >> 22);
        // (0xffff - L) 16 | sentence id 32 | slot 6
        unsigned long long k2 = last ? (((unsigned long long)(0xffffu - (uint32_t)((key >> 6) & 0xffffu)) << 38) | ((key >> 22) << 6) | (key & 63ull)) : ~0ull;
        const unsigned long long lastm = __ballot(last);
        const uint32_t n_cand = __popcll(lastm);
        if (n_cand > 1) k2 = wave_sort64_upto(k2, n_out);
        else if (n_cand == 1) {  // the single candidate moves to lane 0 without a sorting network
          const int src = __ffsll((long long)lastm) - 1;
          k2 = ((unsigned long long)__shfl((uint32_t)(k2 >> 32), src, 64) << 32) | __shfl((uint32_t)k2, src, 64);
        }
        if (lane < n_cand) {
          const uint32_t slot = (uint32_t)(k2 & 63ull);
          Cand c;
          c.sid = (uint32_t)((k2 >> 6) & 0xffffffffull);
          // the edit-distance kernels find the sentence without going through sent_pos: token base in `score` (0 = unknown:
          // candidates rebuilt by k_cand*), sentence length beside the match length
          c.longest = (0xffffu - (uint32_t)(k2 >> 38)) | ((uint32_t)s_sl[slot] << 16);
          c.score = __uint_as_float(s_tb[slot]); c.penalty = 0.f;
          seg[lane] = c;
        }
        if (lane == 0) { w.q_cand[q] = n_cand; w.q_kept[q] = n_out | kKeptReduced; }
      } else if (sole && lane == 0) {
        w.q_kept[q] = n_out;
      }
#ifdef FMX_LIGHT_STATS
      if (LIGHT && lane == 0) {
        auto bk = [](uint32_t v) { return v <= 4 ? v : (v <= 8 ? 5u : (v <= 16 ? 6u : 7u)); };
        const uint32_t pc = p <= 16 ? 0u : (p <= 32 ? 1u : (p <= 64 ? 2u : 3u));
        atomicAdd(&g_light_stats[n_q < 256 ? n_q : 256], 1ull);                  // [0, 257): hits
        atomicAdd(&g_light_stats[300 + (ns_tot < 64 ? ns_tot : 64)], 1ull);      // [300, 365): survivors of dominance + length
        atomicAdd(&g_light_stats[400 + (n_out < 64 ? n_out : 64)], 1ull);        // [400, 465): kept after coverage
        atomicAdd(&g_light_stats[500 + pc * 8 + bk(ns_tot)], 1ull);              // [500, 532): pattern class x survivors
        atomicAdd(&g_light_stats[540 + bk(ns_tot) * 8 + bk(n_out)], 1ull);       // [540, 604): survivors x kept
        atomicAdd(&g_light_stats[610 + (p < 100 ? p : 100)], 1ull);              // [610, 711): pattern length
        atomicAdd(&g_light_stats[720 + pc * 8 + bk(n_out)], 1ull);               // [720, 752): pattern class x kept
      }
#endif
      if (LIGHT && lane == 0) w.q_pb[q] = n_q;  // statistics: the hits this kernel took over from phase A
    }
    if (!extra) return;
    item_idx += gridDim.x - first_q_blocks;
  }
}

// ---------------------------------------------------------------------------------------------
// bitonic sorts by one workgroup (one wave for the per-query kernels, 1024 threads for the big-set
// kernel).  The key/value storage is a typed pointer so that the LDS instantiation compiles to
// ds_read/ds_write: selecting between an LDS and a global pointer at run time would turn every
// access into a (slow) flat operation.
// ---------------------------------------------------------------------------------------------
template <class K>
__device__ __forceinline__ void block_bitonic_sort(K* keys, uint32_t n_pow2) {
  for (uint32_t k = 2; k <= n_pow2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < n_pow2; i += blockDim.x) {
        const uint32_t ixj = i ^ j;
        if (ixj > i) {
          const K a = keys[i], c = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  }
}

template <class K, class V>
__device__ __forceinline__ void block_bitonic_sort_kv(K* keys, V* vals, uint32_t n_pow2) {
  for (uint32_t k = 2; k <= n_pow2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = threadIdx.x; i < n_pow2; i += blockDim.x) {
        const uint32_t ixj = i ^ j;
        if (ixj > i) {
          const K a = keys[i], c = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > c) == up) {
            keys[i] = c; keys[ixj] = a;
            const V va = vals[i], vc = vals[ixj];
            vals[i] = vc; vals[ixj] = va;
          }
        }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t n) {
  uint32_t v = 1;
  while (v < n) v <<= 1;
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_cand / k_cand_big — LongestMatches semantics (max match length per sentence,
// src/ngram_matches.cc:79-81) then get_longest_matches order (longest desc, s_id asc,
// src/ngram_matches.cc:19-29).  Two key sorts: (sid, L) to find each sentence's max, then
// (~L, sid).  k_cand: one wave per query, up to kLdsSortMax survivors sorted in LDS; larger sets
// (rare: the coverage bound has already run) are queued for k_cand_big, 1024 threads per query
// over the global key scratch.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cand_body(unsigned long long* keys, Cand* seg, uint32_t n, uint32_t np2,
                                              uint32_t* s_count) {
  if (threadIdx.x == 0) *s_count = 0;
  for (uint32_t i = threadIdx.x; i < np2; i += blockDim.x)
    keys[i] = i < n ? (((unsigned long long)seg[i].sid << 16) | (seg[i].longest & 0xffffu)) : ~0ull;
  __syncthreads();
  block_bitonic_sort(keys, np2);
  // keep the last entry of each sentence group (its largest L); rewrite as (0xffff - L, sid)
  for (uint32_t i0 = 0; i0 < np2; i0 += blockDim.x) {
    const uint32_t i = i0 + threadIdx.x;
    unsigned long long k2 = ~0ull;
    bool keep = false;
    if (i < n) {
      const unsigned long long a = keys[i];
      const unsigned long long nxt = (i + 1 < n) ? keys[i + 1] : ~0ull;
      keep = (i + 1 >= n) || ((nxt >> 16) != (a >> 16));
      if (keep) k2 = ((unsigned long long)(0xffffu - (uint32_t)(a & 0xffffu)) << 32) | (a >> 16);
    }
    __syncthreads();
    if (i < np2) keys[i] = k2;
    const uint32_t c = __popcll(__ballot(keep));
    if (lane_id() == 0 && c) atomicAdd(s_count, c);
    __syncthreads();
  }
  block_bitonic_sort(keys, np2);
  const uint32_t kept = *s_count;
  for (uint32_t i = threadIdx.x; i < kept; i += blockDim.x) {
    const unsigned long long a = keys[i];
    Cand c;
    c.sid = (uint32_t)(a & 0xffffffffu);
    c.longest = 0xffffu - (uint32_t)(a >> 32);
    c.score = 0.f; c.penalty = 0.f;
    seg[i] = c;
  }
  return kept;
}

// ascending sort of one 64-bit key per lane across the wave: bitonic network on shuffles, no LDS, no barrier
// N-lane variant: sorts every aligned group of N lanes; with the live keys in lanes [0, n), n <= N, and ~0 in
// all other lanes this is the full sort at a fraction of the 21 compare-exchange stages.
template <int N>
__device__ __forceinline__ unsigned long long wave_sort64_n(unsigned long long key) {
  // the lane number through an opaque copy: the direction masks below ((lane & k) == 0) == ((lane & j) == 0) depend on nothing
  // but the lane, so the compiler computes all 21 of them at the top of the calling kernel and keeps them — 42 scalar registers,
  // spilled to VGPR lanes in k_expand_b — for the length of the kernel; two compares where they are used are cheaper
  uint32_t lane = lane_id();
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)key, j, 64), hi = __shfl_xor((uint32_t)(key >> 32), j, 64);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
      key = ((other < key) == take_min) ? other : key;  // one compare, one select (equal keys: either is fine)
    }
  }
  return key;
}
// key + 32-bit payload
template <int N>
__device__ __forceinline__ void wave_sort64_kv_n(unsigned long long& key, uint32_t& val) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)key, j, 64), hi = __shfl_xor((uint32_t)(key >> 32), j, 64);
      const uint32_t ov = __shfl_xor(val, j, 64);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
      const bool other_less = other < key;  // keys are distinct (or both ~0): no tie handling needed
      const bool take_other = take_min ? other_less : !other_less && other != key;
      key = take_other ? other : key;
      val = take_other ? ov : val;
    }
  }
}
__device__ __forceinline__ void wave_sort64_kv_upto(unsigned long long& key, uint32_t& val, uint32_t n) {
  if (n <= 8) wave_sort64_kv_n<8>(key, val);
  else if (n <= 16) wave_sort64_kv_n<16>(key, val);
  else if (n <= 32) wave_sort64_kv_n<32>(key, val);
  else wave_sort64_kv_n<64>(key, val);
}

// n (wave-uniform) live keys in the first lanes, ~0 elsewhere
__device__ __forceinline__ unsigned long long wave_sort64_upto(unsigned long long key, uint32_t n) {
  if (n <= 8) return wave_sort64_n<8>(key);
  if (n <= 16) return wave_sort64_n<16>(key);
  if (n <= 32) return wave_sort64_n<32>(key);
  return wave_sort64_n<64>(key);
}

__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long key) {
  const uint32_t lane = lane_id();
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)key, j, 64), hi = __shfl_xor((uint32_t)(key >> 32), j, 64);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
      key = ((other < key) == take_min) ? other : key;  // one compare, one select (equal keys: either is fine)
    }
  }
  return key;
}

// k_cand_route — one thread per query: nearly every query's survivors were already reduced to candidates by k_expand_b (in
// LDS), so only the rest is handed to a kernel: up to 64 survivors -> small_list (k_cand: registers), 65..1024 -> mid_list
// (k_cand_mid: LDS bitonic), more -> big_list (k_cand_big: block radix sort).
// big_routed (round 6): the huge sets were listed by k_cand_route_big and are being sorted on the side lane already.
__global__ void __launch_bounds__(256) k_cand_route(DevBatch b, DevWork w, uint32_t* big_list, uint32_t* mid_list, uint32_t* small_list, int big_routed) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.nq) return;
  const uint32_t n = w.q_kept[q];
  if (n & kKeptReduced) return;  // k_expand_b already wrote the candidates and q_cand
  if (big_routed && n > kLdsSortMax && n <= w.q_scap[q]) return;  // (k_cand_big wrote q_cand)
  w.q_cand[q] = 0;
  if (n == 0) return;
  if (n > w.q_scap[q]) { atomicMax(&w.counters[4], (unsigned long long)n); return; }  // more survivors than the segment holds
  if (n > kLdsSortMax) big_list[atomicAdd(&w.counters[7], 1ull)] = q;
  else if (n > kWave) mid_list[atomicAdd(&w.counters[13], 1ull)] = q;
  else small_list[atomicAdd(&w.counters[23], 1ull)] = q;
}

// k_cand_route_big (round 6) — the few queries with more than kLdsSortMax survivors, listed as soon as k_expand_b<false> is through
// (a light query has at most light_max hits: never one of them), so that k_cand_big — one workgroup per huge set, three 50 us radix
// passes: 0.29 ms during which a pass otherwise leaves the GPU almost idle — sorts them on the side lane BESIDE the light queries' hit walk.
__global__ void __launch_bounds__(256) k_cand_route_big(DevBatch b, DevWork w, uint32_t* big_list) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= b.nq || w.q_capl[q] != 0) return;   // (light queries: their kernel may still be running)
  const uint32_t n = w.q_kept[q];
  if ((n & kKeptReduced) || n <= kLdsSortMax || n > w.q_scap[q]) return;
  big_list[atomicAdd(&w.counters[7], 1ull)] = q;
}

// k_cand — one wave per query of small_list: up to 64 survivors are de-duplicated and ordered entirely in registers.
__global__ void __launch_bounds__(kWave) k_cand(DevWork w, const uint32_t* small_list) {
  const unsigned long long n_small = w.counters[23];
  const uint32_t lane = lane_id();
  for (unsigned long long idx = blockIdx.x; idx < n_small; idx += gridDim.x) {
    const uint32_t q = small_list[idx];
    const uint32_t n = w.q_kept[q];
    Cand* seg = w.seg + w.q_seg[q];
    unsigned long long key = ~0ull;
    if (lane < n) key = ((unsigned long long)seg[lane].sid << 16) | (seg[lane].longest & 0xffffu);
    key = wave_sort64(key);
    // keep the last entry of each sentence group (its largest L); rewrite as (0xffff - L, sid)
    const uint32_t nlo = __shfl_down((uint32_t)key, 1, 64), nhi = __shfl_down((uint32_t)(key >> 32), 1, 64);
    const unsigned long long nxt = lane == 63 ? ~0ull : (((unsigned long long)nhi << 32) | nlo);
    const bool keep = lane < n && (nxt >> 16) != (key >> 16);
    unsigned long long k2 = keep ? (((unsigned long long)(0xffffu - (uint32_t)(key & 0xffffu)) << 32) | (key >> 16)) : ~0ull;
    const uint32_t kept = __popcll(__ballot(keep));
    k2 = wave_sort64(k2);
    if (lane < kept) {
      Cand c;
      c.sid = (uint32_t)(k2 & 0xffffffffu);
      c.longest = 0xffffu - (uint32_t)(k2 >> 32);
      c.score = 0.f; c.penalty = 0.f;
      seg[lane] = c;
    }
    if (lane == 0) w.q_cand[q] = kept;
  }
}

__global__ void __launch_bounds__(256) k_cand_mid(DevWork w, const uint32_t* mid_list) {
  __shared__ unsigned long long s_keys[kLdsSortMax];
  __shared__ uint32_t s_count;
  const unsigned long long n_mid = w.counters[13];
  for (unsigned long long idx = blockIdx.x; idx < n_mid; idx += gridDim.x) {
    const uint32_t q = mid_list[idx];
    const uint32_t n = w.q_kept[q];
    const uint32_t kept = cand_body(s_keys, w.seg + w.q_seg[q], n, next_pow2(n), &s_count);
    if (threadIdx.x == 0) w.q_cand[q] = kept;
    __syncthreads();
  }
}

#ifndef FMX_BIG_RU
#define FMX_BIG_RU 4
#endif
constexpr int kBigThreads = 1024;
constexpr int kBigWaves = kBigThreads / 64;
constexpr int kBigGrid = 512;

__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* s_wsum) {
  const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t n = __shfl_up(x, d, 64);
    if (lane >= (uint32_t)d) x += n;
  }
  if (lane == 63) s_wsum[wv] = x;
  __syncthreads();
  if (wv == 0) {
    uint32_t y = lane < (uint32_t)kBigWaves ? s_wsum[lane] : 0u;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t n = __shfl_up(y, d, 64);
      if (lane >= (uint32_t)d) y += n;
    }
    if (lane < (uint32_t)kBigWaves) s_wsum[lane] = y;
  }
  __syncthreads();
  const uint32_t base = wv ? s_wsum[wv - 1] : 0u;
  __syncthreads();
  return base + x - v;
}

// One stable LSD radix pass (8-bit digit at `shift`) over n 64-bit keys, src -> dst, by a 1024-thread
// workgroup.  Each wave owns a contiguous chunk; s_hist[wave][digit] first counts, then (after an
// exclusive scan in digit-major, wave-minor order) holds the running output position.  Within a wave
// the rank among equal digits comes from eight ballots, so the pass is stable.
__device__ __forceinline__ void block_radix_pass(const unsigned long long* src, unsigned long long* dst, uint32_t n,
                                                 uint32_t shift, uint32_t* s_hist, uint32_t* s_wsum) {
  const uint32_t tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t len = (n + kBigWaves - 1) / kBigWaves;
  len = (len + 63u) & ~63u;
  const uint32_t beg = wv * len < n ? wv * len : n;
  const uint32_t end = beg + len < n ? beg + len : n;
  for (uint32_t i = tid; i < kBigWaves * 256; i += kBigThreads) s_hist[i] = 0;
  __syncthreads();
  constexpr int kRU = FMX_BIG_RU;  // keys requested together: both loops are chains of global-load latencies otherwise
  for (uint32_t i0 = beg; i0 < end; i0 += 64 * kRU) {
    unsigned long long kk[kRU];
#pragma unroll
    for (int u = 0; u < kRU; u++) { const uint32_t i = i0 + u * 64 + lane; kk[u] = i < end ? src[i] : 0ull; }
#pragma unroll
    for (int u = 0; u < kRU; u++)
      if (i0 + u * 64 + lane < end) atomicAdd(&s_hist[wv * 256 + (uint32_t)((kk[u] >> shift) & 255ull)], 1u);
  }
  __syncthreads();
  {
    // entry e = digit * kBigWaves + wave; each thread owns 4 consecutive entries (kBigWaves * 256 / 1024)
    constexpr int per = kBigWaves * 256 / kBigThreads;
    uint32_t v[per], sum = 0;
#pragma unroll
    for (int k = 0; k < per; k++) {
      const uint32_t e = tid * per + k;
      v[k] = s_hist[(e % kBigWaves) * 256 + e / kBigWaves];
      sum += v[k];
    }
    uint32_t ex = block_excl_scan_u32(sum, s_wsum);
#pragma unroll
    for (int k = 0; k < per; k++) {
      const uint32_t e = tid * per + k;
      s_hist[(e % kBigWaves) * 256 + e / kBigWaves] = ex;
      ex += v[k];
    }
  }
  __syncthreads();
  for (uint32_t g0 = beg; g0 < end; g0 += 64 * kRU) {
    unsigned long long kk[kRU];
#pragma unroll
    for (int u = 0; u < kRU; u++) { const uint32_t i = g0 + u * 64 + lane; kk[u] = i < end ? src[i] : 0ull; }
#pragma unroll
    for (int u = 0; u < kRU; u++) {
    const uint32_t i0 = g0 + u * 64;
    if (i0 >= end) break;
    const uint32_t i = i0 + lane;
    const bool valid = i < end;
    const unsigned long long key = kk[u];
    const uint32_t d = (uint32_t)((key >> shift) & 255ull);
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; bit++) {
      const bool one = (d >> bit) & 1u;
      const unsigned long long bal = __ballot(valid && one);
      same &= one ? bal : ~bal;
    }
    if (valid) {
      const uint32_t rank = __popcll(same & ((1ull << lane) - 1ull));
      const uint32_t base = s_hist[wv * 256 + d];
      dst[base + rank] = key;
      if (rank == 0) s_hist[wv * 256 + d] = base + __popcll(same);
    }
    }
  }
  __syncthreads();
}

// k_cand_big — survivor sets beyond the LDS sorter: LSD radix sort by sentence id, per-sentence max
// of the match length, then a stable radix sort by (0xffff - L): the reference's (longest desc, s_id
// asc).  `big` is the global key scratch laid out like the segments at 2x: two buffers of n keys.
__global__ void __launch_bounds__(kBigThreads) k_cand_big(DevWork w, unsigned long long* big, const uint32_t* big_list,
                                                          uint32_t sid_bytes, uint32_t len_bytes) {
  __shared__ uint32_t s_hist[kBigWaves * 256];
  __shared__ uint32_t s_wsum[kBigWaves];
  __shared__ uint32_t s_count;
  const unsigned long long n_big = w.counters[7];
  for (unsigned long long idx = blockIdx.x; idx < n_big; idx += gridDim.x) {
    const uint32_t q = big_list[idx];
    const uint32_t n = w.q_kept[q];
    Cand* seg = w.seg + w.q_seg[q];
    unsigned long long* A = big + 2ull * w.q_seg[q];
    unsigned long long* B = A + n;
    if (threadIdx.x == 0) s_count = 0;
    for (uint32_t i0 = threadIdx.x; i0 < n; i0 += kBigThreads * 4) {
      unsigned long long kk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t i = i0 + u * kBigThreads;
        kk[u] = 0;
        if (i < n) { const Cand c = seg[i]; kk[u] = ((unsigned long long)c.sid << 16) | (c.longest & 0xffffu); }
      }
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * kBigThreads; if (i < n) A[i] = kk[u]; }
    }
    __syncthreads();
    unsigned long long *src = A, *dst = B;
    for (uint32_t k = 0; k < sid_bytes; k++) {
      block_radix_pass(src, dst, n, 16 + 8 * k, s_hist, s_wsum);
      unsigned long long* t = src; src = dst; dst = t;
    }
    // per-sentence max: among the entries of one sentence keep the largest L (first one on ties)
    uint32_t my_kept = 0;
    for (uint32_t i = threadIdx.x; i < n; i += kBigThreads) {
      const unsigned long long a = src[i];
      const unsigned long long sid = a >> 16;
      const uint32_t L = (uint32_t)(a & 0xffffu);
      bool keep = true;
      for (uint32_t j = i; j > 0 && (src[j - 1] >> 16) == sid;) { j--; if ((uint32_t)(src[j] & 0xffffu) >= L) keep = false; }
      for (uint32_t j = i; j + 1 < n && (src[j + 1] >> 16) == sid;) { j++; if ((uint32_t)(src[j] & 0xffffu) > L) keep = false; }
      dst[i] = keep ? (((unsigned long long)(0xffffu - L) << 32) | sid) : ~0ull;
      my_kept += keep ? 1u : 0u;
    }
    my_kept = wave_sum(my_kept);
    if (lane_id() == 0 && my_kept) atomicAdd(&s_count, my_kept);
    __syncthreads();
    { unsigned long long* t = src; src = dst; dst = t; }
    for (uint32_t k = 0; k < len_bytes; k++) {
      block_radix_pass(src, dst, n, 32 + 8 * k, s_hist, s_wsum);
      unsigned long long* t = src; src = dst; dst = t;
    }
    const uint32_t kept = s_count;
    for (uint32_t i0 = threadIdx.x; i0 < kept; i0 += kBigThreads * 4) {
      unsigned long long kk[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const uint32_t i = i0 + u * kBigThreads; kk[u] = i < kept ? src[i] : 0ull; }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t i = i0 + u * kBigThreads;
        if (i < kept) {
          Cand c;
          c.sid = (uint32_t)(kk[u] & 0xffffffffu);
          c.longest = 0xffffu - (uint32_t)((kk[u] >> 32) & 0xffffu);
          c.score = 0.f; c.penalty = 0.f;
          seg[i] = c;
        }
      }
    }
    if (threadIdx.x == 0) w.q_cand[q] = kept;
    __syncthreads();
  }
}

// Candidates are laid out as three dense lists by pattern-length class so that every edit-distance kernel
// runs with full waves: class 0 = up to 32 pattern tokens (row in 32 registers), 1 = up to 64, 2 = longer
// or any query that needs the FULL cost model (rows in LDS).  k_classify feeds three scans; k_cmap then
// writes (query, candidate index) pairs, class 0 first, class 1 after it, class 2 last.
// Pattern-length order of the queries.  The DP candidate lists are laid out in this order, so the 64
// consecutive candidates of a wave almost always share one pattern length and the register DP can run the
// variant compiled for exactly that length (no padding columns, no per-column guards).
constexpr uint32_t kPBins = 66;  // lengths 0..64 one bin each, longer patterns the last bin
__device__ __forceinline__ uint32_t p_bin(uint32_t p) { return p < kPBins - 1 ? p : kPBins - 1; }

constexpr uint32_t kRankThreads = 1024, kRankPer = 4;  // queries per block = 4096: few global atomics per bin

__global__ void __launch_bounds__(kRankThreads) k_phist(DevBatch b, DevWork w) {
  __shared__ uint32_t s_h[kPBins];
  if (threadIdx.x < kPBins) s_h[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (uint32_t u = 0; u < kRankPer; u++) {
    const uint32_t q = (blockIdx.x * kRankPer + u) * kRankThreads + threadIdx.x;
    if (q < b.nq) atomicAdd(&s_h[p_bin(b.q_off[b.q_begin + q + 1] - b.q_off[b.q_begin + q])], 1u);
  }
  __syncthreads();
  if (threadIdx.x < kPBins && s_h[threadIdx.x]) atomicAdd(&w.p_hist[threadIdx.x], s_h[threadIdx.x]);
}

__global__ void __launch_bounds__(kRankThreads) k_prank(DevBatch b, DevWork w) {
  __shared__ uint32_t s_cnt[kPBins], s_base[kPBins];
  if (threadIdx.x < kPBins) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t bin[kRankPer], local[kRankPer];
#pragma unroll
  for (uint32_t u = 0; u < kRankPer; u++) {
    const uint32_t q = (blockIdx.x * kRankPer + u) * kRankThreads + threadIdx.x;
    bin[u] = 0; local[u] = 0;
    if (q < b.nq) {
      bin[u] = p_bin(b.q_off[b.q_begin + q + 1] - b.q_off[b.q_begin + q]);
      local[u] = atomicAdd(&s_cnt[bin[u]], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < kPBins) {
    uint32_t base = 0;
    for (uint32_t k = 0; k < threadIdx.x; k++) base += w.p_hist[k];
    const uint32_t mine = s_cnt[threadIdx.x];
    s_base[threadIdx.x] = base + (mine ? atomicAdd(&w.p_hist[kPBins + threadIdx.x], mine) : 0u);
  }
  __syncthreads();
#pragma unroll
  for (uint32_t u = 0; u < kRankPer; u++) {
    const uint32_t q = (blockIdx.x * kRankPer + u) * kRankThreads + threadIdx.x;
    if (q < b.nq) w.q_rank[q] = s_base[bin[u]] + local[u];  // order inside one length is irrelevant
  }
}

// length class of a pattern's candidates in the dense lists.  full: 0 = ids only (register kernels for p <= 32 / <= 64, LDS rows
// beyond), 1 = every pair through the general kernel (class 2), 2 = real tokens through the register kernels (round 6), 3 = the
// same with the idf term (four register arrays per pattern token: rows in place; 432 registers for 49..64 tokens — few pairs, and
// still far ahead of the general kernel, whose handful of long pairs took 0.8 ms of C4's pass on a default index)
__device__ __forceinline__ uint32_t dp_class(uint32_t p, int full) {
  if (full == 1) return 2u;
  return p <= 32 ? 0u : (p <= 64 ? 1u : 2u);
}

__global__ void k_classify(DevBatch b, DevWork w, int full) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > b.nq) return;
  unsigned long long v[3] = {0, 0, 0};
  if (q < b.nq) {
    const uint32_t p = b.q_off[b.q_begin + q + 1] - b.q_off[b.q_begin + q];
    v[dp_class(p, full)] = w.q_cand[q];
  }
  const size_t stride = (size_t)b.nq + 1;
  const uint32_t r = q < b.nq ? w.q_rank[q] : b.nq;  // list position: pattern-length order
  w.c_in[r] = v[0]; w.c_in[stride + r] = v[1]; w.c_in[2 * stride + r] = v[2];
}

// lane per query: small candidate sets are written by their own lane, large ones by the whole wave
constexpr uint32_t kCmapSmall = 16;
__global__ void __launch_bounds__(1024) k_cmap(DevBatch b, DevWork w, int full, uint32_t* fin_list, uint32_t list_above) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t lane = lane_id();
  uint32_t n = 0;
  unsigned long long base = 0;
  if (q < b.nq) n = w.q_cand[q];
  {
    // queries for k_finish<.., true> (replay + large sorter): ONE atomic per workgroup (same-address atomics retire one at a
    // time: with the pair kernel a third of the waves list something)
    __shared__ uint32_t s_wave[16];
    __shared__ unsigned long long s_base;
    const bool listed = n > list_above;
    const unsigned long long lm = __ballot(listed);
    const uint32_t wv = threadIdx.x >> 6, n_wv = blockDim.x >> 6;
    if (lane == 0) s_wave[wv] = (uint32_t)__popcll(lm);
    __syncthreads();
    if (threadIdx.x == 0) {
      uint32_t tot = 0;
      for (uint32_t k = 0; k < n_wv; k++) { const uint32_t c = s_wave[k]; s_wave[k] = tot; tot += c; }
      s_base = tot ? atomicAdd(&w.counters[14], (unsigned long long)tot) : 0ull;
    }
    __syncthreads();
    if (listed) fin_list[s_base + s_wave[wv] + __popcll(lm & ((1ull << lane) - 1ull))] = q;
  }
  if (q < b.nq) {
    if (n) {
      const size_t stride = (size_t)b.nq + 1;
      const uint32_t p = b.q_off[b.q_begin + q + 1] - b.q_off[b.q_begin + q];
      const uint32_t cls = dp_class(p, full);
      base = w.c_off[cls * stride + w.q_rank[q]];
      if (cls >= 1) base += w.c_off[b.nq];
      if (cls >= 2) base += w.c_off[stride + b.nq];
    }
  }
  // list entry = query << 32 | absolute slot of the candidate in seg (the host keeps seg_cap below 2^32): the edit-distance
  // kernels request the candidate record together with the query's offsets, not after q_seg[q]
  const uint32_t s0 = n ? (uint32_t)w.q_seg[q] : 0u;
  if (n <= kCmapSmall)
    for (uint32_t i = 0; i < n; i++) w.c_qi[base + i] = ((unsigned long long)q << 32) | (s0 + i);
  unsigned long long big = __ballot(n > kCmapSmall);
  while (big) {
    const int l = __ffsll((long long)big) - 1;
    big &= big - 1ull;
    const uint32_t qb = __shfl(q, l, 64), nb = __shfl(n, l, 64), sb = __shfl(s0, l, 64);
    const unsigned long long bb = ((unsigned long long)__shfl((uint32_t)(base >> 32), l, 64) << 32) | __shfl((uint32_t)base, l, 64);
    for (uint32_t i = lane; i < nb; i += kWave) w.c_qi[bb + i] = ((unsigned long long)qb << 32) | (sb + i);
  }
}

// [begin, end) of a class in the dense candidate list
__device__ __forceinline__ void dp_class_range(const DevBatch& b, const DevWork& w, uint32_t cls, unsigned long long& lo,
                                               unsigned long long& hi) {
  const size_t stride = (size_t)b.nq + 1;
  const unsigned long long t0 = w.c_off[b.nq], t1 = w.c_off[stride + b.nq], t2 = w.c_off[2 * stride + b.nq];
  lo = cls == 0 ? 0ull : (cls == 1 ? t0 : t0 + t1);
  hi = lo + (cls == 0 ? t0 : (cls == 1 ? t1 : t2));
}

// ---------------------------------------------------------------------------------------------
// k_dp — the float edit distance (src/edit_distance.cc:5-77), flattened over every (query,
// candidate) pair of the sub-batch: one candidate per lane whatever its query, so a query with
// ten thousand candidates is spread over the whole chip and a query with three does not idle 61
// lanes.  DP rows are strided through LDS (bank = lane, conflict free).  Writes final cost and
// M = max(row minima, final) into the candidate record; the order-dependent acceptance
// (src/fuzzy_match.cc:587-610) is replayed afterwards by k_finish.
// Dynamic LDS: rows[(p_max+1)*64] floats.
// ---------------------------------------------------------------------------------------------
struct LdsRow {
  float* base;
  __device__ __forceinline__ float& operator[](uint32_t j) const { return base[j * kWave]; }
};

// GROWS: the rows live in HBM (g_rows: row_len x 64 floats per workgroup) instead of dynamic LDS — patterns longer
// than ~590 tokens (--max-tokens-in-pattern above the default, include/fuzzy/suffix_array_index.hh:15).
// ext_list (round 6): the pairs k_dp_reg<.., REAL> deferred (counters[kCtrDpDefer] of them) instead of length class 2.
template <bool FULL, bool GROWS>
__global__ void __launch_bounds__(kWave) k_dp(DevIndex ix, DevBatch b, DevWork w, Params prm, float* g_rows, uint32_t row_len,
                                              const unsigned long long* ext_list = nullptr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_rows = GROWS ? g_rows + (size_t)blockIdx.x * row_len * kWave : (float*)smem;
  const uint32_t lane = lane_id();
  unsigned long long j_lo, j_hi;
  dp_class_range(b, w, 2, j_lo, j_hi);
  const unsigned long long* list = w.c_qi;
  if (ext_list) { list = ext_list; j_lo = 0; j_hi = w.counters[kCtrDpDefer]; }
  const bool use_idf = prm.idf_penalty != 0.f;
  for (unsigned long long j = j_lo + (unsigned long long)blockIdx.x * kWave + lane; j < j_hi;
       j += (unsigned long long)gridDim.x * kWave) {
    const unsigned long long qi = list[j];
    const uint32_t q = (uint32_t)(qi >> 32);
    Cand* cd = w.seg + (uint32_t)qi;
    const Cand c0 = *cd;
    const uint32_t o0 = b.q_off[b.q_begin + q];
    const uint32_t p = b.q_off[b.q_begin + q + 1] - o0;
    const uint32_t* pat = b.q_tok + o0;
    uint32_t tb = __float_as_uint(c0.score), slen = c0.longest >> 16;  // token base handed over by k_expand_b, 0 = unknown
    if (tb == 0) { const uint32_t sp0 = ix.sent_pos[c0.sid]; slen = ix.tok[sp0]; tb = sp0 + 1; }
    const uint32_t sp = tb - 1;
    const uint32_t* sent = ix.tok + tb;
    const float dw = 100.f / normalizer(p, slen, prm.ins, prm.del, prm.rep);
    DpSide side;
    side.s_real = nullptr; side.p_real = nullptr; side.s_itok = nullptr; side.p_itok = nullptr;
    side.itok_dist = nullptr; side.n_itok = 0; side.p_idf = nullptr; side.idf_weight = 0.f;
    if (FULL) {
      if (ix.real && b.q_real) { side.s_real = ix.real + sp + 1; side.p_real = b.q_real + o0; }
      if (ix.itok_dist) {
        side.s_itok = ix.itok ? ix.itok + sp : nullptr;
        side.p_itok = b.q_itok ? b.q_itok + o0 + (b.q_begin + q) : nullptr;
        // a pair without a penalty token on either side (the common one) only ever reads zeros from the distance table: every
        // cost_tag term is + 0.f, the same bits as no term — skip the three table reads per cell
        uint32_t any = ext_list ? 1u : 0u;
        if (!any && side.s_itok) for (uint32_t i = 0; i <= slen; i++) any |= side.s_itok[i];
        if (!any && side.p_itok) for (uint32_t i = 0; i <= p; i++) any |= side.p_itok[i];
        if (any) { side.itok_dist = ix.itok_dist; side.n_itok = ix.n_itok; }
      }
      if (use_idf) {
        side.p_idf = w.t_idf + (o0 - b.t_begin);
        side.idf_weight = (dw * prm.idf_penalty) / prm.idf_max;  // src/fuzzy_match.cc:591
      }
    }
    float M;
    const float final_cost = edit_distance<FULL>(sent, slen, pat, p, prm.ins, prm.del, prm.rep, dw, side, FLT_MAX,
                                                 LdsRow{s_rows + lane}, M);
    cd->score = final_cost;
    cd->penalty = M;
    cd->longest = (c0.longest & 0xffffu) | (slen << 16);
  }
}

// ---------------------------------------------------------------------------------------------
// k_dp_reg — the ids-only edit distance (no real-token / penalty-token / idf terms) for patterns of up
// to 64 tokens with the DP row and the pattern held in registers: every loop over the pattern is fully
// unrolled, columns beyond the pattern length compute don't-care values that never feed a valid cell
// (dependencies only run left to right), and no LDS or memory access remains in the cell update.
// Same operations in the same order as edit_distance<false>, so the results are bit-identical.
// ---------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) Tok4 {
  uint32_t t[4];
};

// REAL (round 6): the real-token term of the full _edit_distance (src/edit_distance.cc:53-62) for pairs WITHOUT penalty tokens — equal
// words whose real tokens differ cost replace x diff_case (1) when the sentence's real token is a case feature, replace x diff_real (2)
// otherwise (include/fuzzy/costs.hh:58-61; NOT scaled by diff_word).  s_real / p_real: interned real-token ids per token slot (bit 31 of
// the sentence's = "first byte in LUMC or empty").  With every cost_tag term zero the full recurrence adds 0.f where this one adds
// nothing: same bits.  What an index built under the reference CLI's default -p nbr,tag,cas runs for every candidate.
template <int P, bool IDF = false, bool EXACT = false, bool REAL = false>
__device__ __forceinline__ float edit_distance_reg(const uint32_t* __restrict__ sent, uint32_t n1,
                                                   const uint32_t* __restrict__ pat, uint32_t n2, float ins,
                                                   float del, float rep, float dw, float& M_out,
                                                   const float* __restrict__ p_idf = nullptr, float idf_weight = 0.f,
                                                   const uint32_t* __restrict__ s_real = nullptr,
                                                   const uint32_t* __restrict__ p_real = nullptr) {
  const float del_dw = del * dw, ins_dw = ins * dw, rep_dw = rep * dw;
  const float rep_case = rep * 1.f, rep_real = rep * 2.f;
  float    row[P + 1];
  uint32_t pt[P];
  uint32_t pr[REAL ? P : 1];
  float    pen[IDF ? P : 1];
#pragma unroll
  for (int j = 0; j < P; j++) {
    pt[j] = (EXACT || (uint32_t)j < n2) ? pat[j] : 0xffffffffu;  // EXACT: n2 == P for every lane of the wave
    if (REAL) pr[j] = (EXACT || (uint32_t)j < n2) ? p_real[j] : 0u;
    if (IDF) pen[j] = (EXACT || (uint32_t)j < n2) ? p_idf[j] * idf_weight : 0.f;  // idf_penalty[j] * idf_weight, edit_distance.cc:36,49
  }
  row[0] = 0.f;
#pragma unroll
  for (int j = 1; j <= P; j++) {
    float v = row[j - 1] + ins_dw;
    if (IDF) v = v + pen[j - 1];
    row[j] = v;
  }
  float final_v = EXACT ? row[P] : 0.f;
  if (!EXACT) {
#pragma unroll
    for (int j = 0; j <= P; j++) final_v = (uint32_t)j == n2 ? row[j] : final_v;
  }
  // One DP row: `in` = row i - 1, `out` = row i (two register sets used in turn: a row written in place needs a copy per cell to keep
  // the old value alive as the next cell's diagonal).  M = max(row minima over columns >= 1, final): the minima are NOT monotone in
  // general (a cheap delete, or a real-token cost above diff_word, lets a later row dip below an earlier one), so all are kept.
  float M = 0.f;
  auto row_step = [&](const float (&in)[P + 1], float (&out)[P + 1], uint32_t wd, uint32_t rl) {
    const uint32_t rd = rl & 0x7fffffffu;
    const float c_real = (rl >> 31) ? rep_case : rep_real;  // strchr("LUMC", real1tok[i-1][0]), edit_distance.cc:57
    float left = in[0] + del_dw;
    out[0] = left;
    float mn = FLT_MAX;
#pragma unroll
    for (int j = 1; j <= P; j++) {
      float bb = left + ins_dw;
      const float same = REAL ? ((rd != pr[REAL ? j - 1 : 0]) ? c_real : 0.f) : 0.f;
      float diff = (wd != pt[j - 1]) ? rep_dw : same;
      if (IDF) {
        bb = bb + pen[j - 1];
        diff = (wd != pt[j - 1]) ? (rep_dw + pen[j - 1]) : same;
      }
      // v_min_f32: same value as std::min for the finite costs of this path
      const float d = __builtin_fminf(__builtin_fminf(in[j] + del_dw, bb), in[j - 1] + diff);
      out[j] = d;
      const float mn2 = __builtin_fminf(mn, d);
      if (EXACT) {
        mn = mn2;
        if (j == P) final_v = d;
      } else {
        mn = (uint32_t)j <= n2 ? mn2 : mn;
        final_v = (uint32_t)j == n2 ? d : final_v;
      }
      left = d;
    }
    if (mn > M) M = mn;
  };
  // idf AND real tokens (four register arrays of P): the second row set would leave one wave per SIMD (270 registers) — the row
  // is updated in place there, a copy per cell (C4's parameters on the default index: 1.36 instead of 1.49 ms)
  constexpr bool kInPlace = IDF && REAL;
  auto row_step_ip = [&](float (&r)[P + 1], uint32_t wd, uint32_t rl) {
    const uint32_t rd = rl & 0x7fffffffu;
    const float c_real = (rl >> 31) ? rep_case : rep_real;
    float diag = r[0];
    float left = diag + del_dw;
    r[0] = left;
    float mn = FLT_MAX;
#pragma unroll
    for (int j = 1; j <= P; j++) {
      const float up = r[j];
      float bb = left + ins_dw;
      const float same = REAL ? ((rd != pr[REAL ? j - 1 : 0]) ? c_real : 0.f) : 0.f;
      float diff = (wd != pt[j - 1]) ? rep_dw : same;
      if (IDF) {
        bb = bb + pen[j - 1];
        diff = (wd != pt[j - 1]) ? (rep_dw + pen[j - 1]) : same;
      }
      const float d = __builtin_fminf(__builtin_fminf(up + del_dw, bb), diag + diff);
      r[j] = d;
      const float mn2 = __builtin_fminf(mn, d);
      if (EXACT) {
        mn = mn2;
        if (j == P) final_v = d;
      } else {
        mn = (uint32_t)j <= n2 ? mn2 : mn;
        final_v = (uint32_t)j == n2 ? d : final_v;
      }
      diag = up;
      left = d;
    }
    if (mn > M) M = mn;
  };
  float rowb[kInPlace ? 1 : P + 1];
  // The sentence's tokens (and real-token ids) are fetched four at a time — one 16-byte load, any 4-byte alignment; both buffers
  // have spare words behind them — one group ahead of their use (round 6: a 4-byte load per row and lane asked the L2 for a
  // 128-byte line each, 85 % of its request rate at three waves per SIMD, and the kernel waited 74 % of its cycles).
  const Tok4* sent4 = reinterpret_cast<const Tok4*>(sent);
  const Tok4* real4 = reinterpret_cast<const Tok4*>(s_real);
  Tok4 nw = n1 ? sent4[0] : Tok4{{0u, 0u, 0u, 0u}};
  Tok4 nr = (REAL && !kInPlace && n1) ? real4[0] : Tok4{{0u, 0u, 0u, 0u}};
  for (uint32_t i0 = 0; i0 < n1; i0 += 4) {
    const Tok4 cw = nw;
    const Tok4 cr = kInPlace ? real4[i0 >> 2] : nr;  // in place: the real ids of THIS group, four registers fewer — two waves per SIMD
    if (i0 + 4 < n1) {
      nw = sent4[(i0 >> 2) + 1];
      if (REAL && !kInPlace) nr = real4[(i0 >> 2) + 1];
    }
    const uint32_t nu = n1 - i0;  // rows of this group: 4, fewer in the sentence's last one (whose state nobody reads afterwards)
    if constexpr (kInPlace) {
      row_step_ip(row, cw.t[0], cr.t[0]);
      if (nu > 1) row_step_ip(row, cw.t[1], cr.t[1]);
      if (nu > 2) row_step_ip(row, cw.t[2], cr.t[2]);
      if (nu > 3) row_step_ip(row, cw.t[3], cr.t[3]);
    } else {
      auto& rb = reinterpret_cast<float (&)[P + 1]>(rowb);
      row_step(row, rb, cw.t[0], cr.t[0]);
      if (nu > 1) row_step(rb, row, cw.t[1], cr.t[1]);
      if (nu > 2) row_step(row, rb, cw.t[2], cr.t[2]);
      if (nu > 3) row_step(rb, row, cw.t[3], cr.t[3]);
    }
  }
  if (final_v > M) M = final_v;
  M_out = M;
  return final_v;
}

// eq = 2 * eq + (a == b): a compare into VCC and an add-with-carry — two VALU instructions per pattern token.  (The compiler's
// version of `eq = (eq << 1) | (a == b)` is compare, wait state, select of a constant, and half a shift-or: the match mask
// was two thirds of the bit-parallel kernel's instructions.)
__device__ __forceinline__ void eq_push(uint32_t& eq, uint32_t a, uint32_t b) {
  asm("v_cmp_eq_u32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(eq) : "v"(a), "v"(b) : "vcc");
}

// Unit costs: bit-parallel Levenshtein distance (fmx_core.hh) with the pattern in registers, then S_k.
// P = register capacity; EXACT: every lane's pattern has exactly P tokens, otherwise columns beyond n2 hold
// 0xffffffff (never equal, their bits stay 0 and never reach bit n2 - 1).
template <int P, bool EXACT>
__device__ __forceinline__ float edit_distance_unit(const uint32_t* __restrict__ sent, uint32_t n1,
                                                    const uint32_t* __restrict__ pat, uint32_t n2, float inc, float& M_out) {
  uint32_t pt[P];
#pragma unroll
  for (int j = 0; j < P; j++) pt[j] = (EXACT || (uint32_t)j < n2) ? pat[j] : 0xffffffffu;
  uint32_t k;
  // The recurrence is short, so the sentence's tokens are fetched four at a time (one 16-byte load, any
  // 4-byte alignment; the token buffer has spare tokens behind it) one group ahead of their use.
  const Tok4* sent4 = reinterpret_cast<const Tok4*>(sent);
  Tok4 nxt = n1 ? sent4[0] : Tok4{{0u, 0u, 0u, 0u}};
  uint32_t pv = ~0u, mv = 0u, score = n2;
  LevState st;
  lev_init(st, n2);
  for (uint32_t i0 = 0; i0 < n1; i0 += 4) {
    uint32_t c0 = nxt.t[0], c1 = nxt.t[1], c2 = nxt.t[2], c3 = nxt.t[3];
    if (i0 + 4 < n1) nxt = sent4[(i0 >> 2) + 1];
    const uint32_t nu = n1 - i0 < 4 ? n1 - i0 : 4;
#pragma unroll 1
    for (uint32_t u = 0; u < nu; u++) {  // not unrolled: one copy of the row per pattern length keeps the code small
      const uint32_t wd = c0;
      c0 = c1; c1 = c2; c2 = c3;
      if (P <= 32) {
        // two independent chains (upper / lower half of the pattern), joined by one shift-or
        constexpr int H = P / 2;
        uint32_t eq_a = 0, eq_b = 0;
#pragma unroll
        for (int t = 0; t < P - H; t++) {  // interleaved: neither chain waits for its own previous add
          eq_push(eq_a, wd, pt[P - 1 - t]);
          if (H - 1 - t >= 0) eq_push(eq_b, wd, pt[H - 1 - t >= 0 ? H - 1 - t : 0]);
        }
        lev_step32(pv, mv, score, H ? ((eq_a << H) | eq_b) : eq_a, n2);
      } else {
        uint32_t eq_lo = 0, eq_hi = 0;
#pragma unroll
        for (int t = 0; t < 32; t++) {
          eq_push(eq_lo, wd, pt[31 - t]);
          if (P - 1 - t >= 32) eq_push(eq_hi, wd, pt[P - 1 - t >= 32 ? P - 1 - t : 32]);
        }
        lev_step(st, ((unsigned long long)eq_hi << 32) | eq_lo, n2);
      }
    }
  }
  k = P <= 32 ? score : st.score;
  const float f = unit_cost_sum(k, inc);
  M_out = f;
  return f;
}

template <int LO, int HI>
__device__ __forceinline__ float edit_distance_unit_dispatch(uint32_t p_u, const uint32_t* __restrict__ sent, uint32_t n1,
                                                             const uint32_t* __restrict__ pat, float inc, float& M) {
  if constexpr (LO == HI) {
    return edit_distance_unit<LO, true>(sent, n1, pat, (uint32_t)LO, inc, M);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (p_u <= (uint32_t)MID) return edit_distance_unit_dispatch<LO, MID>(p_u, sent, n1, pat, inc, M);
    return edit_distance_unit_dispatch<MID + 1, HI>(p_u, sent, n1, pat, inc, M);
  }
}

// Dispatch on the wave's common pattern length: the variant compiled for exactly that length.
template <int LO, int HI, bool IDF, bool REAL = false>
__device__ __forceinline__ float edit_distance_exact_dispatch(uint32_t p_u, const uint32_t* __restrict__ sent, uint32_t n1,
                                                              const uint32_t* __restrict__ pat, float ins, float del,
                                                              float rep, float dw, float& M, const float* p_idf,
                                                              float idf_weight, const uint32_t* __restrict__ s_real = nullptr,
                                                              const uint32_t* __restrict__ p_real = nullptr) {
  if constexpr (LO == HI) {
    return edit_distance_reg<LO, IDF, true, REAL>(sent, n1, pat, (uint32_t)LO, ins, del, rep, dw, M, p_idf, idf_weight, s_real, p_real);
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (p_u <= (uint32_t)MID)
      return edit_distance_exact_dispatch<LO, MID, IDF, REAL>(p_u, sent, n1, pat, ins, del, rep, dw, M, p_idf, idf_weight, s_real, p_real);
    return edit_distance_exact_dispatch<MID + 1, HI, IDF, REAL>(p_u, sent, n1, pat, ins, del, rep, dw, M, p_idf, idf_weight, s_real, p_real);
  }
}

// UNIT: insert = delete = replace >= 0 and no idf term (decided by the host): bit-parallel distance
// REAL (round 6): the index carries real tokens (ix.real) and the batch its patterns' (b.q_real): the float recurrence with the
// real-token term, rows and both pattern streams in registers.  A pair that carries a penalty token on either side (the index has
// interned penalty strings or the batch brings a q_itok array: `check_itok`) is not computed here: its list entry goes to `defer`
// (count in counters[kCtrDpDefer]) for k_dp<true, *>, the general kernel.
#ifndef FMX_DPREG_WAVES
#define FMX_DPREG_WAVES 8
#endif
// LO (round 6): class 1 is shared by k_dp_reg<48, ..> (33..48 tokens: no spills, fewer registers) and k_dp_reg<64, .., 49> — each
// walks the class's list and takes the pairs of its own lengths (the list is in pattern-length order: whole waves skip).
template <int P, bool IDF, bool UNIT = false, bool REAL = false, int LO = (P == 32 ? 1 : 33)>
__global__ void __launch_bounds__(kWave, (UNIT && P == 32) ? FMX_DPREG_WAVES : 1) k_dp_reg(DevIndex ix, DevBatch b, DevWork w, Params prm,
                                                                                          unsigned long long* defer = nullptr, int check_itok = 0,
                                                                                          const unsigned long long* ext_list = nullptr,
                                                                                          const unsigned long long* ext_n = nullptr) {
  const uint32_t lane = lane_id();
  unsigned long long j_lo, j_hi;
  dp_class_range(b, w, P == 32 ? 0u : 1u, j_lo, j_hi);
  // (32-bit loop state: the list holds fewer than 2^32 entries — the host keeps seg_cap below that —, and the candidate is
  // remembered by its slot, not by a pointer: the unit-cost variant for 32 tokens sits at the 64 registers of eight waves per SIMD)
  // ext_list: the pairs k_dp_tab<true> left to this kernel (k_dp_compact's list, *ext_n entries) instead of the length class
  const uint32_t n_list = ext_list ? (uint32_t)*ext_n : (uint32_t)(j_hi - j_lo);
  const unsigned long long* __restrict__ list = ext_list ? ext_list : w.c_qi + j_lo;
  for (uint32_t j = blockIdx.x * kWave + lane; j < n_list; j += gridDim.x * kWave) {
    // two dependency levels in front of the tokens: the list entry, then the query's offsets and the candidate record
    // together (the record names the sentence's token base and length: no detour through sent_pos and the length word)
    const unsigned long long qi = list[j];
    const uint32_t q = (uint32_t)(qi >> 32);
    const uint32_t slot = (uint32_t)qi;
    const Cand c0 = w.seg[slot];
    const uint32_t o0 = b.q_off[b.q_begin + q];
    const uint32_t p = b.q_off[b.q_begin + q + 1] - o0;
    const uint32_t* pat = b.q_tok + o0;
    if constexpr (P != 32 && !(LO == 33 && P == 64)) {
      if (p < (uint32_t)LO || p > (uint32_t)P) continue;
    }
    uint32_t tb = __float_as_uint(c0.score), slen = c0.longest >> 16;  // 0 = unknown (candidates rebuilt by k_cand*)
    if (tb == 0) { const uint32_t sp0 = ix.sent_pos[c0.sid]; slen = ix.tok[sp0]; tb = sp0 + 1; }
    const uint32_t* sent = ix.tok + tb;
    if constexpr (REAL) {
      if (check_itok) {  // penalty tokens sit in slots [0, n] of either side (slot n = the trailing one): any -> the general kernel
        uint32_t any = 0;
        if (ix.itok) for (uint32_t i = 0; i <= slen; i++) any |= ix.itok[tb - 1 + i];
        if (b.q_itok) for (uint32_t i = 0; i <= p; i++) any |= b.q_itok[o0 + (b.q_begin + q) + i];
        if (any) {
          defer[atomicAdd(&w.counters[kCtrDpDefer], 1ull)] = qi;
          continue;
        }
      }
    }
    const float dw = 100.f / normalizer(p, slen, prm.ins, prm.del, prm.rep);
    const float* p_idf = w.t_idf + (o0 - b.t_begin);
    const float idf_weight = (dw * prm.idf_penalty) / prm.idf_max;  // src/fuzzy_match.cc:591
    const uint32_t* s_real = REAL ? ix.real + tb : nullptr;
    const uint32_t* p_real = REAL ? b.q_real + o0 : nullptr;
    float M, final_cost;
    // the list is in pattern-length order: all active lanes of a wave normally share one length
    const uint32_t p_u = __builtin_amdgcn_readfirstlane(p);
    const bool same_p = __ballot(p != p_u) == 0ull;
    if constexpr (UNIT) {
      const float inc = prm.rep * dw;  // = ins * dw = del * dw
      if (same_p) final_cost = edit_distance_unit_dispatch<(P == 32 ? 1 : 33), P>(p_u, sent, slen, pat, inc, M);
      else final_cost = edit_distance_unit<P, false>(sent, slen, pat, p, inc, M);
    } else if (same_p)
      final_cost = edit_distance_exact_dispatch<LO, P, IDF, REAL>(p_u, sent, slen, pat, prm.ins, prm.del, prm.rep,
                                                                                dw, M, p_idf, idf_weight, s_real, p_real);
    else
      final_cost = edit_distance_reg<P, IDF, false, REAL>(sent, slen, pat, p, prm.ins, prm.del, prm.rep, dw, M, p_idf, idf_weight, s_real, p_real);
    Cand* cd = w.seg + slot;
    cd->score = final_cost;
    cd->penalty = M;
    cd->longest = (c0.longest & 0xffffu) | (slen << 16);
  }
}

// ---------------------------------------------------------------------------------------------
// k_dp_tab (round 5) — the unit-cost edit distances of the 32-token class with the match masks LOOKED UP instead of compared.
// k_dp_reg<32, false, true> spends three quarters of its instructions forming, for every row (sentence token), the mask of the
// pattern positions that hold the same word — two VALU instructions per pattern token.  The pattern is the QUERY's: the
// candidates of one query follow one another in the dense list, so the lanes of a wave belong to a handful of queries (runs of
// equal q).  Each run gets a 64-slot hash table in LDS (token -> mask of its positions in the pattern, filled by all lanes of the
// wave over the flat (run, position) pairs), and a row costs one probe (~12 instructions) whatever the pattern's length.  The
// distance itself is the same bit-parallel recurrence on the same masks (lev_step32) and the same float sum: same bits.
// Up to kDpTabRuns runs share the LDS at a time; a wave with more (many queries of one or two candidates) takes its runs in rounds.
// ---------------------------------------------------------------------------------------------
#ifndef FMX_DPTAB_RUNS
#define FMX_DPTAB_RUNS 16
#endif
#ifndef FMX_DPTAB_WAVES
#define FMX_DPTAB_WAVES 4
#endif
constexpr uint32_t kDpTabRuns = FMX_DPTAB_RUNS, kDpTabSlots = 64;
__device__ __forceinline__ uint32_t dp_tab_slot(uint32_t w) { return (w * 0x85EBCA6Bu) >> 26; }

// REAL (round 6): the index carries real tokens (the reference CLI's default -p nbr,tag,cas) and nothing carries penalty tokens.  A pair in
// which no equal word meets a different real token — two thirds of the pairs of such an index at C3 — costs exactly what the ids
// alone cost (every real-token term of src/edit_distance.cc:53-62 is zero), so it is answered here; the table also remembers, per
// pattern word, the real-token id its positions share (or that they differ), and a row whose sentence token finds its word with
// another real token marks the pair for the float kernel: flags[chunk] = the chunk's marked lanes, cnt[chunk] their number
// (k_dp_compact turns them into a dense list in the same order; chunks up to n_chunks inclusive are written: the scan's input).
constexpr uint32_t kRealNone = 0xfffffffeu, kRealMixed = 0xffffffffu;
template <bool REAL>
__global__ void __launch_bounds__(kWave, FMX_DPTAB_WAVES) k_dp_tab(DevIndex ix, DevBatch b, DevWork w, Params prm,
                                                                 unsigned long long* flags = nullptr, unsigned long long* cnt = nullptr,
                                                                 uint32_t n_chunks = 0) {
  // {token, mask of its positions in the run's pattern}: one 8-byte read per row; REAL: {.., the real-token id the token's positions
  // share / kRealMixed, -} in ONE 16-byte entry (a second LDS read behind the probe doubled the row's round trips: the marking kernel
  // took 1.09 ms against the plain one's 0.45) and eight runs per round instead of sixteen (the same 8 KB)
  using Ent = std::conditional_t<REAL, uint4, uint2>;
  constexpr uint32_t kRuns = REAL ? kDpTabRuns / 2 : kDpTabRuns;
  __shared__ Ent s_tab[kRuns][kDpTabSlots];
  __shared__ uint32_t s_rp[kRuns + 1];   // pattern tokens in front of each run of the round (exclusive prefix)
  __shared__ uint32_t s_ro0[kRuns];      // where each run's pattern starts in q_tok
  const uint32_t lane = lane_id();
  unsigned long long j_lo, j_hi;
  dp_class_range(b, w, 0u, j_lo, j_hi);
  const uint32_t n_list = (uint32_t)(j_hi - j_lo);
  const unsigned long long* __restrict__ list = w.c_qi + j_lo;
  // (REAL: cnt[] behind the list's last chunk was zeroed by the host — the scan reads n_chunks + 1 entries)
  for (uint32_t j0 = blockIdx.x * kWave; j0 < n_list; j0 += gridDim.x * kWave) {
    const uint32_t j = j0 + lane;
    const bool act = j < n_list;
    const unsigned long long qi = act ? list[j] : 0ull;
    const uint32_t q = (uint32_t)(qi >> 32);
    const uint32_t slot = (uint32_t)qi;
    Cand c0;
    c0.sid = 0; c0.longest = 0; c0.score = 0.f; c0.penalty = 0.f;
    uint32_t o0 = 0, p = 0;
    if (act) {
      c0 = w.seg[slot];
      o0 = b.q_off[b.q_begin + q];
      p = b.q_off[b.q_begin + q + 1] - o0;
    }
    uint32_t tb = __float_as_uint(c0.score), slen = c0.longest >> 16;  // 0 = unknown (candidates rebuilt by k_cand*)
    if (act && tb == 0) { const uint32_t sp0 = ix.sent_pos[c0.sid]; slen = ix.tok[sp0]; tb = sp0 + 1; }
    if (!act) slen = 0;
    const Tok4* sent4 = reinterpret_cast<const Tok4*>(ix.tok + tb);
    const Tok4* real4 = reinterpret_cast<const Tok4*>(REAL ? ix.real + tb : ix.tok);
    Tok4 nxt = slen ? sent4[0] : Tok4{{0u, 0u, 0u, 0u}};   // (requested before the tables are built: it travels meanwhile)
    Tok4 nxt_r = (REAL && slen) ? real4[0] : Tok4{{0u, 0u, 0u, 0u}};
    bool need = false;   // REAL: some equal word carries another real token: the float kernel computes this pair
    // runs of equal q: a run's number = heads in front of the lane
    const uint32_t pq = __shfl_up(q, 1, 64);
    const bool head = act && (lane == 0 || pq != q);
    const unsigned long long heads = __ballot(head);
    const uint32_t run = act ? (uint32_t)__popcll(heads & ((2ull << lane) - 1ull)) - 1u : 0xffffu;
    const uint32_t n_runs = (uint32_t)__popcll(heads);
    uint32_t pv = ~0u, mv = 0u, score = p;
    const uint32_t m_bits = p ? p : 1u;
    for (uint32_t rb = 0; rb < n_runs; rb += kRuns) {
      const uint32_t n_r = n_runs - rb < kRuns ? n_runs - rb : kRuns;  // runs of this round
      const uint32_t rl = run - rb;                                               // this lane's run inside the round
      const bool mine = act && rl < n_r;
      wave_lds_sync();
      // the round's tables: empty; the runs' pattern lengths and starts (written by each run's head lane)
      for (uint32_t k = lane; k < n_r * kDpTabSlots; k += kWave) {
        if constexpr (REAL) (&s_tab[0][0])[k] = make_uint4(kHashEmpty, 0u, kRealNone, 0u);
        else (&s_tab[0][0])[k] = make_uint2(kHashEmpty, 0u);
      }
      if (head && rl < n_r) { s_rp[rl + 1] = p; s_ro0[rl] = o0; }
      if (lane == 0) s_rp[0] = 0;
      wave_lds_sync();
      if (lane == 0) {  // (at most sixteen values: a serial prefix)
        uint32_t acc = 0;
        for (uint32_t r = 0; r < n_r; r++) { acc += s_rp[r + 1]; s_rp[r + 1] = acc; }
      }
      wave_lds_sync();
      // every (run, pattern position) pair of the round, 64 at a time: token -> its table, position -> its mask
      const uint32_t T = s_rp[n_r];
      for (uint32_t x = lane; x < T; x += kWave) {
        uint32_t r = 0;
#pragma unroll
        for (uint32_t step = kRuns / 2; step >= 1; step >>= 1)
          if (r + step < n_r && s_rp[r + step] <= x) r += step;
        const uint32_t jp = x - s_rp[r];
        const uint32_t t = b.q_tok[s_ro0[r] + jp];
        uint32_t h = dp_tab_slot(t);
        for (;;) {
          const uint32_t old = atomicCAS(&s_tab[r][h].x, kHashEmpty, t);
          if (old == kHashEmpty || old == t) { atomicOr(&s_tab[r][h].y, 1u << jp); break; }
          h = (h + 1) & (kDpTabSlots - 1);
        }
        if constexpr (REAL) {  // the first position's real token stays unless another position of the word brings a different one
          const uint32_t rr = b.q_real[s_ro0[r] + jp];
          const uint32_t was = atomicCAS(&s_tab[r][h].z, kRealNone, rr);
          if (was != kRealNone && was != rr) s_tab[r][h].z = kRealMixed;
        }
      }
      wave_lds_sync();
      // the rows: this lane's sentence against its run's table (one 8-byte read per row and probe step).
      // (Measured at C3 and left as they are: a branch-free row — home slot read unconditionally, state kept by selects, the probe
      // sequence only when some lane collides — 0.53 ms, the four probes of a group requested together 0.55 ms, this loop 0.50 ms.)
      const Ent* tab = &s_tab[mine ? rl : 0u][0];
      const uint32_t rows = mine ? slen : 0u;
      uint32_t rows_max = rows;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(rows_max, d, 64); rows_max = o > rows_max ? o : rows_max; }
      for (uint32_t i0 = 0; i0 < rows_max; i0 += 4) {
        const Tok4 cur = nxt, cur_r = nxt_r;
        if (i0 + 4 < rows) {
          nxt = sent4[(i0 >> 2) + 1];
          if (REAL) nxt_r = real4[(i0 >> 2) + 1];
        }
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
          if (i0 + u < rows) {
            const uint32_t wd = cur.t[u];
            uint32_t h = dp_tab_slot(wd);
            Ent e = tab[h];
            while (e.x != wd && e.x != kHashEmpty) { h = (h + 1) & (kDpTabSlots - 1); e = tab[h]; }
            if constexpr (REAL) need = need || (e.x == wd && e.z != (cur_r.t[u] & 0x7fffffffu));
            lev_step32(pv, mv, score, e.x == wd ? e.y : 0u, m_bits);
          }
        }
      }
    }
    if (REAL) {
      const unsigned long long fl = __ballot(act && need);
      if (lane == 0) { flags[j0 >> 6] = fl; cnt[j0 >> 6] = (unsigned long long)__popcll(fl); }
    }
    if (act && !need) {
      const float dw = 100.f / normalizer(p, slen, prm.ins, prm.del, prm.rep);
      const float f = unit_cost_sum(score, prm.rep * dw);  // = ins * dw = del * dw
      Cand* cd = w.seg + slot;
      cd->score = f;
      cd->penalty = f;   // M = final: row minima never decrease for non-negative unit costs (fmx_core.hh)
      cd->longest = (c0.longest & 0xffffu) | (slen << 16);
    }
  }
}

// k_dp_compact — the pairs k_dp_tab<true> marked, as a dense list in list order (so that the waves of the float kernel keep
// sharing one pattern length): offs = the exclusive scan of cnt, offs[n_chunks + 1] = their number.
__global__ void __launch_bounds__(256) k_dp_compact(DevBatch b, DevWork w, const unsigned long long* flags, const unsigned long long* offs,
                                                    unsigned long long* out) {
  unsigned long long j_lo, j_hi;
  dp_class_range(b, w, 0u, j_lo, j_hi);
  const uint32_t n_list = (uint32_t)(j_hi - j_lo);
  const unsigned long long* __restrict__ list = w.c_qi + j_lo;
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_list; j += gridDim.x * blockDim.x) {
    const unsigned long long fl = flags[j >> 6];
    const uint32_t bit = j & 63u;
    if ((fl >> bit) & 1ull) out[offs[j >> 6] + (unsigned long long)__popcll(fl & ((1ull << bit) - 1ull))] = list[j];
  }
}

// ---------------------------------------------------------------------------------------------
// k_finish — one wave per query.  Lane 0 replays the candidate loop's order-dependent acceptance
// (lowest_costs bound, src/fuzzy_match.cc:587-610) over the candidates in reference order, 64 staged
// through LDS at a time; accepted matches overwrite the head of the query's segment.  Then the
// `result` priority queue's pop order (score desc, s_id asc; CompareMatch src/fuzzy_match.cc:25-33)
// via a key/value sort, and either the plain top-N (:670-679) or the contrastive rerank (:614-669)
// with the ids-only edit distance (src/edit_distance.cc:79-122), one remaining candidate per lane
// per round.  Up to kLdsSortMax accepted matches live in LDS; more use the global scratch.
// Dynamic LDS (contrast only): rows[(max_slen+1)*64] floats.
// ---------------------------------------------------------------------------------------------
struct Rec {  // == fmx_match
  float    score;
  float    penalty;
  int32_t  max_subseq;
  uint32_t s_id;
};

constexpr uint32_t kMaxFlatRounds = 16;       // flattened contrastive rerank: rounds (= nmatch) it is launched for
constexpr uint32_t kMaxFlatAccepted = 65535;  // ... and accepted matches per query it indexes (16 bits of the bid)

struct DevScratch {
  unsigned long long* keys;   // 2 * seg_cap
  uint32_t*           vals;   // 2 * seg_cap
  float*              f0;     // seg_cap
  float*              f1;     // seg_cap
};

struct ReplayShared {
  float    final_cost[kWave], M[kWave], score[kWave];
  uint32_t sid[kWave], L[kWave], slen[kWave];
  float    ub;
};

__device__ __forceinline__ uint32_t replay_body(const Params& prm, uint32_t p, uint32_t n_cand, Cand* seg,
                                                float* heap_store, ReplayShared& sh, uint32_t& max_slen_out) {
  constexpr int kAhead = 8;  // chunks whose candidates are requested together (the loop is a latency chain)
  uint32_t max_slen = 0;
  const uint32_t lane = lane_id();
  CostHeap heap;
  heap.v = heap_store;
  heap.n = 0;
  if (lane == 0) heap.push(FLT_MAX);
  uint32_t acc = 0;
  if (lane == 0) sh.ub = FLT_MAX;
  for (uint32_t g0 = 0; g0 < n_cand; g0 += kWave * kAhead) {
    Cand cds[kAhead];
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      const uint32_t c = g0 + u * kWave + lane;
      cds[u].sid = 0; cds[u].longest = 0; cds[u].score = 0.f; cds[u].penalty = 0.f;
      if (c < n_cand) cds[u] = seg[c];
    }
#pragma unroll
    for (int u = 0; u < kAhead; u++) {
      const uint32_t c0 = g0 + u * kWave;
      if (c0 >= n_cand) break;
      const uint32_t c = c0 + lane;
      __syncthreads();
      // The bound never increases, so a candidate the bound left by the previous chunk already rejects
      // (or the no_perfect rule) cannot change the state: all 64 lanes test that in parallel and lane 0
      // only walks the few that may still be accepted, in order.  (Accepted matches are written to
      // seg[acc++], acc <= c0: never ahead of the chunks already held in registers.)
      bool maybe = false;
      if (c < n_cand) {
        const Cand cd = cds[u];
        const uint32_t slen = cd.longest >> 16;
        sh.final_cost[lane] = cd.score; sh.M[lane] = cd.penalty; sh.sid[lane] = cd.sid;
        sh.L[lane] = cd.longest & 0xffffu; sh.slen[lane] = slen;
        maybe = !((prm.no_perfect && cd.score == 0.f && slen == p) || cd.penalty > sh.ub);
        if (maybe) sh.score[lane] = score_from_cost(cd.score);
      }
      unsigned long long todo = __ballot(maybe);
      __syncthreads();
      if (lane == 0) {
        while (todo) {
          const uint32_t k = (uint32_t)__ffsll((long long)todo) - 1u;
          todo &= todo - 1ull;
          const float score = sh.score[k];
          if (replay_candidate_scored(heap, prm, p, sh.slen[k], sh.final_cost[k], sh.M[k], score)) {
            Cand a;
            a.sid = sh.sid[k]; a.longest = sh.L[k]; a.score = score; a.penalty = 0.f;
            seg[acc++] = a;
            max_slen = sh.slen[k] > max_slen ? sh.slen[k] : max_slen;
          }
        }
        sh.ub = heap.top();
      }
    }
  }
  __syncthreads();
  max_slen_out = __shfl(max_slen, 0, 64);
  return __shfl(acc, 0, 64);
}

// The same replay for a candidate set that fits one wave (the common case), without the detour through
// global memory: accepted candidates are flagged in a mask and their scores replace the costs in LDS.
// Single-wave workgroup: LDS ordering by wave_lds_sync, no barriers.
__device__ __forceinline__ unsigned long long replay_small(const Params& prm, uint32_t p, uint32_t n_cand, const Cand* seg,
                                                           ReplayShared& sh) {
  const uint32_t lane = lane_id();
  bool maybe = false;
  float my_cost = 0.f, my_M = 0.f, my_score = 0.f;
  if (lane < n_cand) {
    const Cand cd = seg[lane];
    const uint32_t slen = cd.longest >> 16;
    my_cost = cd.score; my_M = cd.penalty; my_score = score_from_cost(cd.score);
    sh.sid[lane] = cd.sid; sh.L[lane] = cd.longest & 0xffffu; sh.score[lane] = my_score;  // for the ranking afterwards
    maybe = !(prm.no_perfect && cd.score == 0.f && slen == p);
  }
  unsigned long long todo = __ballot(maybe);
  // lowest_costs as a sorted array spread over the lanes (lane i holds v[i], CostHeap's layout): a turn of the
  // candidate loop is a handful of cross-lane operations on wave-uniform values, with no LDS round trips.
  float hv = FLT_MAX;  // v[0] = FLT_MAX, the reference's initial bound
  uint32_t n = 1;
  unsigned long long acc = 0;
  while (todo) {
    const int k = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    const float fc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_cost), k));
    const float Mk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_M), k));
    const float sc = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_score), k));
    const float ub = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), (int)(n - 1)));
    if (Mk > ub) continue;  // replay_candidate_scored; the no_perfect rule is already in `todo`
    // push(fc): behind the elements <= fc, the larger ones move up one lane
    const uint32_t pos = (uint32_t)__popcll(__ballot(lane < n && hv <= fc));
    const float up = __shfl_up(hv, 1, 64);
    hv = lane == pos ? fc : ((lane > pos && lane <= n) ? up : hv);
    n++;
    if (sc < prm.fuzzy || (prm.cbuf > 0 && n > (uint32_t)prm.cbuf)) n--;  // pop: the largest goes
    if (sc >= prm.fuzzy) acc |= 1ull << k;
  }
  wave_lds_sync();
  return acc;
}

template <int MODE>
__device__ __forceinline__ void select_body(const DevIndex& ix, const Params& prm, const Cand* seg, uint32_t n,
                                            uint32_t np2, unsigned long long* keys, uint32_t* vals, float* pen_sum,
                                            float* pen_cur, float* s_rows, uint32_t cap, Rec* out,
                                            uint32_t& out_count, uint32_t& out_total, const uint32_t* prior_off = nullptr,
                                            const uint32_t* prior_tok = nullptr, uint32_t n_prior = 0, const uint32_t* prior_sid = nullptr) {
  const uint32_t lane = lane_id();
  const uint32_t want = prm.nmatch == 0 ? n : (prm.nmatch < n ? prm.nmatch : n);
  const uint32_t emit = want < cap ? want : cap;
  if (MODE < 0 || !(prm.contrast > 0.f)) {
    // ascending key order = score desc, s_id asc; value = position in the accepted list
    for (uint32_t i = lane; i < np2; i += kWave) {
      keys[i] = i < n ? (((unsigned long long)(~float_key(seg[i].score)) << 32) | seg[i].sid) : ~0ull;
      vals[i] = i;
    }
    __syncthreads();
    block_bitonic_sort_kv(keys, vals, np2);
    for (uint32_t k = lane; k < emit; k += kWave) {
      const Cand c = seg[vals[k]];
      Rec r;
      r.score = c.score; r.penalty = 0.f; r.max_subseq = (int32_t)c.longest; r.s_id = c.sid;
      out[k] = r;
    }
    out_count = emit;
    out_total = want;
    return;
  }
  // ---- contrastive rerank ----
  // The reference drains `result` into a list ordered (score desc, s_id asc) and repeatedly takes the
  // first maximum of score - factor * penalty.  "First in list order" among equal keys is the entry with
  // the higher score, then the smaller s_id, so the list never has to be materialised: the accepted
  // matches stay where the replay left them and ties are broken on (score, s_id) directly.
  // pen_sum: running std::accumulate of penalties in pick order; pen_cur: Match::penalty (0 before the
  // first round); keys: (taken << 63) | bits(running max)
  (void)vals;
  for (uint32_t k = lane; k < n; k += kWave) { pen_sum[k] = 0.f; pen_cur[k] = 0.f; keys[k] = 0ull; }
  __syncthreads();
  DpSide side;
  side.s_real = nullptr; side.p_real = nullptr; side.s_itok = nullptr; side.p_itok = nullptr;
  side.itok_dist = nullptr; side.n_itok = 0; side.p_idf = nullptr; side.idf_weight = 0.f;
  uint32_t picked = 0;
  int last = -1;
  // Entries already in the caller's `matches` (src/fuzzy_match.cc:622-650: "for (Match &match_memory : matches)" runs over
  // them first, in order, in every round): their penalties are folded in before the first pick, the running sum / maximum
  // continue over the picks.  Only the general kernels (MODE >= 2: float DP, rows in LDS / HBM) take priors; k_finish queues
  // every query for them when the batch has any.
  uint32_t n_mem = 0;  // penalties per candidate so far (priors + picks)
  if (MODE >= 2) {
    for (uint32_t r = 0; r < n_prior; r++) {
      const uint32_t* s2 = prior_tok + prior_off[r];
      const uint32_t n2 = prior_off[r + 1] - prior_off[r];
      n_mem++;
      for (uint32_t k = lane; k < n; k += kWave) {
        const uint32_t csp = ix.sent_pos[seg[k].sid];
        const uint32_t n1 = ix.tok[csp];
        const float dw = 100.f / normalizer(n1, n2, 1.f, 1.f, 1.f);
        float M;
        const float ed = edit_distance<false>(ix.tok + csp + 1, n1, s2, n2, 1.f, 1.f, 1.f, dw, side, FLT_MAX, LdsRow{s_rows + lane}, M);
        const float pen = score_from_cost(ed);
        const float sum = pen_sum[k] + pen;
        pen_sum[k] = sum;
        float mx = __uint_as_float((uint32_t)(keys[k] & 0xffffffffull));
        mx = (n_mem == 1) ? pen : (mx < pen ? pen : mx);
        keys[k] = (unsigned long long)__float_as_uint(mx);
        pen_cur[k] = prm.reduce == 1 ? mx : (sum / (float)n_mem);
      }
      __syncthreads();
    }
  }
  while (picked < n && (prm.nmatch == 0 || picked < prm.nmatch)) {
    if (last >= 0) {
      const uint32_t msid = seg[last].sid;
      const uint32_t msp = ix.sent_pos[msid];
      uint32_t n2 = ix.tok[msp];
      const uint32_t* s2 = ix.tok + msp + 1;
      // The reference memoises a pair's cost under the two S_IDS (src/fuzzy_match.cc:627, 637-650): a pick whose id an entry
      // already in the caller's vector carries — a match of another translation memory — is seen by every candidate through
      // the cost computed for that entry, i.e. on ITS tokens.  (Prior entries that share an id among themselves were resolved
      // on the host: fmx_match_one_prior_ids.)
      if (MODE >= 2 && prior_sid != nullptr)
        for (uint32_t r = 0; r < n_prior; r++)
          if (prior_sid[r] == msid) { s2 = prior_tok + prior_off[r]; n2 = prior_off[r + 1] - prior_off[r]; break; }
      n_mem++;
      for (uint32_t k = lane; k < n; k += kWave) {
        if (keys[k] >> 63) continue;
        const uint32_t csid = seg[k].sid;
        const uint32_t csp = ix.sent_pos[csid];
        const uint32_t n1 = ix.tok[csp];
        const float dw = 100.f / normalizer(n1, n2, 1.f, 1.f, 1.f);
        float M, ed;
        if (MODE == 0) ed = edit_distance_unit<32, false>(ix.tok + csp + 1, n1, s2, n2, 1.f * dw, M);
        else if (MODE == 1) ed = edit_distance_unit<64, false>(ix.tok + csp + 1, n1, s2, n2, 1.f * dw, M);
        else ed = edit_distance<false>(ix.tok + csp + 1, n1, s2, n2, 1.f, 1.f, 1.f, dw, side, FLT_MAX,
                                       LdsRow{s_rows + lane}, M);
        const float pen = score_from_cost(ed);
        const float sum = pen_sum[k] + pen;
        pen_sum[k] = sum;
        float mx = __uint_as_float((uint32_t)(keys[k] & 0xffffffffull));
        mx = (n_mem == 1) ? pen : (mx < pen ? pen : mx);
        keys[k] = (unsigned long long)__float_as_uint(mx);
        pen_cur[k] = prm.reduce == 1 ? mx : (sum / (float)n_mem);
      }
      __syncthreads();
    }
    // std::max_element over the remaining list: first maximum of score - factor * penalty
    float best_key = 0.f, best_score = 0.f;
    uint32_t best_sid = 0;
    int best = -1;
    auto better = [](float key, float score, uint32_t sid, float bkey, float bscore, uint32_t bsid) {
      return bkey < key || (bkey == key && (bscore < score || (bscore == score && sid < bsid)));
    };
    for (uint32_t k = lane; k < n; k += kWave) {
      if (keys[k] >> 63) continue;
      const Cand c = seg[k];
      const float key = c.score - (prm.contrast * pen_cur[k]);
      if (best < 0 || better(key, c.score, c.sid, best_key, best_score, best_sid)) {
        best = (int)k; best_key = key; best_score = c.score; best_sid = c.sid;
      }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const float ok = __shfl_xor(best_key, d, 64);
      const float os = __shfl_xor(best_score, d, 64);
      const uint32_t oi = __shfl_xor(best_sid, d, 64);
      const int   ob = __shfl_xor(best, d, 64);
      if (ob >= 0 && (best < 0 || better(ok, os, oi, best_key, best_score, best_sid))) {
        best = ob; best_key = ok; best_score = os; best_sid = oi;
      }
    }
    if (lane == 0) {
      if (picked < cap) {
        const Cand c = seg[best];
        Rec r;
        r.score = c.score; r.penalty = pen_cur[best]; r.max_subseq = (int32_t)c.longest; r.s_id = c.sid;
        out[picked] = r;
      }
      keys[best] |= 1ull << 63;
    }
    picked++;
    last = best;
    __syncthreads();
  }
  out_count = picked < cap ? picked : cap;
  out_total = picked;
}

// LIST = false: one wave per query of the sub-batch, handles the queries with at most 64 candidates (tiny
// LDS footprint -> full occupancy) and queues the others; LIST = true: grid-stride over that queue with
// the large LDS sorter.
template <uint32_t SORT_CAP, bool LIST>
__global__ void __launch_bounds__(kWave) k_finish(DevIndex ix, DevBatch b, DevWork w, Params prm, DevScratch sc,
                                                  uint32_t cap, Rec* out_rec, uint32_t* out_count,
                                                  uint32_t* out_total, uint32_t* contrast_list, uint32_t* fin_list) {
  __shared__ unsigned long long s_keys[SORT_CAP];
  __shared__ uint32_t s_vals[SORT_CAP];
  __shared__ ReplayShared sh;
  __shared__ float s_heap[kHeapLdsMax + 2];
  const uint32_t lane = lane_id();
  const bool contrast = prm.contrast > 0.f;
  const unsigned long long n_items = LIST ? w.counters[14] : (unsigned long long)b.nq;
  for (unsigned long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint32_t q = LIST ? fin_list[item] : (uint32_t)item;
    const uint32_t gq = b.q_begin + q;
    const uint32_t n_cand = w.q_cand[q];
    if (!LIST && n_cand > kWave) continue;  // queued by k_cmap for the list kernel (large sorter / multi-chunk replay)
    Rec* out = out_rec + (unsigned long long)gq * cap;
    uint32_t cnt = 0, tot = 0;
    if (!LIST && !contrast && n_cand != 0 && prm.cbuf > 0 && prm.cbuf <= (int)kHeapLdsMax) {
      // one wave holds the whole candidate set: replay, rank and emit from registers / LDS
      const unsigned long long seg0 = uni(w.q_seg[q]);
      const uint32_t p = uni(b.q_off[gq + 1]) - uni(b.q_off[gq]);
      const unsigned long long acc = replay_small(prm, p, n_cand, w.seg + seg0, sh);
      const uint32_t n = (uint32_t)__popcll(acc);
      const uint32_t want = prm.nmatch == 0 ? n : (prm.nmatch < n ? prm.nmatch : n);
      const uint32_t emit = want < cap ? want : cap;
      const bool mine = (acc >> lane) & 1ull;
      // ascending key = score desc, s_id asc (sentence ids are unique within a candidate set)
      unsigned long long key = mine ? (((unsigned long long)(~float_key(sh.score[lane])) << 32) | sh.sid[lane]) : ~0ull;
      uint32_t idx = lane;
      wave_sort64_kv_upto(key, idx, n_cand);
      if (lane < emit) {
        Rec r;
        r.score = sh.score[idx]; r.penalty = 0.f; r.max_subseq = (int32_t)sh.L[idx]; r.s_id = sh.sid[idx];
        out[lane] = r;
      }
      if (lane == 0) { w.q_acc[q] = n; out_count[gq] = emit; if (out_total) out_total[gq] = want; }
      wave_lds_sync();
      continue;
    }
    if (n_cand != 0) {
      const unsigned long long seg0 = w.q_seg[q];
      Cand* seg = w.seg + seg0;
      const uint32_t p = b.q_off[gq + 1] - b.q_off[gq];
      uint32_t n, mx;
      if (!LIST && n_cand <= kWave && prm.cbuf > 0 && prm.cbuf <= (int)kHeapLdsMax) {
        // (contrastive rerank with a candidate set of one wave) the replay spread over the lanes, like the plain path; the
        // accepted matches are then compacted to the head of the segment in candidate order, as replay_body leaves them
        const uint32_t my_slen = lane < n_cand ? (seg[lane].longest >> 16) : 0u;
        const unsigned long long acc = replay_small(prm, p, n_cand, seg, sh);
        const bool mine = (acc >> lane) & 1ull;
        n = (uint32_t)__popcll(acc);
        uint32_t m2 = mine ? my_slen : 0u;
#pragma unroll
        for (int d2 = 32; d2 >= 1; d2 >>= 1) { const uint32_t o = __shfl_xor(m2, d2, 64); m2 = o > m2 ? o : m2; }
        mx = m2;
        if (mine) {
          Cand a;
          a.sid = sh.sid[lane]; a.longest = sh.L[lane]; a.score = sh.score[lane]; a.penalty = 0.f;
          seg[__popcll(acc & ((1ull << lane) - 1ull))] = a;  // every lane read its own candidate before (replay_small)
        }
        __syncthreads();
      } else if (prm.cbuf > 0 && prm.cbuf <= (int)kHeapLdsMax) n = replay_body(prm, p, n_cand, seg, s_heap, sh, mx);
      else n = replay_body(prm, p, n_cand, seg, w.heap + 2ull * seg0, sh, mx);  // n_cand + 1 <= 2 x the segment's capacity; segments are not in query order
      if (lane == 0) {
        w.q_acc[q] = n;
        // contrastive rerank: queries whose accepted sentences are longer than 32 / 64 tokens are queued for the
        // wider kernels; everything else is done by k_contrast<0> in place
        if (contrast && n != 0 && (mx > 32 || n > kMaxFlatAccepted || b.n_prior != 0)) {
          // (prior matches of the caller's vector, any length: only the general kernel folds them in)
          const uint32_t cls = (mx <= 64 && n <= kMaxFlatAccepted && b.n_prior == 0) ? 1u : 2u;  // the flattened rerank indexes a query's matches with 16 bits
          // (the flattened rerank finds the queries of class 1 through their class bits: only the per-query kernels read the
          // list, and a same-address atomic per query is what this kernel would then spend its time on)
          const bool flat = prm.nmatch >= 1 && prm.nmatch <= kMaxFlatRounds;
          if (cls == 2 || !flat) contrast_list[(size_t)(cls - 1) * b.nq + atomicAdd(&w.counters[9 + cls], 1ull)] = q;
          w.q_acc[q] = n | (cls << 30);  // length class of the accepted sentences (0: <= 32, 1: <= 64, 2: longer)
        }
      }
      if (n != 0 && !contrast) {
        const uint32_t np2 = next_pow2(n);
        if (np2 <= SORT_CAP) select_body<-1>(ix, prm, seg, n, np2, s_keys, s_vals, nullptr, nullptr, nullptr, cap, out, cnt, tot);
        else select_body<-1>(ix, prm, seg, n, np2, sc.keys + 2ull * seg0, sc.vals + 2ull * seg0, nullptr, nullptr, nullptr,
                             cap, out, cnt, tot);
      }
    } else if (lane == 0)
      w.q_acc[q] = 0;
    // with a contrastive factor the accepted list is handed to the rerank kernels, which write the output
    if (lane == 0 && !(contrast && n_cand != 0)) { out_count[gq] = cnt; if (out_total) out_total[gq] = tot; }
    __syncthreads();
  }
}

// k_contrast<MODE> — contrastive rerank (src/fuzzy_match.cc:614-669) of the accepted matches k_finish left in
// the query's segment.  Three launches partition the queries by their longest accepted sentence (MODE 0:
// <= 32 tokens, 1: <= 64, 2: longer), so the common case keeps the picked sentence and the DP row in registers.
constexpr uint32_t kContrastLds = 256;

// MODE 3 = MODE 2 with the DP rows in HBM (g_rows: row_len x 64 floats per workgroup): indexed sentences too long for
// the LDS rows.
// k_finish_pair — two queries per wave (lanes 0..31 and 32..63) for the common case: at most kPairMax candidates each, no
// contrastive rerank, lowest_costs of at most kPairMax - 2 entries.  The replay of the candidate loop is an instruction
// stream per query whatever the number of candidates (the kernel is bound by VALU issue): sharing each instruction between
// two queries halves it.  Same steps as k_finish's one-wave path: lowest_costs as a sorted array spread over the half's
// lanes, cross-lane reads by ds_bpermute instead of v_readlane (the two halves walk different candidates).
constexpr uint32_t kPairMax = 32;
// wave_sort64_kv_n for the two halves of a wave at once: every half sorts ascending (the direction of the last merge stage
// comes from the lane number inside the half; wave_sort64_kv_n<32> would sort lanes 32..63 descending)
template <int N>
__device__ __forceinline__ void half_sort64_kv_n(unsigned long long& key, uint32_t& val, uint32_t hl) {  // hl: the lane inside its group
#pragma unroll
  for (int k = 2; k <= N; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const uint32_t lo = __shfl_xor((uint32_t)key, j, 64), hi = __shfl_xor((uint32_t)(key >> 32), j, 64);
      const uint32_t ov = __shfl_xor(val, j, 64);
      const unsigned long long other = ((unsigned long long)hi << 32) | lo;
      const bool take_min = ((hl & k) == 0) == ((hl & j) == 0);
      const bool other_less = other < key;  // keys are distinct (or both ~0): no tie handling needed
      const bool take_other = take_min ? other_less : !other_less && other != key;
      key = take_other ? other : key;
      val = take_other ? ov : val;
    }
  }
}
// pair_max: 32 candidates per query while lowest_costs is capped below 32 entries (contrast_buffer <= 30), else 31 — the bound
// list then never holds more than the candidates + 1 = 32 entries, one per lane of the half.  With a contrastive factor the
// accepted matches are not ranked and emitted but compacted to the head of the segment for the rerank kernels, like k_finish does.
// Round 5: the body is written for a GROUP of W lanes per query — W = 32: two queries per wave, W = 16: FOUR (candidate sets of
// up to 16 / 15 entries: most of them) — and a wave takes four consecutive queries: all four in one go when none of them holds
// more than a quarter wave's worth, else two and two.  The replay's instruction stream is shared by twice as many queries.
template <int W>
__device__ __forceinline__ void finish_group(const DevBatch& b, const DevWork& w, const Params& prm, uint32_t cap, Rec* out_rec,
                                             uint32_t* out_count, uint32_t* out_total, uint32_t grp_max, uint32_t pair_max,
                                             uint32_t* contrast_list, uint32_t q_first, float* s_score, uint32_t* s_sid, uint32_t* s_L) {
  constexpr uint32_t kMaskW = W == 32 ? 0xffffffffu : ((1u << (W & 31)) - 1u);
  const uint32_t lane = lane_id(), hl = lane & (uint32_t)(W - 1), base = lane & ~(uint32_t)(W - 1);
  const uint32_t q = q_first + lane / (uint32_t)W;
  const bool have = q < b.nq;
  const uint32_t gq = b.q_begin + (have ? q : 0u);
  const uint32_t n_cand = have ? w.q_cand[q] : 0u;
  const bool act_q = have && n_cand != 0 && n_cand <= grp_max;  // larger sets: the wider group, or queued by k_cmap for the list kernel
  const bool contrast = prm.contrast > 0.f;
  uint32_t p = 0, my_slen = 0;
  bool maybe = false;
  float my_cost = 0.f, my_M = 0.f, my_score = 0.f;
  if (act_q) p = b.q_off[gq + 1] - b.q_off[gq];
  wave_lds_sync();
  if (act_q && hl < n_cand) {
    const Cand cd = w.seg[w.q_seg[q] + hl];
    const uint32_t slen = cd.longest >> 16;
    my_slen = slen;
    my_cost = cd.score; my_M = cd.penalty; my_score = score_from_cost(cd.score);
    s_sid[lane] = cd.sid; s_L[lane] = cd.longest & 0xffffu; s_score[lane] = my_score;
    maybe = !(prm.no_perfect && cd.score == 0.f && slen == p);
  }
  uint32_t todo = (uint32_t)(__ballot(maybe) >> base) & kMaskW;
  float hv = FLT_MAX;  // v[0] = FLT_MAX, the reference's initial bound; lane hl of the group holds v[hl]
  uint32_t n = 1, acc = 0;
  while (__ballot(todo != 0)) {
    const bool on = todo != 0;
    const uint32_t k = on ? (uint32_t)__builtin_ctz(todo) : 0u;
    todo &= todo - 1u;  // 0 stays 0
    const float fc = __shfl(my_cost, (int)(base + k), 64);
    const float Mk = __shfl(my_M, (int)(base + k), 64);
    const float sc = __shfl(my_score, (int)(base + k), 64);
    const float ub = __shfl(hv, (int)(base + n - 1u), 64);
    const bool go = on && !(Mk > ub);  // replay_candidate_scored; the no_perfect rule is already in `todo`
    // push(fc): behind the elements <= fc, the larger ones move up one lane
    const uint32_t pos = (uint32_t)__popc((uint32_t)(__ballot(go && hl < n && hv <= fc) >> base) & kMaskW);
    const float up = __shfl_up(hv, 1, 64);
    if (go) {
      hv = hl == pos ? fc : ((hl > pos && hl <= n) ? up : hv);
      n++;
      if (sc < prm.fuzzy || (prm.cbuf > 0 && n > (uint32_t)prm.cbuf)) n--;  // pop: the largest goes
      if (sc >= prm.fuzzy) acc |= 1u << k;
    }
  }
  const uint32_t n_acc = (uint32_t)__popc(acc);
  if (contrast) {  // wave-uniform: the accepted matches to the head of the segment in candidate order, the class of the rerank
    const bool mine_c = act_q && ((acc >> hl) & 1u);
    uint32_t mx = mine_c ? my_slen : 0u;
#pragma unroll
    for (int d2 = W / 2; d2 >= 1; d2 >>= 1) { const uint32_t o = __shfl_xor(mx, d2, 64); mx = o > mx ? o : mx; }  // within the group
    wave_lds_sync();
    if (mine_c) {
      Cand a;
      a.sid = s_sid[lane]; a.longest = s_L[lane]; a.score = my_score; a.penalty = 0.f;
      w.seg[w.q_seg[q] + (uint32_t)__popc(acc & ((1u << hl) - 1u))] = a;  // every lane read its own candidate at the top
    }
    if (have && hl == 0 && n_cand <= grp_max) {
      uint32_t qa = act_q ? n_acc : 0u;
      if (act_q && n_acc != 0 && (mx > 32 || n_acc > kMaxFlatAccepted || b.n_prior != 0)) {
        const uint32_t cls = (mx <= 64 && n_acc <= kMaxFlatAccepted && b.n_prior == 0) ? 1u : 2u;
        const bool flat = prm.nmatch >= 1 && prm.nmatch <= kMaxFlatRounds;
        if (cls == 2 || !flat) contrast_list[(size_t)(cls - 1) * b.nq + atomicAdd(&w.counters[9 + cls], 1ull)] = q;
        qa |= cls << 30;
      }
      w.q_acc[q] = qa;
      if (!act_q) { out_count[gq] = 0u; if (out_total) out_total[gq] = 0u; }  // no candidates: nothing for the rerank to write
    }
    return;
  }
  const uint32_t want = prm.nmatch == 0 ? n_acc : (prm.nmatch < n_acc ? prm.nmatch : n_acc);
  const uint32_t emit = want < cap ? want : cap;
  const bool mine = act_q && ((acc >> hl) & 1u);
  // ascending key = score desc, s_id asc (sentence ids are unique within a candidate set); each group is sorted on its own
  unsigned long long key = mine ? (((unsigned long long)(~float_key(my_score)) << 32) | s_sid[lane]) : ~0ull;
  uint32_t idx = hl;
  uint32_t n_max = n_cand <= grp_max ? n_cand : 0u;
#pragma unroll
  for (int d2 = 32; d2 >= W; d2 >>= 1) { const uint32_t o = __shfl_xor(n_max, d2, 64); n_max = o > n_max ? o : n_max; }
  n_max = __builtin_amdgcn_readfirstlane(n_max);
  if (W == 32 && n_max > 16) half_sort64_kv_n<32>(key, idx, hl);
  else if (n_max > 8) half_sort64_kv_n<16>(key, idx, hl);
  else if (n_max > 1) half_sort64_kv_n<8>(key, idx, hl);
  wave_lds_sync();
  if (act_q && hl < emit) {
    Rec r;
    r.score = s_score[base + idx]; r.penalty = 0.f; r.max_subseq = (int32_t)s_L[base + idx]; r.s_id = s_sid[base + idx];
    out_rec[(unsigned long long)gq * cap + hl] = r;
  }
  if (have && hl == 0 && n_cand <= grp_max) {
    w.q_acc[q] = act_q ? n_acc : 0u;
    out_count[gq] = act_q ? emit : 0u;
    if (out_total) out_total[gq] = act_q ? want : 0u;
  }
}

__global__ void __launch_bounds__(kWave) k_finish_pair(DevBatch b, DevWork w, Params prm, uint32_t cap, Rec* out_rec,
                                                       uint32_t* out_count, uint32_t* out_total, uint32_t pair_max,
                                                       uint32_t* contrast_list) {
  __shared__ float s_score[kWave];
  __shared__ uint32_t s_sid[kWave], s_L[kWave];
  const uint32_t q4 = blockIdx.x * 4u;
  // a quarter wave per query: lowest_costs (contrast_buffer + 1 entries at most) and the candidates + 1 must fit sixteen lanes
  const uint32_t quad_max = (prm.cbuf > 0 && prm.cbuf <= 14) ? 16u : 15u;
  const uint32_t qx = q4 + (lane_id() >> 4);
  const uint32_t nc = qx < b.nq ? w.q_cand[qx] : 0u;
  const bool wide = nc > quad_max && nc <= pair_max;   // needs half a wave
  if (__ballot(wide) == 0ull) {
    finish_group<16>(b, w, prm, cap, out_rec, out_count, out_total, quad_max, pair_max, contrast_list, q4, s_score, s_sid, s_L);
  } else {
    finish_group<32>(b, w, prm, cap, out_rec, out_count, out_total, pair_max, pair_max, contrast_list, q4, s_score, s_sid, s_L);
    finish_group<32>(b, w, prm, cap, out_rec, out_count, out_total, pair_max, pair_max, contrast_list, q4 + 2u, s_score, s_sid, s_L);
  }
}

template <int MODE>
__global__ void __launch_bounds__(kWave) k_contrast(DevIndex ix, DevBatch b, DevWork w, Params prm, DevScratch sc,
                                                    uint32_t cap, Rec* out_rec, uint32_t* out_count,
                                                    uint32_t* out_total, const uint32_t* contrast_list, float* g_rows,
                                                    uint32_t row_len) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_rows = MODE == 3 ? g_rows + (size_t)blockIdx.x * row_len * kWave : (float*)smem;
  constexpr int QMODE = MODE == 3 ? 2 : MODE;  // the queue k_finish filled
  // small LDS footprint (the accepted list is rarely longer than the contrast buffer): more waves per CU
  __shared__ unsigned long long s_keys[kContrastLds];
  __shared__ uint32_t s_vals[kContrastLds];
  __shared__ float    s_f0[kContrastLds], s_f1[kContrastLds];
  const uint32_t lane = lane_id();
  // MODE 0: one wave per query of the sub-batch; MODE 1 / 2: grid-stride over the queue k_finish filled
  const unsigned long long n_items = MODE == 0 ? (unsigned long long)b.nq : w.counters[9 + QMODE];
  for (unsigned long long item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint32_t q = MODE == 0 ? (uint32_t)item : contrast_list[(size_t)(QMODE - 1) * b.nq + item];
    const uint32_t gq = b.q_begin + q;
    if (w.q_cand[q] == 0) continue;  // k_finish already wrote an empty result
    const uint32_t acc = w.q_acc[q];
    if (MODE == 0 && (acc >> 30)) continue;  // queued for a wider kernel
    const uint32_t n = acc & 0x3fffffffu;
    const unsigned long long seg0 = w.q_seg[q];
    const Cand* seg = w.seg + seg0;
    Rec* out = out_rec + (unsigned long long)gq * cap;
    uint32_t cnt = 0, tot = 0;
    if (n != 0) {
      const uint32_t np2 = next_pow2(n);
      if (np2 <= kContrastLds) select_body<MODE>(ix, prm, seg, n, np2, s_keys, s_vals, s_f0, s_f1, s_rows, cap, out, cnt, tot, b.prior_off, b.prior_tok, b.n_prior, b.prior_sid);
      else select_body<MODE>(ix, prm, seg, n, np2, sc.keys + 2ull * seg0, sc.vals + 2ull * seg0, sc.f0 + seg0, sc.f1 + seg0,
                             s_rows, cap, out, cnt, tot, b.prior_off, b.prior_tok, b.n_prior, b.prior_sid);
    }
    if (lane == 0) { out_count[gq] = cnt; if (out_total) out_total[gq] = tot; }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Flattened contrastive rerank (1 <= nmatch <= kMaxFlatRounds; accepted sentences of at most 64 tokens and at most
// 65535 accepted matches per query, everything else goes to k_contrast<2>): ONE kernel per round, one lane per
// still-unpicked accepted match of ANY query (every wave is full).  In round r the lane learns from q_best which match
// its query picked in round r - 1 (src/fuzzy_match.cc:633-668: std::max_element of score - factor * penalty over the
// remaining list, i.e. first maximum in (score desc, s_id asc) order); the picked match emits itself; every other match
// computes its ids-only edit distance to the picked sentence, updates its penalty and bids for round r's pick with one
// 64-bit atomicMax: orderable(score - factor * penalty) << 32 | (0xffff - list rank) << 16 | index.  The bids of three
// consecutive rounds live in three arrays (read r - 1, write r, clear r + 1), so there is no separate pick kernel and no
// second pass over the accepted lists.  State per accepted match lives in the global scratch (pen_sum = sc.f0,
// Match::penalty = sc.f1, sc.keys = taken << 63 | list rank << 32 | running max, sc.vals = where its sentence starts),
// indexed by segment slot.
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint32_t float_orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u >> 31) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long contrast_bid(float key, uint32_t rank, uint32_t k) {
  return ((unsigned long long)float_orderable(key) << 32) | ((unsigned long long)(0xffffu - rank) << 16) | k;
}

__global__ void k_cacc(DevBatch b, DevWork w) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q > b.nq) return;
  unsigned long long v[2] = {0, 0};
  if (q < b.nq && w.q_cand[q] != 0) {
    const uint32_t acc = w.q_acc[q];
    const uint32_t cls = acc >> 30;
    if (cls < 2) v[cls] = acc & 0x3fffffffu;
  }
  const size_t stride = (size_t)b.nq + 1;
  w.c_in[q] = v[0];
  w.c_in[stride + q] = v[1];
}

// wave per query: the flat list of (query, accepted match) pairs, each match's rank in list order and sentence position,
// the bid of round 0 (penalties are zero: the best score, first in list order)
__device__ __forceinline__ void camap_query(const DevIndex& ix, const DevBatch& b, const DevWork& w, const DevScratch& sc, const Params& prm,
                                            uint32_t* out_count, uint32_t* out_total, uint32_t q) {
  if (w.q_cand[q] == 0) return;  // k_finish wrote the empty result
  const uint32_t acc = w.q_acc[q];
  const uint32_t cls = acc >> 30, n = acc & 0x3fffffffu;
  if (cls >= 2) return;
  const uint32_t lane = lane_id();
  if (n == 0) {
    if (lane == 0) { out_count[b.q_begin + q] = 0; if (out_total) out_total[b.q_begin + q] = 0; }
    return;
  }
  const size_t stride = (size_t)b.nq + 1;
  unsigned long long base = w.c_off[cls * stride + q];
  if (cls == 1) base += w.c_off[b.nq];
  const unsigned long long seg0 = w.q_seg[q];
  const Cand* seg = w.seg + seg0;
  unsigned long long best = 0;
  for (uint32_t k0 = 0; k0 < n; k0 += kWave) {
    const uint32_t k = k0 + lane;
    Cand c;
    c.sid = 0; c.longest = 0; c.score = 0.f; c.penalty = 0.f;
    if (k < n) c = seg[k];
    uint32_t rank = 0;
    if (n <= kWave) {  // the usual case: the whole list sits in the wave's registers
      for (uint32_t j = 0; j < n; j++) {
        const float os = __shfl(c.score, (int)j, 64);
        const uint32_t oid = __shfl(c.sid, (int)j, 64);
        rank += (os > c.score || (os == c.score && oid < c.sid)) ? 1u : 0u;
      }
    } else if (k < n) {
      for (uint32_t j = 0; j < n; j++) {
        const float os = seg[j].score;
        const uint32_t oid = seg[j].sid;
        rank += (os > c.score || (os == c.score && oid < c.sid)) ? 1u : 0u;
      }
    }
    if (k >= n) continue;
    w.c_qi[base + k] = ((unsigned long long)q << 32) | k;
    sc.f0[seg0 + k] = 0.f; sc.f1[seg0 + k] = 0.f;
    sc.keys[2ull * seg0 + k] = (unsigned long long)rank << 32;
    sc.vals[2ull * seg0 + k] = ix.sent_pos[c.sid];
    const unsigned long long bid = contrast_bid(c.score - (prm.contrast * 0.f), rank, k);
    best = bid > best ? bid : best;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(best >> 32), d, 64) << 32) | __shfl_xor((uint32_t)best, d, 64);
    best = o > best ? o : best;
  }
  if (lane == 0) { w.q_best[q] = best; w.q_best[(size_t)b.nq + q] = 0; w.q_best[2 * (size_t)b.nq + q] = 0; }
}

// k_camap: FOUR queries per wave, sixteen lanes each (a query of the rerank has ~7 accepted matches: one wave per query ran a
// tenth full); a wave whose four queries do not all fit (more than 16 accepted matches) takes them one after the other with
// the whole wave.
constexpr uint32_t kCamapGroup = 16;
__global__ void __launch_bounds__(kWave) k_camap(DevIndex ix, DevBatch b, DevWork w, DevScratch sc, Params prm, uint32_t* out_count,
                                                 uint32_t* out_total) {
  const uint32_t lane = lane_id(), gl = lane & (kCamapGroup - 1), gb = lane & ~(kCamapGroup - 1);
  const uint32_t q0 = blockIdx.x * (kWave / kCamapGroup);
  const uint32_t q = q0 + lane / kCamapGroup;
  bool work = false;      // this lane's query takes part in the flattened rerank
  uint32_t n = 0, cls = 0;
  if (q < b.nq && w.q_cand[q] != 0) {
    const uint32_t acc = w.q_acc[q];
    cls = acc >> 30; n = acc & 0x3fffffffu;
    work = cls < 2;
  }
  if (__ballot(work && n > kCamapGroup)) {  // (wave-uniform)
    for (uint32_t k = 0; k < kWave / kCamapGroup; k++)
      if (q0 + k < b.nq) camap_query(ix, b, w, sc, prm, out_count, out_total, q0 + k);
    return;
  }
  if (work && n == 0 && gl == 0) { out_count[b.q_begin + q] = 0; if (out_total) out_total[b.q_begin + q] = 0; }
  uint32_t n_max = work ? n : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(n_max, d, 64); n_max = o > n_max ? o : n_max; }
  if (n_max == 0) return;
  const bool mine = work && gl < n;
  const size_t stride = (size_t)b.nq + 1;
  unsigned long long base = 0, seg0 = 0;
  Cand c;
  c.sid = 0; c.longest = 0; c.score = 0.f; c.penalty = 0.f;
  if (mine) {
    base = w.c_off[cls * stride + q];
    if (cls == 1) base += w.c_off[b.nq];
    seg0 = w.q_seg[q];
    c = w.seg[seg0 + gl];
  }
  uint32_t rank = 0;
  for (uint32_t j = 0; j < n_max; j++) {
    const float os = __shfl(c.score, (int)(gb + j), 64);
    const uint32_t oid = __shfl(c.sid, (int)(gb + j), 64);
    rank += (j < n && (os > c.score || (os == c.score && oid < c.sid))) ? 1u : 0u;
  }
  unsigned long long best = 0;
  if (mine) {
    w.c_qi[base + gl] = ((unsigned long long)q << 32) | gl;
    sc.f0[seg0 + gl] = 0.f; sc.f1[seg0 + gl] = 0.f;
    sc.keys[2ull * seg0 + gl] = (unsigned long long)rank << 32;
    sc.vals[2ull * seg0 + gl] = ix.sent_pos[c.sid];
    best = contrast_bid(c.score - (prm.contrast * 0.f), rank, gl);
  }
#pragma unroll
  for (int d = kCamapGroup / 2; d >= 1; d >>= 1) {
    const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(best >> 32), d, 64) << 32) | __shfl_xor((uint32_t)best, d, 64);
    best = o > best ? o : best;
  }
  if (work && n != 0 && gl == 0) { w.q_best[q] = best; w.q_best[(size_t)b.nq + q] = 0; w.q_best[2 * (size_t)b.nq + q] = 0; }
}

template <int P>
__global__ void __launch_bounds__(kWave) k_cround(DevIndex ix, DevBatch b, DevWork w, Params prm, DevScratch sc, uint32_t round,
                                                  uint32_t cap, Rec* out_rec, uint32_t* out_count, uint32_t* out_total) {
  const uint32_t lane = lane_id();
  const size_t stride = (size_t)b.nq + 1;
  const unsigned long long t0 = w.c_off[b.nq], t1 = w.c_off[stride + b.nq];
  const unsigned long long j_lo = P == 32 ? 0ull : t0, j_hi = P == 32 ? t0 : t0 + t1;
  const unsigned long long* bid_prev = w.q_best + (size_t)((round + 2) % 3) * b.nq;  // round - 1
  unsigned long long* bid_cur = w.q_best + (size_t)(round % 3) * b.nq;
  unsigned long long* bid_next = w.q_best + (size_t)((round + 1) % 3) * b.nq;
  uint32_t n_dist = 0;  // statistics: sentence-pair distances of this lane
  for (unsigned long long j = j_lo + (unsigned long long)blockIdx.x * kWave + lane; j < j_hi;
       j += (unsigned long long)gridDim.x * kWave) {
    const unsigned long long qi = w.c_qi[j];
    const uint32_t q = (uint32_t)(qi >> 32), k = (uint32_t)qi;
    const uint32_t n = w.q_acc[q] & 0x3fffffffu;
    const uint32_t want = prm.nmatch < n ? prm.nmatch : n;
    if (round > want) continue;  // this query has emitted everything it will emit
    const unsigned long long seg0 = w.q_seg[q];
    unsigned long long* keys = sc.keys + 2ull * seg0;
    const unsigned long long kw = keys[k];
    if (kw >> 63) continue;      // picked in an earlier round
    const uint32_t kb = (uint32_t)bid_prev[q] & 0xffffu;  // the match picked in round - 1
    const Cand c = w.seg[seg0 + k];
    if (k == kb) {
      const uint32_t gq = b.q_begin + q;
      if (round - 1 < cap) {
        Rec r;
        r.score = c.score; r.penalty = sc.f1[seg0 + k]; r.max_subseq = (int32_t)c.longest; r.s_id = c.sid;
        out_rec[(unsigned long long)gq * cap + (round - 1)] = r;
      }
      keys[k] = kw | (1ull << 63);
      if (round == want) { out_count[gq] = want < cap ? want : cap; if (out_total) out_total[gq] = want; }
      bid_next[q] = 0;
      continue;
    }
    if (round >= want) continue;  // nothing more will be picked for this query
    const uint32_t msp = sc.vals[2ull * seg0 + kb], csp = sc.vals[2ull * seg0 + k];
    const uint32_t n2 = ix.tok[msp], n1 = ix.tok[csp];
    const float dw = 100.f / normalizer(n1, n2, 1.f, 1.f, 1.f);
    float M;
    const float ed = edit_distance_unit<P, false>(ix.tok + csp + 1, n1, ix.tok + msp + 1, n2, 1.f * dw, M);  // costs (1, 1, 1)
    const float pen = score_from_cost(ed);
    const float sum = sc.f0[seg0 + k] + pen;
    sc.f0[seg0 + k] = sum;
    float mx = __uint_as_float((uint32_t)(kw & 0xffffffffull));
    mx = (round == 1) ? pen : (mx < pen ? pen : mx);
    keys[k] = (kw & 0x0000ffff00000000ull) | (unsigned long long)__float_as_uint(mx);
    const float pen_now = prm.reduce == 1 ? mx : (sum / (float)round);
    sc.f1[seg0 + k] = pen_now;
    atomicMax(&bid_cur[q], contrast_bid(c.score - (prm.contrast * pen_now), (uint32_t)(kw >> 32) & 0xffffu, k));
    n_dist++;
  }
  n_dist = wave_sum(n_dist);  // (every lane is back here: the loop's exits are per lane)
  if (lane == 0 && n_dist) atomicAdd(&w.counters[27], (unsigned long long)n_dist);
}

// ---------------------------------------------------------------------------------------------
// Contrastive rerank of the SMALL accepted sets inside one launch (round 4; src/fuzzy_match.cc:614-669).  The flattened rounds
// above pay ~1900 instructions per sentence-pair distance, two thirds of them the match mask (the row's token against all 32
// register-held tokens of the picked sentence), and a launch per round.  Here a GROUP of G lanes owns one query with at most G
// accepted matches (k_cgrp_route sorts the queries into lists by size: G = 8, 16, 32; sentences of up to 32 tokens, no prior
// matches), one match per lane, and runs every round itself: the pick is a maximum over the group's lanes; the picked
// sentence's tokens go into a small LDS hash table of the group (token -> bit mask of its positions: 64 slots, 512 bytes), and
// every other lane walks ITS sentence, one table look-up per token instead of 32 compares, through the same bit-parallel
// recurrence (lev_step32) and the same float sum — penalties, keys and tie-breaks exactly as k_camap / k_cround compute them.
// A lane keeps its sentence in REGISTERS for all rounds (eight 16-byte loads, once) and the picked lane hands its tokens to the
// group through LDS: the first version re-read the tokens from memory four at a time in every round — a chain of ~30 dependent
// loads per query, 80 us per group and no faster than the flattened rounds (0.5 ms per launch whatever the list held).
// ---------------------------------------------------------------------------------------------
// Table of the picked sentence per group: {token, mask of its positions}.  Round 5: 128 slots (a sentence of <= 32 tokens: a quarter
// full) and a look-up that reads the home slot and its two successors at once, no probe loop — the loop was 21 VALU + 39 SALU
// instructions of every DP row.  A table in which some token sits further than two slots from home (8 % of them) is rebuilt with the
// next of four hash multipliers; a group that runs out of multipliers sends its wave through the probe loop for that round.
constexpr uint32_t kGrpSlots = 128;
constexpr uint32_t kGrpWindow = 3;
constexpr uint32_t kGrpTries = 4;
__device__ __forceinline__ uint32_t grp_mult(uint32_t k) { return tab_mult(k); }

// one thread per query: queries the group kernels take are listed by size (up to 8, 16, 32, 64 accepted matches: the last list
// lives in the half of contrast_list the flattened mode leaves unused) and marked (class 3: the flattened path skips them)
__global__ void __launch_bounds__(1024) k_cgrp_route(DevBatch b, DevWork w, uint32_t* list8, uint32_t* list16, uint32_t* list32, uint32_t* list64) {
  __shared__ uint32_t s_n[4], s_base[4];
  if (threadIdx.x < 4) s_n[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  int g = -1;
  uint32_t at = 0;
  if (q < b.nq && w.q_cand[q] != 0) {
    const uint32_t acc = w.q_acc[q];
    const uint32_t n = acc & 0x3fffffffu;
    if ((acc >> 30) == 0 && n >= 1 && n <= 64) g = n <= 8 ? 0 : (n <= 16 ? 1 : (n <= 32 ? 2 : 3));
  }
  if (g >= 0) at = atomicAdd(&s_n[g], 1u);
  __syncthreads();
  if (threadIdx.x < 4) s_base[threadIdx.x] = s_n[threadIdx.x] ? (uint32_t)atomicAdd(&w.counters[28 + threadIdx.x], (unsigned long long)s_n[threadIdx.x]) : 0u;
  __syncthreads();
  if (g >= 0) {
    (g == 0 ? list8 : (g == 1 ? list16 : (g == 2 ? list32 : list64)))[s_base[g] + at] = q;
    w.q_acc[q] |= 3u << 30;
  }
}

// DICT (round 6): per-query dictionary ids.  The tokens of the group's sentences are mapped ONCE per query to slots of a group
// dictionary (16 x G slots: the accepted matches of a query are near-duplicates of it and share most of their words), the rows keep
// slot numbers instead of tokens, and the picked sentence's position masks are indexed by slot: a DP row is one LDS read in front of the
// recurrence instead of a hash, three 8-byte reads, three compares and the selects (12 of a row's 28 instructions).  A group whose
// sentences do not fit the dictionary (a token further than kDictProbe slots from home) hands its query to `fb_list` (count in
// counters[ctr_fb]); the hash-table variant (DICT = false) takes those in a second, mostly empty launch.
constexpr uint32_t kDictProbe = 8;
constexpr int kCtrRerankFb = kCtrDpDefer + 1;   // [+0 .. +3]: queries the dictionary variant handed back, per list
template <int G, bool DICT = false>
__global__ void __launch_bounds__(kWave) k_crerank_g(DevIndex ix, DevBatch b, DevWork w, Params prm, const uint32_t* list,
                                                     uint32_t ctr_in, uint32_t cap, Rec* out_rec, uint32_t* out_count, uint32_t* out_total,
                                                     uint32_t* fb_list = nullptr, uint32_t ctr_fb = 0) {
  constexpr uint32_t kGroups = kWave / G;
  constexpr uint32_t kDictSlots = 16u * G;
  __shared__ uint2 s_tab[DICT ? 1 : kGroups][kGrpSlots + kGrpWindow - 1];   // {token, mask of its positions in the picked sentence}; the last slots mirror the first
  __shared__ uint32_t s_key[DICT ? kGroups : 1][DICT ? kDictSlots : 1];     // DICT: the group's dictionary (token per slot)
  __shared__ uint32_t s_msk[DICT ? kGroups : 1][DICT ? kDictSlots : 1];     // DICT: mask of the slot's positions in the picked sentence
  __shared__ __attribute__((aligned(16))) uint32_t s_pick[kGroups][32];   // the picked sentence's tokens (DICT: their slots)
  const uint32_t lane = lane_id(), gl = lane % G, gi = lane / G, gb = lane - gl;
  const unsigned long long n_list = w.counters[ctr_in];
  const unsigned long long gmask_all = G == 64 ? ~0ull : (((1ull << (G & 63)) - 1ull) << gb);
  uint32_t n_dist = 0;
  for (unsigned long long item0 = (unsigned long long)blockIdx.x * kGroups; item0 < n_list; item0 += (unsigned long long)gridDim.x * kGroups) {
    const unsigned long long item = item0 + gi;
    bool have = item < n_list;
    const uint32_t q = have ? list[item] : 0u;
    const uint32_t n = have ? (w.q_acc[q] & 0x3fffffffu) : 0u;
    const uint32_t gq = b.q_begin + q;
    const bool mine = gl < n;
    Cand c;
    c.sid = 0; c.longest = 0; c.score = 0.f; c.penalty = 0.f;
    if (mine) c = w.seg[w.q_seg[q] + gl];
    const uint32_t csp = mine ? ix.sent_pos[c.sid] : 0u;
    const uint32_t n1 = mine ? ix.tok[csp] : 0u;
    uint32_t tk[32];  // this lane's sentence (<= 32 tokens; the token buffer has kTokSpare spare words behind it)
    {
      // all eight groups of four are requested together with the length word, not behind it (round 6: one dependent level of
      // memory latency fewer in a kernel that waits for five of them per item); what lies past the sentence's end is never used
      const Tok4* s4 = reinterpret_cast<const Tok4*>(ix.tok + csp + 1);
#pragma unroll
      for (int u = 0; u < 8; u++) {
        Tok4 v = Tok4{{0u, 0u, 0u, 0u}};
        if (mine) v = s4[u];
        tk[4 * u] = v.t[0]; tk[4 * u + 1] = v.t[1]; tk[4 * u + 2] = v.t[2]; tk[4 * u + 3] = v.t[3];
      }
    }
    // rank in list order (score desc, s_id asc), like k_camap
    uint32_t rank = 0;
    for (uint32_t j = 0; j < (uint32_t)G; j++) {
      const float os = __shfl(c.score, (int)(gb + j), 64);
      const uint32_t oid = __shfl(c.sid, (int)(gb + j), 64);
      rank += (j < n && (os > c.score || (os == c.score && oid < c.sid))) ? 1u : 0u;
    }
    uint32_t n2_prev = 0;   // DICT: tokens of the sentence whose masks sit in s_msk (its slots in s_pick)
    if constexpr (DICT) {
      // the group's dictionary: every token of its sentences -> a slot; the rows keep the slots
      wave_lds_sync();
      for (uint32_t k = gl; k < kDictSlots; k += G) { s_key[gi][k] = kHashEmpty; s_msk[gi][k] = 0u; }
      wave_lds_sync();
      bool ovf = false;
#pragma unroll
      for (int i = 0; i < 32; i++) {
        if (mine && (uint32_t)i < n1) {
          const uint32_t t = tk[i];
          uint32_t h = (t * 0x9E3779B1u) >> (32 - (G == 8 ? 7 : (G == 16 ? 8 : (G == 32 ? 9 : 10))));
          uint32_t d = 0;
          for (;;) {
            const uint32_t old = atomicCAS(&s_key[gi][h], kHashEmpty, t);
            if (old == kHashEmpty || old == t) break;
            h = (h + 1) & (kDictSlots - 1);
            if (++d >= kDictProbe) { ovf = true; break; }
          }
          tk[i] = h;
        }
      }
      const bool g_ovf = (__ballot(ovf) & gmask_all) != 0ull;
      if (g_ovf) {  // the hash-table variant takes this query
        if (have && gl == 0) fb_list[atomicAdd(&w.counters[ctr_fb], 1ull)] = q;
        have = false;
      }
      wave_lds_sync();
    }
    const bool live = have;   // (DICT: false for a group that handed its query back)
    const uint32_t want = !live ? 0u : (prm.nmatch == 0 ? n : (prm.nmatch < n ? prm.nmatch : n));
    uint32_t want_max = want;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(want_max, d, 64); want_max = o > want_max ? o : want_max; }
    bool picked = !mine;
    float sum = 0.f, mx = 0.f, pen_now = 0.f;
    for (uint32_t r = 0; r < want_max; r++) {       // round r + 1 of k_cround
      const bool on = r < want;                       // this group's query still picks
      unsigned long long bid = (on && !picked) ? contrast_bid(c.score - (prm.contrast * pen_now), rank, gl) : 0ull;
#pragma unroll
      for (int d = G / 2; d >= 1; d >>= 1) {
        const unsigned long long o = ((unsigned long long)__shfl_xor((uint32_t)(bid >> 32), d, 64) << 32) | __shfl_xor((uint32_t)bid, d, 64);
        bid = o > bid ? o : bid;
      }
      const uint32_t kb = (uint32_t)bid & 0xffffu;   // the group's pick of this round
      if (on && mine && gl == kb) {
        if (r < cap) {
          Rec rec;
          rec.score = c.score; rec.penalty = pen_now; rec.max_subseq = (int32_t)c.longest; rec.s_id = c.sid;
          out_rec[(unsigned long long)gq * cap + r] = rec;
        }
        picked = true;
      }
      const bool more = on && r + 1 < want;           // another pick follows: the others need their distance to this one
      if (__ballot(more) == 0ull) { if (r + 1 >= want_max) break; else continue; }
      // the picked sentence's tokens -> the group's table (token -> mask of its positions)
      const uint32_t n2 = __shfl(n1, (int)(gb + kb), 64);
      wave_lds_sync();
      uint32_t mk = 0;        // the group's hash multiplier of this round
      bool use_loop = false;
      if constexpr (DICT) {
        // masks by dictionary slot: the previous pick's go, this pick's come (the tokens ARE slots here)
        if (more) for (uint32_t j = gl; j < n2_prev; j += G) s_msk[gi][s_pick[gi][j]] = 0u;
        wave_lds_sync();
        if (more && mine && gl == kb) {
#pragma unroll
          for (int u = 0; u < 8; u++)
            if ((uint32_t)(4 * u) < n1) *reinterpret_cast<uint4*>(&s_pick[gi][4 * u]) = make_uint4(tk[4 * u], tk[4 * u + 1], tk[4 * u + 2], tk[4 * u + 3]);
        }
        wave_lds_sync();
        if (more) {
          for (uint32_t j = gl; j < n2; j += G) atomicOr(&s_msk[gi][s_pick[gi][j]], 1u << j);
          n2_prev = n2;
        }
        wave_lds_sync();
      } else {
      if (more && mine && gl == kb) {  // four tokens per LDS write (words past the sentence's end are zero and never read)
#pragma unroll
        for (int u = 0; u < 8; u++)
          if ((uint32_t)(4 * u) < n1) *reinterpret_cast<uint4*>(&s_pick[gi][4 * u]) = make_uint4(tk[4 * u], tk[4 * u + 1], tk[4 * u + 2], tk[4 * u + 3]);
      }
      bool build = more, stuck = false;
      for (uint32_t tr = 0;; tr++) {
        if (build)
          for (uint32_t k = gl; k < kGrpSlots; k += G) s_tab[gi][k] = make_uint2(kHashEmpty, 0u);
        wave_lds_sync();
        bool far = false;
        if (build) {
          const uint32_t mult = grp_mult(mk);
          for (uint32_t j = gl; j < n2; j += G) {
            const uint32_t t = s_pick[gi][j];
            uint32_t h = (t * mult) >> 25, d = 0;
            for (;;) {
              const uint32_t old = atomicCAS(&s_tab[gi][h].x, kHashEmpty, t);
              if (old == kHashEmpty || old == t) { atomicOr(&s_tab[gi][h].y, 1u << j); break; }
              h = (h + 1) & (kGrpSlots - 1); d++;
            }
            far = far || d >= kGrpWindow;
          }
        }
        wave_lds_sync();
        const unsigned long long fm = __ballot(far);
        const bool gfar = (fm & gmask_all) != 0ull;            // some token of my group's table sits too far from home
        build = gfar && tr + 1 < kGrpTries;
        stuck = gfar && tr + 1 >= kGrpTries;
        if (build) mk++;
        if (__ballot(build) == 0ull) break;
      }
      use_loop = __ballot(stuck) != 0ull;       // (wave-uniform: rare)
      if (more && gl < kGrpWindow - 1) s_tab[gi][kGrpSlots + gl] = s_tab[gi][gl];
      wave_lds_sync();
      }
      // every unpicked match of the group: bit-parallel distance of its sentence (rows) to the picked one (bits)
      const bool act = more && mine && !picked;
      const uint32_t rows = act ? n1 : 0u;
      uint32_t rows_max = rows;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(rows_max, d, 64); rows_max = o > rows_max ? o : rows_max; }
      uint32_t pv = ~0u, mv = 0u, score = n2;
      const uint32_t m_bits = n2 ? n2 : 1u;
      const uint32_t mult = grp_mult(mk);
      if constexpr (DICT) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
          if ((uint32_t)i >= rows_max) break;   // (wave-uniform)
          if ((uint32_t)i < rows) lev_step32(pv, mv, score, s_msk[gi][tk[i]], m_bits);
        }
      } else if (!use_loop) {
#pragma unroll
        for (int i = 0; i < 32; i++) {
          if ((uint32_t)i >= rows_max) break;   // (wave-uniform)
          if ((uint32_t)i < rows) {
            const uint32_t t = tk[i];
            const uint32_t h = (t * mult) >> 25;
            const uint2 e0 = s_tab[gi][h], e1 = s_tab[gi][h + 1], e2 = s_tab[gi][h + 2];
            const uint32_t eq = (e0.x == t ? e0.y : 0u) | (e1.x == t ? e1.y : 0u) | (e2.x == t ? e2.y : 0u);
            lev_step32(pv, mv, score, eq, m_bits);
          }
        }
      } else {
#pragma unroll 1
        for (int i = 0; i < 32; i++) {
          if ((uint32_t)i >= rows_max) break;   // (wave-uniform)
          uint32_t t = 0;
#pragma unroll
          for (int u = 0; u < 32; u++) t = u == i ? tk[u] : t;   // (the rare path: no second unrolled copy of the rows)
          if ((uint32_t)i < rows) {
            uint32_t h = (t * mult) >> 25;
            uint2 e = s_tab[gi][h];
            while (e.x != t && e.x != kHashEmpty) { h = (h + 1) & (kGrpSlots - 1); e = s_tab[gi][h]; }
            const uint32_t eq = e.x == t ? e.y : 0u;
            lev_step32(pv, mv, score, eq, m_bits);
          }
        }
      }
      if (act) {
        const float dw = 100.f / normalizer(n1, n2, 1.f, 1.f, 1.f);
        const float ed = n2 ? unit_cost_sum(score, 1.f * dw) : unit_cost_sum(n1, 1.f * dw);
        const float pen = score_from_cost(ed);
        sum = sum + pen;
        mx = (r == 0) ? pen : (mx < pen ? pen : mx);
        pen_now = prm.reduce == 1 ? mx : (sum / (float)(r + 1));
        n_dist++;
      }
    }
    if (live && gl == 0) { out_count[gq] = want < cap ? want : cap; if (out_total) out_total[gq] = want; }
  }
  n_dist = wave_sum(n_dist);
  if (lane == 0 && n_dist) atomicAdd(&w.counters[27], (unsigned long long)n_dist);
}

// per-call statistics from the per-query arrays (kept out of the hot kernels: one shared atomic
// per query costs more than the kernels' real work)
__global__ void __launch_bounds__(1024) k_stats(DevWork w, uint32_t nq) {
  __shared__ unsigned long long s_v[7];  // [0..4], [6] sums; [5] maximum
  if (threadIdx.x < 7) s_v[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long v[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long mx = 0;
  if (q < nq) {
    v[0] = w.q_m[q] == kNoMatchLen ? 0ull : w.q_hits[q];
    v[1] = w.q_kept[q] & ~kKeptReduced; v[2] = w.q_cand[q]; v[3] = w.q_cand[q]; v[4] = w.q_pb[q];
    v[5] = w.q_capl[q];  // hits of the light queries (walked by k_expand_b<true>, not by phase A)
    mx = w.q_kept[q] & ~kKeptReduced;
  }
#pragma unroll
  for (int k = 0; k < 6; k++) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v[k] += __shfl_xor(v[k], d, 64);
    if ((threadIdx.x & 63) == 0 && v[k]) atomicAdd(&s_v[k < 5 ? k : 6], v[k]);
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const unsigned long long o = __shfl_xor(mx, d, 64); mx = o > mx ? o : mx; }
  if ((threadIdx.x & 63) == 0 && mx) atomicMax(&s_v[5], mx);
  __syncthreads();
  const int slot[5] = {0, 1, 2, 3, 8};
  if (threadIdx.x < 5 && s_v[threadIdx.x]) atomicAdd(&w.counters[slot[threadIdx.x]], s_v[threadIdx.x]);
  if (threadIdx.x == 5 && s_v[5]) atomicMax(&w.counters[9], s_v[5]);
  if (threadIdx.x == 6 && s_v[6]) atomicAdd(&w.counters[22], s_v[6]);
  if (blockIdx.x == 0 && threadIdx.x < 3 * kStatSlots) {  // k_expand_h's slots -> [16], [17], [18]
    const unsigned long long v2 = w.counters[kStatBase + threadIdx.x];
    if (v2) atomicAdd(&w.counters[16 + threadIdx.x / kStatSlots], v2);
  }
}

}  // namespace fmx
