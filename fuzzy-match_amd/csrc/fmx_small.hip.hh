// fmx_small.hip.hh — match() for a micro-batch in ONE kernel launch.
//
// The pipeline of fmx_kernels.hip.hh is built for throughput: ~30 dependent launches per pass, ~230 us for a pass of one
// query.  The reference's calling pattern is N threads with one pattern per call (cli/src/FuzzyMatch-cli.cc:139-147), so the
// coalescer's micro-batches are small and every caller waits a whole pass.  k_match_small runs the same steps for one query
// per wave — probe (src/suffix_array.cc:105-212), hit walk with dominance / length / exact coverage filters
// (src/ngram_matches.cc:31-83, src/pattern_coverage.cc:8-28), max per sentence and candidate order
// (src/ngram_matches.cc:19-29), bit-parallel edit distance (src/edit_distance.cc:5-77, unit costs), lowest_costs replay and
// top-N (src/fuzzy_match.cc:564-612,670-679) — built from the same per-lane functions, for the queries that are "light" in
// every respect (pattern of one wave, at most kSmallMaxHits hits, none in a long range, at most 64 survivors); any other
// query is flagged and goes through the pipeline afterwards.  Same records, bit for bit (tests/test_gpu_small.py).
#pragma once
#include "fmx_kernels.hip.hh"

namespace fmx {

constexpr uint32_t kSmallMaxQueries = 512;   // micro-batches up to this size take the one-launch path
constexpr uint32_t kSmallMaxHits = 256;      // like the light queries of the pipeline

struct SmallArgs {
  const uint32_t* q_off;     // nq + 1 offsets (absolute into q_tok)
  const uint32_t* q_tok;
  uint32_t        nq;
  uint32_t        cap;
  uint32_t        heavy_min;
  Rec*            out;       // nq x cap
  uint32_t*       out_count; // nq
  uint32_t*       out_total; // nq (may be null)
  uint32_t*       deferred;  // nq: 1 = not handled here
};

// Host-side eligibility of the parameters: ids-only unit costs (the bit-parallel distance), no idf term, no contrastive rerank,
// lowest_costs small enough for the in-register replay.  (Real-token / penalty-token cost terms are excluded by the caller.)
inline bool small_path_params_ok(const Params& prm) {
  return prm.contrast <= 0.f && prm.idf_penalty == 0.f && prm.ins == prm.del && prm.del == prm.rep && prm.rep >= 0.f &&
         prm.cbuf > 0 && prm.cbuf <= (int)kHeapLdsMax;
}

__global__ void __launch_bounds__(kWave) k_match_small(DevIndex ix, SmallArgs a, Params prm) {
  __shared__ uint32_t s_pat[kWave];
  __shared__ PatHash ph;
  __shared__ uint32_t s_npref[kWave], s_nlo[kWave], s_nit[kWave];
  __shared__ SurvB s_surv[kWave * (kBUnroll + 1)];
  __shared__ unsigned long long s_out[kWave];
  __shared__ Cand s_cand[kWave];
  __shared__ ReplayShared sh;
  const uint32_t q = blockIdx.x;
  if (q >= a.nq) return;
  const uint32_t lane = lane_id();
  const unsigned long long below = (1ull << lane) - 1ull;
  const uint32_t o0 = uni(a.q_off[q]);
  const uint32_t p = uni(a.q_off[q + 1]) - o0;
  const uint32_t* pat = a.q_tok + o0;
  auto finish = [&](uint32_t count, uint32_t total, uint32_t defer) {
    if (lane == 0) { a.out_count[q] = count; if (a.out_total) a.out_total[q] = total; a.deferred[q] = defer; }
  };
  const uint32_t msl = min_seq_len(p, prm);
  if (msl == kNoMatchLen) { finish(0, 0, 0); return; }  // match() returns early (src/fuzzy_match.cc:455-467)
  if (p > kWave) { finish(0, 0, 1); return; }
  const uint32_t m = seed_len(p, msl);
  // ---- probe: lane = start position ----
  uint32_t lo = 0, hi = 0;
  {
    const bool reachable = (p == 1) ? (msl <= 1) : true;
    if (lane < p && reachable && lane + m <= p)
      mgram_range_hashed(ix.gram, ix.suf_key, ix.suf, ix.tok, ix.qva, ix.n_qva, pat + lane, m, lo, hi);
  }
  const uint32_t mycnt = hi - lo;
  const unsigned long long big = __ballot(mycnt >= a.heavy_min || mycnt > kSmallMaxHits);
  const uint32_t incl = wave_incl_scan(big ? 0u : mycnt);
  const uint32_t n_hits = __shfl(incl, 63, 64);
  if (big || n_hits > kSmallMaxHits) { finish(0, 0, 1); return; }
  if (n_hits == 0) { finish(0, 0, 0); return; }
  // ---- the non-empty start positions, compacted (hit e belongs to the last one whose prefix is <= e) ----
  s_pat[lane] = lane < p ? pat[lane] : 0u;
  {
    const unsigned long long nz = __ballot(mycnt != 0);
    if (mycnt != 0) {
      const uint32_t k = __popcll(nz & below);
      s_npref[k] = incl - mycnt; s_nlo[k] = lo; s_nit[k] = lane;
    }
    if (lane >= (uint32_t)__popcll(nz)) s_npref[lane] = 0xffffffffu;
  }
  wave_lds_sync();
  pat_hash_build(ph, s_pat, p);
  // ---- hit walk: records, dominance + length filter, then the token work four lanes per survivor ----
  constexpr uint32_t kSubLanes = 4, kSubSurv = kWave / kSubLanes;
  uint32_t n_out = 0, n_s = 0;
  bool overflow = false;
  auto token_work = [&](uint32_t j0, uint32_t n_act) {
    const uint32_t sub = lane / kSubLanes, part = lane % kSubLanes;
    const bool act = sub < n_act;
    SurvB sv;
    sv.off = 0; sv.rem_slen = 0; sv.sid = 0; sv.it_need = 0;
    if (act) sv = s_surv[j0 + sub];
    const uint32_t it = sv.it_need >> 16;
    const uint32_t rem = sv.rem_slen & 0xffffu, sl = sv.rem_slen >> 16;
    constexpr int kLcpAhead = 4, kCovAhead = 8;
    const uint32_t lim = act ? (rem < p - it ? rem : p - it) : 0u;
    const uint32_t* sent = ix.tok + (sv.off - (sl - rem));
    uint32_t lt[kLcpAhead], ct[kCovAhead];
#pragma unroll
    for (int u = 0; u < kLcpAhead; u++) { const uint32_t k = m + u * kSubLanes + part; lt[u] = k < lim ? ix.tok[sv.off + k] : 0u; }
#pragma unroll
    for (int u = 0; u < kCovAhead; u++) { const uint32_t k = part + u * kSubLanes; ct[u] = (act && k < sl) ? sent[k] : kHashEmpty; }
    uint32_t L = m < lim ? m : lim;
    bool done = !act || m >= lim;
    uint32_t k0 = m;
#pragma unroll
    for (int u = 0; u < kLcpAhead; u++) {
      const uint32_t k = k0 + part;
      const bool mism = !done && (k >= lim || lt[u] != s_pat[it + k]);
      const uint32_t g = (uint32_t)(__ballot(mism) >> (sub * kSubLanes)) & ((1u << kSubLanes) - 1u);
      if (!done) { if (g) { L = k0 + (uint32_t)__ffs((int)g) - 1u; done = true; } else k0 += kSubLanes; }
    }
    while (__ballot(!done)) {
      const uint32_t k = k0 + part;
      const bool mism = !done && (k >= lim || ix.tok[sv.off + k] != s_pat[it + k]);
      const uint32_t g = (uint32_t)(__ballot(mism) >> (sub * kSubLanes)) & ((1u << kSubLanes) - 1u);
      if (!done) { if (g) { L = k0 + (uint32_t)__ffs((int)g) - 1u; done = true; } else k0 += kSubLanes; }
    }
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
      for (uint32_t kb = part + kCovAhead * kSubLanes; kb < sl; kb += kSubLanes) probe(sent[kb]);
    }
#pragma unroll
    for (uint32_t d = 1; d < kSubLanes; d <<= 1) {
      const uint32_t l2 = __shfl_xor((uint32_t)seen, d, 64), h2 = __shfl_xor((uint32_t)(seen >> 32), d, 64);
      seen |= ((unsigned long long)h2 << 32) | l2;
    }
    uint32_t cover = (uint32_t)__popcll(seen);
    if (L >= p) cover = p;
    // NGramMatches::theoretical_rejection / _cover, evaluated exactly (the pipeline's tabulated thresholds are these)
    const bool keep = act && part == 0 && !reject_len(p, sl, prm) && !reject_cover(p, sl, cover, prm);
    const unsigned long long mask = __ballot(keep);
    if (mask) {
      const uint32_t cnt = __popcll(mask);
      if (n_out + cnt > kWave) overflow = true;
      else if (keep) s_out[n_out + __popcll(mask & below)] = ((unsigned long long)sv.sid << 16) | (L & 0xffffu);
      n_out += cnt;
    }
  };
  for (uint32_t e0 = 0; e0 < n_hits && !overflow; e0 += kWave * kBUnroll) {
    QHit hq[kBUnroll];
    SufRecB rb[kBUnroll];
    bool v[kBUnroll];
#pragma unroll
    for (int u = 0; u < kBUnroll; u++) {
      const uint32_t e = e0 + u * kWave + lane;
      v[u] = e < n_hits;
      hq[u].i = 0; hq[u].it_need = 0;
      if (v[u]) {
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = kWave / 2; step >= 1; step >>= 1)
          if (s_npref[k + step] <= e) k += step;
        hq[u].i = s_nlo[k] + (e - s_npref[k]);
        hq[u].it_need = s_nit[k] << 16;
      }
    }
#pragma unroll
    for (int u = 0; u < kBUnroll; u++) {
      rb[u].off = 0; rb[u].rem = 0; rb[u].slen = 0; rb[u].prev = 0; rb[u].sid = 0;
      if (v[u]) rb[u] = ix.suf_b[hq[u].i];
    }
#pragma unroll
    for (int u = 0; u < kBUnroll; u++) {
      const uint32_t it = hq[u].it_need >> 16;
      const bool dominated = it > 0 && rb[u].rem < rb[u].slen && rb[u].prev == s_pat[it ? it - 1 : 0];
      const bool keep = v[u] && !dominated && !reject_len(p, rb[u].slen, prm);
      const unsigned long long mask = __ballot(keep);
      if (keep) {
        SurvB sv;
        sv.off = rb[u].off; sv.rem_slen = (uint32_t)rb[u].rem | ((uint32_t)rb[u].slen << 16);
        sv.sid = rb[u].sid; sv.it_need = hq[u].it_need;
        s_surv[n_s + __popcll(mask & below)] = sv;
      }
      n_s += __popcll(mask);
    }
    wave_lds_sync();
    const bool more = e0 + kWave * kBUnroll < n_hits;
    uint32_t j0 = 0;
    for (; j0 + kSubSurv <= n_s; j0 += kSubSurv) token_work(j0, kSubSurv);
    const uint32_t rest = n_s - j0;
    if (!more) { if (rest) token_work(j0, rest); n_s = 0; }
    else {
      if (j0) {
        SurvB tmp;
        if (lane < rest) tmp = s_surv[j0 + lane];
        wave_lds_sync();
        if (lane < rest) s_surv[lane] = tmp;
        wave_lds_sync();
      }
      n_s = rest;
    }
  }
  if (overflow) { finish(0, 0, 1); return; }
  if (n_out == 0) { finish(0, 0, 0); return; }
  // ---- LongestMatches: max match length per sentence, order (longest desc, sentence id asc) ----
  wave_lds_sync();
  unsigned long long key = lane < n_out ? s_out[lane] : ~0ull;
  if (n_out > 1) key = wave_sort64_upto(key, n_out);
  const uint32_t nlo = __shfl_down((uint32_t)key, 1, 64), nhi = __shfl_down((uint32_t)(key >> 32), 1, 64);
  const unsigned long long nxt = lane == 63 ? ~0ull : (((unsigned long long)nhi << 32) | nlo);
  const bool last = lane < n_out && (nxt >> 16) != (key >> 16);
  unsigned long long k2 = last ? (((unsigned long long)(0xffffu - (uint32_t)(key & 0xffffu)) << 32) | (key >> 16)) : ~0ull;
  const unsigned long long lastm = __ballot(last);
  const uint32_t n_cand = __popcll(lastm);
  if (n_cand > 1) k2 = wave_sort64_upto(k2, n_out);
  else {
    const int src = __ffsll((long long)lastm) - 1;
    k2 = ((unsigned long long)__shfl((uint32_t)(k2 >> 32), src, 64) << 32) | __shfl((uint32_t)k2, src, 64);
  }
  // ---- edit distance, one candidate per lane (unit costs: bit-parallel distance, then the float sum of the DP) ----
  if (lane < n_cand) {
    const uint32_t sid = (uint32_t)(k2 & 0xffffffffu);
    const uint32_t L = 0xffffu - (uint32_t)(k2 >> 32);
    const uint32_t sp = ix.sent_pos[sid];
    const uint32_t slen = ix.tok[sp];
    const uint32_t* sent = ix.tok + sp + 1;
    const float dw = 100.f / normalizer(p, slen, prm.ins, prm.del, prm.rep);
    const float inc = prm.rep * dw;
    float M;
    const float fc = p <= 32 ? edit_distance_unit<32, false>(sent, slen, pat, p, inc, M) : edit_distance_unit<64, false>(sent, slen, pat, p, inc, M);
    Cand c;
    c.sid = sid; c.longest = (L & 0xffffu) | (slen << 16); c.score = fc; c.penalty = M;
    s_cand[lane] = c;
  }
  wave_lds_sync();
  // ---- candidate loop with the lowest_costs bound, then the result queue's order (score desc, s_id asc) ----
  const unsigned long long acc = replay_small(prm, p, n_cand, s_cand, sh);
  const uint32_t n = (uint32_t)__popcll(acc);
  const uint32_t want = prm.nmatch == 0 ? n : (prm.nmatch < n ? prm.nmatch : n);
  const uint32_t emit = want < a.cap ? want : a.cap;
  const bool mine = (acc >> lane) & 1ull;
  unsigned long long okey = mine ? (((unsigned long long)(~float_key(sh.score[lane])) << 32) | sh.sid[lane]) : ~0ull;
  uint32_t idx = lane;
  wave_sort64_kv_upto(okey, idx, n_cand);
  if (lane < emit) {
    Rec r;
    r.score = sh.score[idx]; r.penalty = 0.f; r.max_subseq = (int32_t)sh.L[idx]; r.s_id = sh.sid[idx];
    a.out[(size_t)q * a.cap + lane] = r;
  }
  finish(emit, want, 0);
}

}  // namespace fmx
