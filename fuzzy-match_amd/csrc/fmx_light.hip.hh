// fmx_light.hip.hh — the light queries' hit walk, FLAT (round 5).
//
// k_expand_b<true> gives every light query a wave of its own: ranges walked, dominance + length filter, match length, exact
// coverage, max per sentence and candidate order — a fixed instruction stream of ~500 VALU instructions per query whatever the
// query holds, and a light query of C3 holds about thirty hits, nine survivors of the cheap tests and seven kept ones: the longest
// kernel of the pipeline ran at a fraction of its lanes (round-4 review).  Here a wave takes a TILE of G consecutive queries and
// works through it in sub-tiles of as many whole queries as fit a token and a hit budget (TB, HB); inside a sub-tile nothing is
// per query any more:
//
//   stage     one lane per pattern token of the sub-tile: pattern -> LDS, the non-empty start positions compacted with their
//             running hit count (the flat hit index space of the sub-tile)
//   table     ONE hash set for the sub-tile, keyed by (query, word): slot -> a representative position of the word in its
//             pattern; per representative the number of positions that carry the word (count_covered_words sums those,
//             src/pattern_coverage.cc:15-28)
//   hits      one lane per HIT, 64 at a time, whatever query it belongs to (binary search over the compacted starts): the
//             16-byte record (ix.suf_b), dominance by the previous start position, the tabulated length filter of k_cmin
//             (src/ngram_matches.cc:31-39, 61-83); survivors are ballot-compacted into LDS
//   tokens    four lanes per survivor, sixteen survivors a pass, any mix of queries: match length (the LCP the reference's
//             narrowing loop registers, src/fuzzy_match.cc:500-550) and the exact coverage test
//             (src/ngram_matches.cc:41-59) through the shared table
//   reduce    the kept hits of several queries at once, up to 64 a round: two sorting networks on composite keys — (query,
//             sentence, length) for the maximum per sentence (src/ngram_matches.cc:79-81), (query, length desc, sentence asc)
//             for get_longest_matches' order (src/ngram_matches.cc:19-29) — and a segmented rank for the position inside
//             each query's segment
//
// Results are those of k_expand_b<true> entry for entry: the same candidates in the same order in the same segments (q_segl),
// q_cand, q_kept | kKeptReduced, token base and sentence length beside each candidate for the edit-distance kernels.  A query
// that keeps more than 64 hits leaves them unreduced in its segment for k_cand_route, like there.  Queries with more hits than
// the budget (flat_max < light_max) stay with k_expand_b<true>.
#pragma once
#include "fmx_kernels.hip.hh"

namespace fmx {

#ifndef FMX_LF_WAVES
#define FMX_LF_WAVES 4
#endif
constexpr uint32_t kLfMaxG = 32;  // queries per tile (one lane of the first half-wave holds a query's scalars)

__device__ __forceinline__ uint32_t lf_hash(uint32_t w, uint32_t ql) { return (w * 0x85EBCA6Bu + ql * 0x9E3779B1u) >> 16; }

template <int TB, int HB>
struct LfShared {
  static constexpr int kSlots = 2 * TB;       // hash slots: load <= 0.5
  static constexpr int kStarts = TB;          // a start position per pattern token at most (power of two: the search needs no bound)
  uint32_t pat[TB];                            // pattern tokens of the sub-tile's queries, query after query
  uint32_t npref[kStarts];                     // compacted non-empty starts: hits in front (exclusive), sentinel behind the last
  uint32_t nlo[kStarts];                       // ... first suffix of the range
  uint16_t nqi[kStarts];                       // ... query << 6 | start position
  unsigned long long hkey[kSlots];             // (query + 1) << 32 | word; 0 = empty
  uint8_t  hj0[kSlots];                        // representative position of the word in its pattern
  uint32_t cnt[TB / 4];                        // positions per representative, one byte each
  SurvB    surv[kWave + 16];                   // survivors of the cheap tests waiting for the token work
  SurvB    keep[HB];                           // kept hits: {sid, L | query << 16, token base of the sentence, its length}
  uint16_t q_tok0[kLfMaxG];                    // per query of the tile: first token in pat
  uint8_t  q_p[kLfMaxG], q_m[kLfMaxG];         // pattern length, seed length
  uint32_t q_seg0[kLfMaxG];                    // segment base
  uint16_t q_kbeg[kLfMaxG], q_kend[kLfMaxG];   // its kept hits in keep[]
};

template <int TB, int HB>
__global__ void __launch_bounds__(kWave, FMX_LF_WAVES) k_expand_lf(DevIndex ix, DevBatch b, DevWork w, Params prm, uint32_t G,
                                                                   uint32_t flat_max) {
  static_assert((TB & (TB - 1)) == 0 && TB >= 64 && HB >= 64, "budgets");
  __shared__ LfShared<TB, HB> S;
  constexpr uint32_t kSlotMask = LfShared<TB, HB>::kSlots - 1;
  const uint32_t lane = lane_id();
  const uint32_t q0 = blockIdx.x * G;
  if (q0 >= b.nq) return;
  // ---- per query of the tile (lane ql): everything requested together ----
  const uint32_t qme = q0 + lane;
  const bool in_tile = lane < G && qme < b.nq;
  unsigned long long seg0 = 0, seg1 = 0;
  uint32_t o0 = 0, o1 = 0, msl = kNoMatchLen;
  if (in_tile) {
    seg0 = w.q_segl[qme]; seg1 = w.q_segl[qme + 1];
    o0 = b.q_off[b.q_begin + qme]; o1 = b.q_off[b.q_begin + qme + 1];
    msl = w.q_m[qme];
  }
  const uint32_t n_q = (uint32_t)(seg1 - seg0);  // a light query's hits; 0 = not light (k_caps_scan)
  const uint32_t p_me = o1 - o0;
  bool flat = in_tile && n_q != 0 && n_q <= flat_max && msl != kNoMatchLen;
  if (flat && seg1 > w.seg_cap) {  // segments too small: the host re-runs the pass (like k_expand_b)
    atomicMax(&w.counters[6], w.q_segl[b.nq]);
    flat = false;
  }
  const uint32_t tok_eff = flat ? p_me : 0u, hit_eff = flat ? n_q : 0u;
  const uint32_t scan_t = wave_incl_scan(tok_eff), scan_h = wave_incl_scan(hit_eff);
  if (flat) {
    S.q_p[lane] = (uint8_t)p_me; S.q_m[lane] = (uint8_t)seed_len(p_me, msl); S.q_seg0[lane] = (uint32_t)seg0;
  }
  uint32_t a = 0;  // first query of the tile not yet worked on
  while (a < G) {
    // ---- sub-tile: queries [a, a + n_sub) — as many as fit the budgets (a flat query alone always fits) ----
    const uint32_t t_base = __shfl(scan_t - tok_eff, a, 64), h_base = __shfl(scan_h - hit_eff, a, 64);
    const bool fits = lane >= a && lane < G && scan_t - t_base <= (uint32_t)TB && scan_h - h_base <= (uint32_t)HB;
    const uint32_t n_sub = __popcll(__ballot(fits));  // (a prefix of the lanes from a on: the sums are monotone)
    const uint32_t a_end = a + n_sub;
    const uint32_t T = __shfl(scan_t, a_end - 1, 64) - t_base;   // tokens, hits of the sub-tile
    const uint32_t H = __shfl(scan_h, a_end - 1, 64) - h_base;
    if (H == 0) { a = a_end; continue; }
    const bool mine = lane >= a && lane < a_end;
    if (mine) { S.q_tok0[lane] = (uint16_t)(scan_t - tok_eff - t_base); S.q_kbeg[lane] = 0; S.q_kend[lane] = 0; }
    for (uint32_t i = lane; i < LfShared<TB, HB>::kSlots; i += kWave) S.hkey[i] = 0ull;
    if (lane < TB / 4) S.cnt[lane] = 0u;
    if (TB / 4 > kWave && lane + kWave < TB / 4) S.cnt[lane + kWave] = 0u;
    wave_lds_sync();
    // ---- stage: one lane per pattern token; the query of token tl = the last one whose first token is <= tl ----
    uint32_t ns = 0, hsum = 0;
    for (uint32_t t0 = 0; t0 < T; t0 += kWave) {
      const uint32_t tl = t0 + lane;
      const bool act = tl < T;
      uint32_t ql = a, wd = 0, cnt = 0, lo = 0, it = 0;
      if (act) {
        // (queries that are not flat have no tokens here: their q_tok0 equals their successor's, the search skips them)
        uint32_t lo_q = a, hi_q = a_end;  // invariant: q_tok0[lo_q] <= tl, q_tok0[hi_q] > tl (hi_q == a_end: none)
        while (hi_q - lo_q > 1) {
          const uint32_t mid = (lo_q + hi_q) >> 1;
          if (S.q_tok0[mid] <= tl) lo_q = mid; else hi_q = mid;
        }
        ql = lo_q;
      }
      // the query's own scalars live in lane ql
      const uint32_t q_o0 = __shfl(o0, ql, 64);
      if (act) {
        it = tl - S.q_tok0[ql];
        const uint32_t tg = q_o0 - b.t_begin + it;
        wd = b.q_tok[b.t_begin + tg];
        cnt = w.t_cnt[tg];
        lo = w.t_lo[tg];
        S.pat[tl] = wd;
      }
      const uint32_t incl = wave_incl_scan(cnt);
      const unsigned long long nz = __ballot(cnt != 0);
      if (cnt != 0) {
        const uint32_t k = ns + __popcll(nz & ((1ull << lane) - 1ull));
        S.npref[k] = hsum + incl - cnt;
        S.nlo[k] = lo;
        S.nqi[k] = (uint16_t)((ql << 6) | it);
      }
      ns += __popcll(nz);
      hsum += __shfl(incl, 63, 64);
      // the word into the sub-tile's hash set: the first lane to claim (query, word) names the representative position
      if (act) {
        const unsigned long long key = ((unsigned long long)(ql + 1) << 32) | wd;
        uint32_t h = lf_hash(wd, ql) & kSlotMask;
        for (;;) {
          const unsigned long long old = atomicCAS(&S.hkey[h], 0ull, key);
          if (old == 0ull) { S.hj0[h] = (uint8_t)it; break; }
          if (old == key) break;
          h = (h + 1) & kSlotMask;
        }
      }
    }
    for (uint32_t k = ns + lane; k < (uint32_t)LfShared<TB, HB>::kStarts; k += kWave) S.npref[k] = 0xffffffffu;  // sentinels
    wave_lds_sync();
    // positions per representative (one byte each: at most 64)
    for (uint32_t tl = lane; tl < T; tl += kWave) {
      uint32_t lo_q = a, hi_q = a_end;
      while (hi_q - lo_q > 1) {
        const uint32_t mid = (lo_q + hi_q) >> 1;
        if (S.q_tok0[mid] <= tl) lo_q = mid; else hi_q = mid;
      }
      const uint32_t wd = S.pat[tl];
      const unsigned long long key = ((unsigned long long)(lo_q + 1) << 32) | wd;
      uint32_t h = lf_hash(wd, lo_q) & kSlotMask;
      while (S.hkey[h] != key) h = (h + 1) & kSlotMask;
      const uint32_t slot = S.q_tok0[lo_q] + S.hj0[h];
      atomicAdd(&S.cnt[slot >> 2], 1u << (8 * (slot & 3)));
    }
    wave_lds_sync();

    // ---- hits, 64 at a time; survivors of the cheap tests wait in S.surv for full passes of the token work ----
    uint32_t n_s = 0, n_k = 0;
    constexpr uint32_t kSubLanes = 4, kSubSurv = kWave / kSubLanes;
    auto token_work = [&](uint32_t j0, uint32_t n_act) {
      const uint32_t sub = lane / kSubLanes, part = lane % kSubLanes;
      const bool act = sub < n_act;
      SurvB sv;
      sv.off = 0; sv.rem_slen = 0; sv.sid = 0; sv.it_need = 0;
      if (act) sv = S.surv[j0 + sub];
      const uint32_t need = sv.it_need & 0xffffu, it = (sv.it_need >> 16) & 63u, ql = sv.it_need >> 24;
      const uint32_t rem = sv.rem_slen & 0xffffu, sl = sv.rem_slen >> 16;
      const uint32_t p = act ? S.q_p[ql] : 0u, m = act ? S.q_m[ql] : 0u, tok0 = act ? S.q_tok0[ql] : 0u;
      // everything the group needs from the sentence is requested at once (one memory latency per survivor)
      constexpr int kLcpAhead = 4, kCovAhead = 8;
      const uint32_t lim = act ? (rem < p - it ? rem : p - it) : 0u;
      const uint32_t* sent = ix.tok + (sv.off - (sl - rem));
      uint32_t lt[kLcpAhead], ct[kCovAhead];
#pragma unroll
      for (int u = 0; u < kLcpAhead; u++) { const uint32_t k = m + u * kSubLanes + part; lt[u] = k < lim ? ix.tok[sv.off + k] : 0u; }
#pragma unroll
      for (int u = 0; u < kCovAhead; u++) { const uint32_t k = part + u * kSubLanes; ct[u] = (act && k < sl) ? sent[k] : kHashEmpty; }
      // match length: kSubLanes tokens per step, a position at or beyond the limit counts as a mismatch
      const uint32_t* pat = S.pat + tok0 + it;
      uint32_t L = m < lim ? m : lim;
      bool done = !act || m >= lim;
      uint32_t k0 = m;
#pragma unroll
      for (int u = 0; u < kLcpAhead; u++) {
        const uint32_t k = k0 + part;
        const bool mism = !done && (k >= lim || lt[u] != pat[k]);
        const uint32_t g = (uint32_t)(__ballot(mism) >> (sub * kSubLanes)) & ((1u << kSubLanes) - 1u);
        if (!done) {
          if (g) { L = k0 + (uint32_t)__ffs((int)g) - 1u; done = true; }
          else k0 += kSubLanes;
        }
      }
      while (__ballot(!done)) {  // longer matches: step by step
        const uint32_t k = k0 + part;
        const bool mism = !done && (k >= lim || ix.tok[sv.off + k] != pat[k]);
        const uint32_t g = (uint32_t)(__ballot(mism) >> (sub * kSubLanes)) & ((1u << kSubLanes) - 1u);
        if (!done) {
          if (g) { L = k0 + (uint32_t)__ffs((int)g) - 1u; done = true; }
          else k0 += kSubLanes;
        }
      }
      // coverage: the representatives of the pattern's words present in the sentence
      unsigned long long seen = 0;
      if (act && L < p) {
        const unsigned long long kq = (unsigned long long)(ql + 1) << 32;
        auto probe = [&](uint32_t t) {
          const unsigned long long key = kq | t;
          uint32_t h = lf_hash(t, ql) & kSlotMask;
          unsigned long long kk = S.hkey[h];
          while (kk != key && kk != 0ull) { h = (h + 1) & kSlotMask; kk = S.hkey[h]; }
          if (kk == key) seen |= 1ull << S.hj0[h];
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
      // count_covered_words = sum over the representatives seen of the positions that carry their word: each lane of the group
      // adds up a quarter of the 64 positions
      uint32_t cover = 0;
      {
        uint32_t bits = (uint32_t)(seen >> (16 * part)) & 0xffffu;
        const uint32_t base = tok0 + 16 * part;
        while (bits) {
          const uint32_t j = (uint32_t)__ffs((int)bits) - 1u;
          bits &= bits - 1u;
          const uint32_t slot = base + j;
          cover += (S.cnt[slot >> 2] >> (8 * (slot & 3))) & 0xffu;
        }
#pragma unroll
        for (uint32_t d = 1; d < kSubLanes; d <<= 1) cover += __shfl_xor(cover, d, 64);
      }
      if (L >= p) cover = p;
      bool keep = act && part == 0;
      if (keep) {
        if (need != kNeedExact && need != 0) keep = cover >= need;
        // need == 0: sentence length outside the threshold table, the length filter is still to be applied
        else keep = !reject_len(p, sl, prm) && !reject_cover(p, sl, cover, prm);
      }
      const unsigned long long mask = __ballot(keep);
      if (keep) {
        SurvB kp;
        kp.off = sv.sid; kp.rem_slen = L | (ql << 16); kp.sid = sv.off - (sl - rem); kp.it_need = sl;
        S.keep[n_k + __popcll(mask & ((1ull << lane) - 1ull))] = kp;
      }
      n_k += __popcll(mask);
    };
    uint32_t hi_i = 0, hi_qi = 0;
    bool hv = false;
    SufRecB rb;
    auto load_block = [&](uint32_t e0) {
      const uint32_t e = e0 + lane;
      hv = e < H;
      hi_i = 0; hi_qi = 0;
      if (hv) {
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = LfShared<TB, HB>::kStarts / 2; step >= 1; step >>= 1)
          if (S.npref[k + step] <= e) k += step;  // entries past the last start hold the sentinel
        hi_i = S.nlo[k] + (e - S.npref[k]);
        hi_qi = S.nqi[k];
      }
      rb.off = 0; rb.rem = 0; rb.slen = 0; rb.prev = 0; rb.sid = 0;
      if (hv) rb = ix.suf_b[hi_i];
    };
    load_block(0);
    for (uint32_t e0 = 0; e0 < H;) {
      {
        const uint32_t ql = hi_qi >> 6, it = hi_qi & 63u;
        const uint32_t tok0 = S.q_tok0[ql], p = S.q_p[ql];
        const bool dominated = it > 0 && rb.rem < rb.slen && rb.prev == S.pat[tok0 + (it ? it - 1 : 0)];
        // the length filter: the tabulated threshold of the (clamped) sentence length, 0xffff = rejected (k_cmin)
        uint32_t cmin = 0xffffu;
        if (hv && !dominated) cmin = w.cmin_tab[(size_t)p * kCminRow + (rb.slen < kSigLenClamp ? rb.slen : kSigLenClamp)];
        const bool keep = hv && !dominated && cmin != 0xffffu;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
          SurvB sv;
          sv.off = rb.off; sv.rem_slen = (uint32_t)rb.rem | ((uint32_t)rb.slen << 16);
          sv.sid = rb.sid; sv.it_need = cmin | (it << 16) | (ql << 24);
          S.surv[n_s + __popcll(mask & ((1ull << lane) - 1ull))] = sv;
        }
        n_s += __popcll(mask);
      }
      e0 += kWave;
      const bool more = e0 < H;
      if (more) load_block(e0);  // the next block's records travel while this block's survivors are worked on
      wave_lds_sync();
      uint32_t j0 = 0;
      for (; j0 + kSubSurv <= n_s; j0 += kSubSurv) token_work(j0, kSubSurv);
      const uint32_t rest = n_s - j0;
      if (!more) {
        if (rest) token_work(j0, rest);
        n_s = 0;
      } else {
        if (j0) {  // carry the incomplete pass to the front
          SurvB tmp;
          if (lane < rest) tmp = S.surv[j0 + lane];
          wave_lds_sync();
          if (lane < rest) S.surv[lane] = tmp;
          wave_lds_sync();
        }
        n_s = rest;
      }
    }
    wave_lds_sync();

    // ---- reduce: where each query's kept hits lie (they are in query order) ----
    for (uint32_t i = lane; i < n_k; i += kWave) {
      const uint32_t ql = S.keep[i].rem_slen >> 16;
      if (i == 0 || (S.keep[i - 1].rem_slen >> 16) != ql) S.q_kbeg[ql] = (uint16_t)i;
      if (i + 1 == n_k || (S.keep[i + 1].rem_slen >> 16) != ql) S.q_kend[ql] = (uint16_t)(i + 1);
    }
    wave_lds_sync();
    const uint32_t kb_me = mine ? S.q_kbeg[lane] : 0u, ke_me = mine ? S.q_kend[lane] : 0u;
    const uint32_t kc_me = (mine && flat) ? ke_me - kb_me : 0u;
    const bool big_me = kc_me > (uint32_t)kWave;
    const uint32_t kscan = wave_incl_scan(big_me ? 0u : kc_me);
    const unsigned long long big_mask = __ballot(big_me);
    // a query that keeps more than one wave's worth: its hits go to the segment unreduced, k_cand_route takes it from there
    for (unsigned long long bm = big_mask; bm; bm &= bm - 1ull) {
      const uint32_t ql = (uint32_t)__ffsll((long long)bm) - 1u;
      const uint32_t kb = __shfl(kb_me, ql, 64), kc = __shfl(kc_me, ql, 64);
      Cand* seg = w.seg + S.q_seg0[ql];
      for (uint32_t i = lane; i < kc; i += kWave) {
        const SurvB kp = S.keep[kb + i];
        Cand c;
        c.sid = kp.off; c.longest = kp.rem_slen & 0xffffu; c.score = 0.f; c.penalty = 0.f;
        seg[i] = c;
      }
      if (lane == 0) w.q_kept[q0 + ql] = kc;
    }
    // the others, as many whole queries a round as fit one wave
    uint32_t r = a;
    while (r < a_end) {
      const uint32_t base = __shfl(kscan - (big_me ? 0u : kc_me), r, 64);
      const bool fit_r = lane >= r && lane < a_end && kscan - base <= (uint32_t)kWave;
      const uint32_t n_r = __popcll(__ballot(fit_r));      // queries of this round: [r, r + n_r)
      const uint32_t tot = __shfl(kscan, r + n_r - 1, 64) - base;  // their kept hits (big queries count as none)
      if (tot != 0) {
        // the round's entries: query by query, in keep[] order.  Lane l takes entry l: its query = the one whose running count covers l
        uint32_t ql, ent;
        {
          // the query of entry l = the first one of the round whose running count exceeds l (at most 32 queries: five halvings,
          // every lane takes every step — a shuffle may not sit in divergent control flow)
          uint32_t lo_q = r, hi_q = r + n_r;
#pragma unroll
          for (int s = 0; s < 6; s++) {
            const uint32_t mid = (lo_q + hi_q) >> 1;
            const uint32_t ks = __shfl(kscan, mid & 63u, 64) - base;
            if (lo_q < hi_q) { if (ks > lane) hi_q = mid; else lo_q = mid + 1; }
          }
          ql = lo_q < r + n_r ? lo_q : r + n_r - 1;
          const uint32_t ks_prev = __shfl(kscan - (big_me ? 0u : kc_me), ql, 64) - base;
          ent = __shfl(kb_me, ql, 64) + (lane - ks_prev);
        }
        const bool live = lane < tot;
        SurvB kp;
        kp.off = 0; kp.rem_slen = 0; kp.sid = 0; kp.it_need = 0;
        if (live) kp = S.keep[ent];
        // (query 6 | sentence 32 | match length 16 | lane 6): the last entry of a (query, sentence) run carries its longest match
        unsigned long long key = live ? (((unsigned long long)ql << 54) | ((unsigned long long)kp.off << 22) |
                                         ((unsigned long long)(kp.rem_slen & 0xffffu) << 6) | lane) : ~0ull;
        if (tot > 1) key = wave_sort64_upto(key, tot);
        const uint32_t nlo = __shfl_down((uint32_t)key, 1, 64), nhi = __shfl_down((uint32_t)(key >> 32), 1, 64);
        const unsigned long long nxt = lane == 63 ? ~0ull : (((unsigned long long)nhi << 32) | nlo);
        const bool last = lane < tot && (nxt >> 22) != (key >> 22);
        // (query 6 | 0xffff - length 16 | sentence 32 | slot 6): get_longest_matches' order inside every query
        unsigned long long k2 = last ? (((key >> 54) << 54) | ((unsigned long long)(0xffffu - (uint32_t)((key >> 6) & 0xffffu)) << 38) |
                                        (((key >> 22) & 0xffffffffull) << 6) | (key & 63ull)) : ~0ull;
        const uint32_t n_c = __popcll(__ballot(last));
        if (n_c > 1) k2 = wave_sort64_upto(k2, tot);
        else if (n_c == 1) {
          const int src = __ffsll((long long)__ballot(last)) - 1;
          k2 = ((unsigned long long)__shfl((uint32_t)(k2 >> 32), src, 64) << 32) | __shfl((uint32_t)k2, src, 64);
        }
        // position inside the query: lanes since the head of the query's run
        const bool cl = lane < n_c;
        const uint32_t cq = cl ? (uint32_t)(k2 >> 54) : r;
        const uint32_t pq = __shfl_up(cq, 1, 64);
        const bool head = cl && (lane == 0 || pq != cq);
        const unsigned long long heads = __ballot(head);
        // where the query's entries of this round start (every lane shuffles: the source lanes must be active)
        const uint32_t s_prev = __shfl(kscan - (big_me ? 0u : kc_me), cq, 64) - base;
        const uint32_t s_kb = __shfl(kb_me, cq, 64);
        if (cl) {
          const unsigned long long upto = heads & (lane == 63 ? ~0ull : ((2ull << lane) - 1ull));
          const uint32_t first = 63u - (uint32_t)__clzll((long long)upto);
          const uint32_t slot = (uint32_t)(k2 & 63ull);  // the lane that loaded the entry in this round
          const SurvB src = S.keep[s_kb + (slot - s_prev)];
          Cand c;
          c.sid = (uint32_t)((k2 >> 6) & 0xffffffffull);
          // the edit-distance kernels find the sentence without going through sent_pos: token base in `score`, sentence length
          // beside the match length (like k_expand_b's in-place reduce)
          c.longest = (0xffffu - (uint32_t)((k2 >> 38) & 0xffffu)) | (src.it_need << 16);
          c.score = __uint_as_float(src.sid); c.penalty = 0.f;
          w.seg[S.q_seg0[cq] + (lane - first)] = c;
          if (head) {
            const unsigned long long above = heads & ~((2ull << lane) - 1ull);
            const uint32_t nxt_head = (lane == 63 || above == 0ull) ? n_c : (uint32_t)__ffsll((long long)above) - 1u;
            w.q_cand[q0 + cq] = nxt_head - lane;
          }
        }
      }
      // every flat query of the round, empty ones included: reduced (k_cand_route has nothing to do)
      if (lane >= r && lane < r + n_r && flat && !big_me) {
        w.q_kept[qme] = kc_me | kKeptReduced;
        if (kc_me == 0) w.q_cand[qme] = 0;
      }
      r += n_r;
    }
    if (mine && flat) w.q_pb[qme] = n_q;  // statistics: the hits walked here instead of by phase A
    wave_lds_sync();
    a = a_end;
  }
}

}  // namespace fmx
