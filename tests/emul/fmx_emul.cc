// fmx_emul.cc — TEST-ONLY serial emulation of the HIP pipeline (fmx_kernels.hip.hh), built from the
// very same per-lane functions (fmx_core.hh).  It lets the CPU test-suite check, without a GPU,
// that the kernel decomposition (seed-n-gram range + LCP + dominance filter, (final, M) replay in
// 64-candidate chunks with a stale bound, key sorts) reproduces the oracle exactly.  It is not
// linked into libfmx.so and no product path can reach it.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../fuzzy-match_amd/csrc/fmx_core.hh"

using namespace fmx;

struct Rec { float score; float penalty; int32_t max_subseq; uint32_t s_id; };

struct EmulIndex {
  const SufRec* suf; const uint64_t* suf_sig; const SufKey* suf_key; const GramEntry* gram; const G5Bucket* gram5; const GramEntry* gram7; const SufRecB* suf_b; const uint32_t* tok; const uint32_t* sent_pos; const uint32_t* qva;
  const float* idf; const uint32_t* real; const uint32_t* itok; const uint16_t* itok_dist;
  uint32_t n_itok, n_qva, gram_mask, gram5_buckets, gram7_mask;
};

struct VecRow {
  float* base;
  float& operator[](uint32_t j) const { return base[j]; }
};

// the probe's range function alone (fmx_core.hh: mgram_range_hashed — n-gram tables, 5-gram buckets, keys), for n-grams of any length:
// ngrams = n x stride tokens, lens[k] = length of n-gram k; out = n x {lo, hi}
extern "C" void emul_ngram_ranges(const EmulIndex* ix, const uint32_t* ngrams, const uint32_t* lens, uint64_t n, uint32_t stride,
                                  uint32_t* out) {
  const GramTables gt{ix->gram, ix->gram5, ix->gram7, ix->gram_mask, ix->gram5_buckets, ix->gram7_mask};
  for (uint64_t k = 0; k < n; k++) {
    uint32_t lo = 0, hi = 0;
    mgram_range_hashed(gt, ix->suf_key, ix->suf, ix->tok, ix->qva, ix->n_qva, ngrams + k * stride, lens[k], lo, hi);
    out[2 * k] = lo; out[2 * k + 1] = hi;
  }
}

extern "C" int emul_match_batch(const EmulIndex* ix, const Params* prm_in, uint64_t nq, const uint32_t* q_off,
                                const uint32_t* q_tok, const uint32_t* q_real, const uint32_t* q_itok, Rec* out,
                                uint32_t* out_count, uint32_t* out_total, uint32_t cap, uint64_t* stats) {
  const Params prm = *prm_in;
  const bool full = (ix->real && q_real) || q_itok || ix->itok || prm.idf_penalty != 0.f;
  std::vector<Cand> seg;
  std::vector<float> row, heapv, idf;
  for (uint64_t q = 0; q < nq; q++) {
    const uint32_t o0 = q_off[q], p = q_off[q + 1] - o0;
    const uint32_t* pat = q_tok + o0;
    out_count[q] = 0;
    if (out_total) out_total[q] = 0;
    const uint32_t msl = min_seq_len(p, prm);
    if (msl == kNoMatchLen) continue;
    const uint32_t m = seed_len(p, msl);
    const bool reachable = (p == 1) ? (msl <= 1) : true;
    // k_probe + k_expand
    seg.clear();
    std::vector<uint8_t> hbit(p);
    for (uint32_t j = 0; j < p; j++) hbit[j] = (uint8_t)sig_bit(pat[j]);
    const bool use_sig = prm.ins >= 0.f && prm.del >= 0.f && prm.rep >= 0.f;
    for (uint32_t it = 0; it < p; it++) {
      uint32_t lo = 0, hi = 0;
      if (reachable && it + m <= p) mgram_range_hashed(GramTables{ix->gram, ix->gram5, ix->gram7, ix->gram_mask, ix->gram5_buckets, ix->gram7_mask}, ix->suf_key, ix->suf, ix->tok, ix->qva, ix->n_qva, pat + it, m, lo, hi);
      if (stats) stats[0] += hi - lo;
      for (uint32_t i = lo; i < hi; i++) {
        const SufRec r = ix->suf[i];
        // phase A reads only the packed (sentence length, signature) word
        const unsigned long long packed = ix->suf_sig[i];
        if (sig_slen(packed) != (r.slen < kSigLenClamp ? r.slen : kSigLenClamp)) return -2;
        const bool clamped = sig_slen(packed) == kSigLenClamp;  // the kernel's table entry 0: phase B evaluates everything
        const uint32_t sl = r.slen;
        if (!clamped && reject_len(p, sl, prm)) continue;
        // the kernel tabulates cover_threshold per sentence length and compares the signature's upper bound
        uint32_t cmin = (use_sig && !clamped) ? cover_threshold(p, sl, prm) : 0u;
        if (use_sig && !clamped && (cmin == 0xffffu || sig_cover_upper(sig_bits(packed), hbit.data(), p) < cmin)) continue;
        if (stats) stats[4]++;  // queued for phase B
        if (clamped) {
          if (reject_len(p, sl, prm)) continue;
          cmin = use_sig ? cover_threshold(p, sl, prm) : 0u;
          if (use_sig && cmin == 0xffffu) continue;
        }
        const SufRecB rb = ix->suf_b[i];
        if (rb.off != r.off || rb.rem != r.rem || rb.slen != r.slen) return -3;
        if (dominated_hit_b(rb, pat + it, it) != dominated_hit(ix->tok, r, pat + it, it)) return -4;
        if (dominated_hit_b(rb, pat + it, it)) continue;
        if (stats) stats[5]++;  // not dominated
        Cand c;
        c.longest = lcp_extend(ix->tok, r, pat + it, m, p - it);
        // coverage bound in k_expand: independent of the match length, so it commutes with the max
        const uint32_t* sent0 = ix->tok + (r.off - (uint32_t)(r.slen - r.rem));
        const uint32_t cov = c.longest < p ? cover_count(sent0, r.slen, pat, p) : p;
        if (use_sig ? cov < cmin : reject_cover(p, r.slen, cov, prm)) continue;
        c.sid = rb.sid;
        c.score = c.penalty = 0.f;
        seg.push_back(c);
      }
    }
    if (stats) stats[1] += seg.size();
    // k_cand: key sort by (sid, L), keep last of each group, then by (0xffff - L, sid)
    std::vector<uint64_t> keys(seg.size());
    for (size_t i = 0; i < seg.size(); i++) keys[i] = ((uint64_t)seg[i].sid << 16) | (seg[i].longest & 0xffffu);
    std::sort(keys.begin(), keys.end());
    std::vector<uint64_t> k2;
    for (size_t i = 0; i < keys.size(); i++)
      if (i + 1 >= keys.size() || (keys[i + 1] >> 16) != (keys[i] >> 16))
        k2.push_back(((uint64_t)(0xffffu - (uint32_t)(keys[i] & 0xffffu)) << 32) | (keys[i] >> 16));
    std::sort(k2.begin(), k2.end());
    const uint32_t n_cand = (uint32_t)k2.size();
    seg.resize(n_cand);
    for (uint32_t i = 0; i < n_cand; i++) {
      seg[i].sid = (uint32_t)(k2[i] & 0xffffffffu);
      seg[i].longest = 0xffffu - (uint32_t)(k2[i] >> 32);
    }
    if (stats) stats[2] += n_cand;
    // k_score
    row.assign((size_t)p + 1, 0.f);
    idf.assign(p, 0.f);
    const bool use_idf = prm.idf_penalty != 0.f;
    for (uint32_t j = 0; j < p; j++) {
      const uint32_t wd = pat[j];
      idf[j] = (use_idf && wd + 1 < ix->n_qva && wd != 1u) ? ix->idf[wd] : 0.f;
    }
    heapv.assign((size_t)n_cand + 2, 0.f);
    CostHeap heap;
    heap.v = heapv.data();
    heap.n = 0;
    heap.push(FLT_MAX);
    float s_ub = FLT_MAX;
    uint32_t acc = 0;
    std::vector<float> c_final(64), c_M(64);
    std::vector<uint32_t> c_pass(64), c_slen(64);
    for (uint32_t c0 = 0; c0 < n_cand; c0 += 64) {
      const uint32_t lim = std::min<uint32_t>(64, n_cand - c0);
      for (uint32_t k = 0; k < lim; k++) {
        const Cand& c = seg[c0 + k];
        const uint32_t sp = ix->sent_pos[c.sid];
        const uint32_t slen = ix->tok[sp];
        const uint32_t* sent = ix->tok + sp + 1;
        bool pass = true;
        float final_cost = 0.f, M = 0.f;
        if (pass) {
          const float dw = 100.f / normalizer(p, slen, prm.ins, prm.del, prm.rep);
          DpSide side;
          side.s_real = nullptr; side.p_real = nullptr; side.s_itok = nullptr; side.p_itok = nullptr;
          side.itok_dist = nullptr; side.n_itok = 0; side.p_idf = nullptr; side.idf_weight = 0.f;
          if (full) {
            if (ix->real) {
              side.s_real = ix->real + sp + 1;
              side.p_real = q_real ? q_real + o0 : nullptr;
              if (!side.p_real) side.s_real = nullptr;
            }
            if (ix->itok_dist) {
              side.itok_dist = ix->itok_dist;
              side.n_itok = ix->n_itok;
              side.s_itok = ix->itok ? ix->itok + sp : nullptr;
              side.p_itok = q_itok ? q_itok + o0 + q : nullptr;
            }
            if (use_idf) { side.p_idf = idf.data(); side.idf_weight = (dw * prm.idf_penalty) / prm.idf_max; }
            final_cost = edit_distance<true>(sent, slen, pat, p, prm.ins, prm.del, prm.rep, dw, side, s_ub,
                                             VecRow{row.data()}, M);
          } else {
            final_cost = edit_distance<false>(sent, slen, pat, p, prm.ins, prm.del, prm.rep, dw, side, s_ub,
                                              VecRow{row.data()}, M);
            // the kernels' unit-cost path: bit-parallel distance k, then k additions of the increment
            if (!use_idf && p <= 64 && prm.ins == prm.del && prm.del == prm.rep && prm.rep >= 0.f) {
              const float f = unit_cost_sum(lev_distance(sent, slen, pat, p), prm.rep * dw);
              if (f != final_cost || f != M) return -6;
            }
          }
          if (stats) stats[3]++;
        }
        c_final[k] = final_cost; c_M[k] = M; c_pass[k] = pass; c_slen[k] = slen;
      }
      for (uint32_t k = 0; k < lim; k++) {
        if (!c_pass[k]) continue;
        float score;
        if (replay_candidate(heap, prm, p, c_slen[k], c_final[k], c_M[k], score)) {
          Cand a = seg[c0 + k];
          a.score = score; a.penalty = 0.f;
          seg[acc++] = a;
        }
      }
      // k_dp evaluates every candidate without a bound (flattened over queries); the bound only acts in the replay
    }
    // k_select
    const uint32_t n = acc;
    if (n == 0) continue;
    std::vector<uint64_t> sk(n);
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) { sk[i] = ((uint64_t)(~float_key(seg[i].score)) << 32) | seg[i].sid; order[i] = i; }
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sk[a] < sk[b]; });
    const uint32_t want = prm.nmatch == 0 ? n : std::min(prm.nmatch, n);
    Rec* o = out + q * (uint64_t)cap;
    if (!(prm.contrast > 0.f)) {
      const uint32_t emit = std::min(want, cap);
      for (uint32_t k = 0; k < emit; k++) {
        const Cand& c = seg[order[k]];
        o[k] = Rec{c.score, 0.f, (int32_t)c.longest, c.sid};
      }
      out_count[q] = emit;
      if (out_total) out_total[q] = want;
      continue;
    }
    std::vector<float> pen_sum(n, 0.f), pen_cur(n, 0.f), pen_max(n, 0.f);
    std::vector<uint8_t> taken(n, 0);
    uint32_t picked = 0;
    int last = -1;
    std::vector<float> crow;
    DpSide none;
    none.s_real = nullptr; none.p_real = nullptr; none.s_itok = nullptr; none.p_itok = nullptr;
    none.itok_dist = nullptr; none.n_itok = 0; none.p_idf = nullptr; none.idf_weight = 0.f;
    while (picked < n && (prm.nmatch == 0 || picked < prm.nmatch)) {
      if (last >= 0) {
        const uint32_t msid = seg[order[last]].sid;
        const uint32_t msp = ix->sent_pos[msid];
        const uint32_t n2 = ix->tok[msp];
        crow.assign((size_t)n2 + 1, 0.f);
        for (uint32_t k = 0; k < n; k++) {
          if (taken[k]) continue;
          const uint32_t csp = ix->sent_pos[seg[order[k]].sid];
          const uint32_t n1 = ix->tok[csp];
          const float dw = 100.f / normalizer(n1, n2, 1.f, 1.f, 1.f);
          float M;
          const float ed = edit_distance<false>(ix->tok + csp + 1, n1, ix->tok + msp + 1, n2, 1.f, 1.f, 1.f, dw, none,
                                                FLT_MAX, VecRow{crow.data()}, M);
          if (n2 <= 64 && unit_cost_sum(lev_distance(ix->tok + csp + 1, n1, ix->tok + msp + 1, n2), 1.f * dw) != ed) return -7;
          const float pen = score_from_cost(ed);
          pen_sum[k] = pen_sum[k] + pen;
          pen_max[k] = (picked == 1) ? pen : (pen_max[k] < pen ? pen : pen_max[k]);
          pen_cur[k] = prm.reduce == 1 ? pen_max[k] : (pen_sum[k] / (float)picked);
        }
      }
      int best = -1;
      float best_key = 0.f;
      for (uint32_t k = 0; k < n; k++) {
        if (taken[k]) continue;
        const float key = seg[order[k]].score - (prm.contrast * pen_cur[k]);
        if (best < 0 || best_key < key) { best = (int)k; best_key = key; }
      }
      if (picked < cap) {
        const Cand& c = seg[order[best]];
        o[picked] = Rec{c.score, pen_cur[best], (int32_t)c.longest, c.sid};
      }
      taken[best] = 1;
      picked++;
      last = best;
    }
    out_count[q] = std::min(picked, cap);
    if (out_total) out_total[q] = picked;
  }
  return 0;
}
