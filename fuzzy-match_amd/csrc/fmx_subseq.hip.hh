// fmx_subseq.hip.hh — kernels of FuzzyMatch::subsequence() (src/fuzzy_match.cc:250-365): the reference pops
// sub-n-grams of the pattern from a priority queue (by summed idf weight or by length), takes the first whose suffix
// range holds a usable sentence, and scores up to number_of_matches sentences of that range with the full edit
// distance.  Split here as
//   k_sub_extent  lane per (query, start): the longest sub-n-gram starting there that occurs in the index at all —
//                 every shorter one occurs too, every longer one has an empty range the reference's loop skips
//   (host)        the priority queue itself: O(p^2) float compares per query, host prep like the vocabulary lookup
//   k_sub_eval    wave per (query, chosen sub-n-gram): its suffix range in suffix-array order, 64 suffixes at a time:
//                 distinct unseen sentences, one full edit distance per lane (real-token / penalty-token terms, costs
//                 (1, 1, 1), rows in LDS), then the reference's in-order acceptance with its int-truncated bound
// Included by fmx_gpu.hip after fmx_kernels.hip.hh.
#pragma once
#include "fmx_kernels.hip.hh"

namespace fmx {

struct SubItem {
  uint32_t q;    // query of the batch
  uint32_t it;   // start of the sub-n-gram in the pattern
  uint32_t len;  // its length
};

// what the reference keeps across the sub-n-grams of one query (src/fuzzy_match.cc:292-296)
struct SubState {
  int32_t  md;        // max_distance (an int in the reference: costs are truncated when stored)
  uint32_t n_cand;    // candidates.size()
  uint32_t n_seen;    // candidates + perfect, the sentences not to look at again
  uint32_t pos;       // start of the winning sub-n-gram
  Rec      best;
  uint32_t overflow;  // seen list full
  uint32_t done;      // a perfect match ended the search (cost == 0)
};

__global__ void k_sub_tq(const uint32_t* q_off, uint32_t nq, uint32_t* t_q) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  for (uint32_t t = q_off[q]; t < q_off[q + 1]; t++) t_q[t] = q;
}

__global__ void __launch_bounds__(64) k_sub_extent(DevIndex ix, const uint32_t* q_off, const uint32_t* q_tok, const uint32_t* t_q,
                                                   uint32_t nt, uint16_t* t_len) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nt) return;
  const uint32_t q = t_q[t];
  const uint32_t end = q_off[q + 1];
  // a sub-n-gram never contains ｟unk｠ (weight -1 ends the inner loop, src/fuzzy_match.cc:284)
  uint32_t limit = 0;
  while (t + limit < end && q_tok[t + limit] != 1u && limit < 0xffffu) limit++;
  uint32_t a = 0, b = limit + 1;  // sub-n-grams of a tokens occur, of b tokens do not
  while (b - a > 1) {
    const uint32_t mid = a + ((b - a) >> 1);
    uint32_t lo = 0, hi = 0;
    mgram_range_hashed(ix.gram, ix.suf_key, ix.suf, ix.tok, ix.qva, ix.n_qva, q_tok + t, mid, lo, hi);
    if (hi > lo) a = mid; else b = mid;
  }
  t_len[t] = (uint16_t)a;
}

// dynamic LDS: (p_max + 1) x 64 floats (DP rows)
// GROWS: the DP rows live in HBM (g_rows: row_len x 64 floats per workgroup) instead of dynamic LDS — patterns of more than
// ~590 tokens (--max-tokens-in-pattern above the default, include/fuzzy/suffix_array_index.hh:15), like k_dp<FULL, true>.
template <bool GROWS>
__global__ void __launch_bounds__(kWave) k_sub_eval(DevIndex ix, const uint32_t* q_off, const uint32_t* q_tok, const uint32_t* q_real,
                                                    const uint32_t* q_itok, const SubItem* items, uint32_t n_items, SubState* states,
                                                    uint32_t* seen, uint32_t seen_cap, uint32_t nm, int no_perfect, float* g_rows,
                                                    uint32_t row_len) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_rows = GROWS ? g_rows + (size_t)blockIdx.x * row_len * kWave : (float*)smem;
  __shared__ float s_final[kWave], s_M[kWave];
  __shared__ uint32_t s_sid[kWave];
  __shared__ SubState s_st;
  const uint32_t lane = lane_id();
  for (uint32_t item = blockIdx.x; item < n_items; item += gridDim.x) {
    const uint32_t q = items[item].q, it = items[item].it, len = items[item].len;
    const uint32_t o0 = q_off[q], p = q_off[q + 1] - o0;
    const uint32_t* pat = q_tok + o0;
    uint32_t lo = 0, hi = 0;
    mgram_range_hashed(ix.gram, ix.suf_key, ix.suf, ix.tok, ix.qva, ix.n_qva, pat + it, len, lo, hi);
    __syncthreads();
    if (lane == 0) s_st = states[q];
    __syncthreads();
    uint32_t* my_seen = seen + (size_t)q * seen_cap;
    DpSide side;
    side.s_real = nullptr; side.p_real = nullptr; side.s_itok = nullptr; side.p_itok = nullptr;
    side.itok_dist = nullptr; side.n_itok = 0; side.p_idf = nullptr; side.idf_weight = 0.f;  // idf_weight = 0 (:325)
    for (uint32_t c = lo; c < hi; c += kWave) {
      const uint32_t n_cand = s_st.n_cand, n_seen = s_st.n_seen;
      if (n_cand >= nm || s_st.done || s_st.overflow) break;  // candidates.size() < number_of_matches (:312)
      const uint32_t i = c + lane;
      const bool valid = i < hi;
      const uint32_t sid = valid ? ix.suf_b[i].sid : 0xffffffffu;
      bool dup = !valid;
      for (int d = 0; d < kWave; d++) {  // the same sentence earlier in this chunk
        const uint32_t other = __shfl(sid, d, 64);
        if ((uint32_t)d < lane && other == sid) dup = true;
      }
      for (uint32_t k = 0; k < n_seen && !dup; k++)  // candidates / perfect (:315-316); written by lane 0 in earlier chunks
        dup = __hip_atomic_load(&my_seen[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sid;
      float final_cost = 0.f, M = 0.f;
      if (!dup) {
        const uint32_t sp = ix.sent_pos[sid];
        const uint32_t slen = ix.tok[sp];
        const float dw = 100.f / normalizer(p, slen, 1.f, 1.f, 1.f);  // EditCosts() and Costs(p, s, ...) (:320-321)
        if (ix.real && q_real) { side.s_real = ix.real + sp + 1; side.p_real = q_real + o0; }
        if (ix.itok_dist) {
          side.itok_dist = ix.itok_dist;
          side.n_itok = ix.n_itok;
          side.s_itok = ix.itok ? ix.itok + sp : nullptr;
          side.p_itok = q_itok ? q_itok + o0 + q : nullptr;
        }
        final_cost = edit_distance<true>(ix.tok + sp + 1, slen, pat, p, 1.f, 1.f, 1.f, dw, side, FLT_MAX, LdsRow{s_rows + lane}, M);
      }
      s_final[lane] = final_cost; s_M[lane] = M; s_sid[lane] = dup ? 0xffffffffu : sid;
      __syncthreads();
      if (lane == 0) {
        SubState st = s_st;
        for (uint32_t l = 0; l < kWave && st.n_cand < nm && !st.done; l++) {
          const uint32_t sd = s_sid[l];
          if (sd == 0xffffffffu) continue;
          if (st.n_seen >= seen_cap) { st.overflow = 1; break; }
          const float ub = (float)st.md;
          // the row-minimum early exit returns a cost above the bound: never perfect, never accepted (:73-74 of
          // src/edit_distance.cc), but the sentence still counts as a candidate
          const bool early = s_M[l] > ub;
          const float cost = s_final[l];
          if (!early && cost == 0.f && no_perfect) { my_seen[st.n_seen++] = sd; continue; }  // perfect.insert (:329-332)
          if (!early && cost < ub) {                                                           // :333-351
            st.best.score = score_from_cost(cost);
            st.best.penalty = 0.f;
            st.best.max_subseq = (int32_t)len;
            st.best.s_id = sd;
            st.pos = it;
            st.md = (int)cost;
            if (cost == 0.f) { st.done = 1; break; }
          }
          my_seen[st.n_seen++] = sd;  // candidates.insert (:353)
          st.n_cand++;
        }
        s_st = st;
        __threadfence();
      }
      __syncthreads();
    }
    if (lane == 0) states[q] = s_st;
    __syncthreads();
  }
}

}  // namespace fmx
