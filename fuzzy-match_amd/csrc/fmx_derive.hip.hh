// fmx_derive.hip.hh — the derived accelerator arrays of DESIGN.md §3 built ON THE DEVICE from the reference's own
// four arrays (SuffixArray::_suffixes, _sentence_buffer, _sentence_pos: include/fuzzy/suffix_array.hh:50-61), so that
// an index — freshly sorted or read from a .fmi — is ready in about a second instead of the ~10 s the same loops take
// on the host (HostIndex::derive_heavy, kept as the test oracle of these kernels), and only ~1.3 GB instead of ~10 GB
// cross PCIe at C3.  Index build, not the hot path: every kernel is a gather / streaming pass; the hash tables are
// filled with one atomicCAS per distinct n-gram.
#pragma once
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "fmx_core.hh"

namespace fmx {

// word signature per sentence (fmx_core.hh: sig_bit, sig_half): sent_sig2[2 s + h] = half h of the 112-bucket signature;
// the 56-bit signature of suf_sig is the OR of the two
// The device copy of the sentence buffer keeps every sentence [len, tokens, 0] inside one 128-byte line when it fits in one
// (DeviceIndex::upload lays out the positions): a candidate's sentence then costs one line of HBM traffic instead of ~1.8.
// One thread per sentence copies its words of `src` (the token buffer, or a per-slot twin: real-token / penalty-token ids).
__global__ void __launch_bounds__(256) k_relayout(const uint32_t* __restrict__ len_buf, const uint32_t* __restrict__ len_pos,
                                                  const uint32_t* __restrict__ src, const uint32_t* __restrict__ old_pos,
                                                  const uint32_t* __restrict__ new_pos, uint32_t n_sent, uint32_t* __restrict__ dst) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_sent) return;
  const uint32_t n = len_buf[len_pos[s]] + 2u, a = old_pos[s], b = new_pos[s];
  for (uint32_t k = 0; k < n; k++) dst[b + k] = src[a + k];
}

__global__ void __launch_bounds__(256) k_dv_sent_sig(const uint32_t* __restrict__ buf, const uint32_t* __restrict__ sent_pos,
                                                     uint32_t n_sent, unsigned long long* __restrict__ sent_sig2) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_sent) return;
  const uint32_t sp = sent_pos[s], len = buf[sp];
  unsigned long long sg[2] = {0, 0};
  for (uint32_t k = 0; k < len; k++) { const uint32_t w = buf[sp + 1 + k]; sg[sig_half(w)] |= 1ull << sig_bit(w); }
  sent_sig2[2ull * s] = sg[0];
  sent_sig2[2ull * s + 1] = sg[1];
}

// per suffix: SufRec, SufRecB, packed signature word, the eight key tokens
__global__ void __launch_bounds__(256) k_dv_suffix(const uint32_t* __restrict__ buf, const uint32_t* __restrict__ sent_pos,
                                                   const uint32_t* __restrict__ suf_sid, const uint16_t* __restrict__ suf_pos,
                                                   const unsigned long long* __restrict__ sent_sig2, unsigned long long n,
                                                   SufRec* __restrict__ suf, SufRecB* __restrict__ suf_b,
                                                   unsigned long long* __restrict__ suf_sig, SufKey* __restrict__ suf_key) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t sid = suf_sid[i], pos = suf_pos[i];
  const uint32_t sp = sent_pos[sid], slen = buf[sp];
  SufRec r;
  r.off = sp + pos;
  r.slen = (uint16_t)slen;
  r.rem = (uint16_t)(slen - (pos - 1u));
  suf[i] = r;
  SufRecB rb;
  rb.off = r.off; rb.rem = r.rem; rb.slen = r.slen;
  rb.prev = r.rem < r.slen ? buf[r.off - 1] : 0u;
  rb.sid = sid;
  suf_b[i] = rb;
  suf_sig[i] = sig_pack(r.slen, sent_sig2[2ull * sid] | sent_sig2[2ull * sid + 1]);
  SufKey k;
#pragma unroll
  for (uint32_t t = 0; t < kKeyTokens; t++) k.t[t] = t + 1 < r.rem ? buf[r.off + 1 + t] : 0u;
  suf_key[i] = k;
}

// Group boundaries in suffix-array order.  A trigram group is a maximal run of suffixes with at least three tokens and
// the same first three; suffixes of one or two tokens sort in front of the longer ones that share their tokens, so
// they end a run as well.
struct DvTriBoundary {
  const SufRec* suf; const SufKey* key; const uint32_t* tok;
  __device__ bool operator()(uint32_t i) const {
    const SufRec r = suf[i];
    if (r.rem < 3 || i == 0) return true;
    const SufRec q = suf[i - 1];
    if (q.rem < 3) return true;
    return tok[r.off] != tok[q.off] || key[i].t[0] != key[i - 1].t[0] || key[i].t[1] != key[i - 1].t[1];
  }
};
// 5-gram groups inside the trigram groups: additionally cut at suffixes of fewer than five tokens and where (w3, w4) change
struct DvFiveBoundary {
  DvTriBoundary tri;
  __device__ bool operator()(uint32_t i) const {
    if (tri(i)) return true;
    if (tri.suf[i].rem < 5 || tri.suf[i - 1].rem < 5) return true;
    return tri.key[i].t[2] != tri.key[i - 1].t[2] || tri.key[i].t[3] != tri.key[i - 1].t[3];
  }
};

// (grid-stride, one atomic per wave at the end: one atomic per 64 boundaries on a single address was 6 ms at C3)
__global__ void __launch_bounds__(256) k_dv_count_starts(const uint32_t* __restrict__ bnd, uint32_t n_bnd, const SufRec* __restrict__ suf,
                                                         uint32_t min_rem, unsigned long long* __restrict__ count) {
  unsigned long long mine = 0;
  for (unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; k < n_bnd; k += (unsigned long long)gridDim.x * blockDim.x)
    mine += suf[bnd[k]].rem >= min_rem ? 1ull : 0ull;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) mine += __shfl_xor(mine, d, 64);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

__device__ __forceinline__ void dv_insert(GramEntry* g, uint32_t mask, uint32_t h, uint32_t w1, uint32_t w2, uint32_t lo, uint32_t cnt) {
  uint32_t slot = h & mask;
  for (;;) {
    if (atomicCAS(&g[slot].lo, kGramEmpty, lo) == kGramEmpty) { g[slot].w1 = w1; g[slot].w2 = w2; g[slot].cnt = cnt; return; }
    slot = (slot + 1) & mask;
  }
}

__global__ void __launch_bounds__(256) k_dv_fill_empty(GramEntry* g, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { GramEntry e; e.w1 = 0; e.w2 = 0; e.lo = kGramEmpty; e.cnt = 0; g[i] = e; }
}

// one thread per boundary: a boundary whose suffix has >= 3 tokens starts a trigram group that ends at the next boundary
__global__ void __launch_bounds__(256) k_dv_insert3(const uint32_t* __restrict__ bnd, uint32_t n_bnd, uint32_t n_suf,
                                                    const SufRec* __restrict__ suf, const SufKey* __restrict__ key,
                                                    const uint32_t* __restrict__ tok, GramEntry* g, uint32_t mask) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_bnd) return;
  const uint32_t i = bnd[k];
  const SufRec r = suf[i];
  if (r.rem < 3) return;
  const uint32_t e = k + 1 < n_bnd ? bnd[k + 1] : n_suf;
  const uint32_t w0 = tok[r.off], w1 = key[i].t[0], w2 = key[i].t[1];
  dv_insert(g, mask, gram_hash(w0, w1, w2), w1, w2, i, e - i);
}

// one thread per 5-gram boundary: lo3[k] = start of the enclosing trigram group if that group is deep (more than
// kDeepGramMin suffixes) and the boundary starts a 5-gram group, else kGramEmpty
__global__ void __launch_bounds__(256) k_dv_mark5(const uint32_t* __restrict__ bnd5, uint32_t n_bnd5, const uint32_t* __restrict__ bnd3,
                                                  uint32_t n_bnd3, uint32_t n_suf, const SufRec* __restrict__ suf,
                                                  uint32_t* __restrict__ lo3, unsigned long long* __restrict__ count) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  bool is = false;
  if (k < n_bnd5) {
    const uint32_t j = bnd5[k];
    uint32_t v = kGramEmpty;
    if (suf[j].rem >= 5) {
      // the enclosing trigram group: the last trigram boundary at or before j
      uint32_t a = 0, b = n_bnd3;  // bnd3[a] <= j < bnd3[b]
      while (b - a > 1) { const uint32_t mid = a + ((b - a) >> 1); if (bnd3[mid] <= j) a = mid; else b = mid; }
      const uint32_t g_lo = bnd3[a], g_hi = a + 1 < n_bnd3 ? bnd3[a + 1] : n_suf;
      if (g_hi - g_lo > kDeepGramMin) { v = g_lo; is = true; }
    }
    lo3[k] = v;
  }
  const unsigned long long m = __ballot(is);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

__global__ void __launch_bounds__(256) k_dv_insert5(const uint32_t* __restrict__ bnd5, uint32_t n_bnd5, uint32_t n_suf,
                                                    const uint32_t* __restrict__ lo3, const SufKey* __restrict__ key, GramEntry* g,
                                                    uint32_t mask) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_bnd5) return;
  const uint32_t g3 = lo3[k];
  if (g3 == kGramEmpty) return;
  const uint32_t j = bnd5[k];
  const uint32_t e = k + 1 < n_bnd5 ? bnd5[k + 1] : n_suf;
  const uint32_t w3 = key[j].t[2], w4 = key[j].t[3];
  dv_insert(g, mask, gram_hash(g3, w3, w4), w3, w4, j, e - j);
}

}  // namespace fmx
