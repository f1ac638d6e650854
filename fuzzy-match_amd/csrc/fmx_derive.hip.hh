// fmx_derive.hip.hh — the derived accelerator arrays of DESIGN.md §3 built ON THE DEVICE from the reference's own
// four arrays (SuffixArray::_suffixes, _sentence_buffer, _sentence_pos: include/fuzzy/suffix_array.hh:50-61), so that
// an index — freshly sorted or read from a .fmi — is ready in about a second instead of the ~10 s the same loops take
// on the host (HostIndex::derive_heavy, kept as the test oracle of these kernels), and only ~1.3 GB instead of ~10 GB
// cross PCIe at C3.  Index build, not the hot path: every kernel is a gather / streaming pass; the hash tables are
// filled with one atomicCAS per distinct n-gram.
#pragma once
#include <hip/hip_runtime.h>

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

// 7-gram groups inside the 5-gram groups: additionally cut at suffixes of fewer than seven tokens and where (w5, w6) change
struct DvSevenBoundary {
  DvFiveBoundary five;
  __device__ bool operator()(uint32_t i) const {
    if (five(i)) return true;
    if (five.tri.suf[i].rem < 7 || five.tri.suf[i - 1].rem < 7) return true;
    return five.tri.key[i].t[4] != five.tri.key[i - 1].t[4] || five.tri.key[i].t[5] != five.tri.key[i - 1].t[5];
  }
};

// Stream compaction of the indices with pred(i) (DeviceIndex::derive's `select`): a wave per 64 consecutive indices writes its ballot
// and its count (chunk n_ch, behind the last, gets zeros: the scan's extra element = the total); k_scan_lb<1> turns the counts into
// offsets; k_dv_scatter writes the indices in order.
template <class Pred>
__global__ void __launch_bounds__(64) k_dv_mark(Pred pred, unsigned long long n, uint32_t n_ch, unsigned long long* __restrict__ flags,
                                                unsigned long long* __restrict__ cnt) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t ch = blockIdx.x; ch <= n_ch; ch += gridDim.x) {
    const unsigned long long i = (unsigned long long)ch * 64 + lane;
    const unsigned long long fl = __ballot(i < n && pred((uint32_t)i));
    if (lane == 0) { flags[ch] = fl; cnt[ch] = (unsigned long long)__popcll(fl); }
  }
}
__global__ void __launch_bounds__(64) k_dv_scatter(unsigned long long n, const unsigned long long* __restrict__ flags,
                                                   const unsigned long long* __restrict__ offs, uint32_t* __restrict__ out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t n_ch = (uint32_t)((n + 63) / 64);
  for (uint32_t ch = blockIdx.x; ch < n_ch; ch += gridDim.x) {
    const unsigned long long fl = flags[ch];
    if ((fl >> lane) & 1ull) out[offs[ch] + (unsigned long long)__popcll(fl & ((1ull << lane) - 1ull))] = ch * 64u + lane;
  }
}

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

// one thread per 5-gram boundary: a boundary whose suffix has >= 5 tokens starts a 5-gram group that ends at the next boundary.
// Its entry goes into the first bucket with a free slot from its home bucket on (fmx_core.hh: G5Bucket): tokens 1..4, the range,
// tokens 5..8 of the first suffix and how many of them the last suffix shares (= what the whole sorted group shares).
__global__ void __launch_bounds__(256) k_dv_insert5x(const uint32_t* __restrict__ bnd5, uint32_t n_bnd5, uint32_t n_suf,
                                                     const SufRec* __restrict__ suf, const SufKey* __restrict__ key,
                                                     const uint32_t* __restrict__ tok, G5Bucket* g, uint32_t n_buckets) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_bnd5) return;
  const uint32_t i = bnd5[k];
  const SufRec r = suf[i];
  if (r.rem < 5) return;
  const uint32_t e = k + 1 < n_bnd5 ? bnd5[k + 1] : n_suf;
  const SufKey first = key[i], last = key[e - 1];
  uint32_t lcp = 0;
  while (lcp < 4 && first.t[4 + lcp] == last.t[4 + lcp]) lcp++;
  const uint32_t w0 = tok[r.off];
  uint32_t b = g5_bucket_of(gram5_hash(w0, first.t[0], first.t[1], first.t[2], first.t[3]), n_buckets);
  const uint32_t tag = gram5_tag(w0, first.t[0], first.t[1], first.t[2], first.t[3]);
  for (;;) {
    // claim a slot: entries in use = bits 16..17 of dir_b (compare-and-swap: the count never passes three)
    uint32_t old = __hip_atomic_load(&g[b].dir_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), slot = kG5PerBucket;
    while (g5_used(old) < kG5PerBucket) {
      const uint32_t seen = atomicCAS(&g[b].dir_b, old, old + (1u << 16));
      if (seen == old) { slot = g5_used(old); break; }
      old = seen;
    }
    if (slot < kG5PerBucket) {
      G5Entry en;
      en.w1 = first.t[0]; en.w2 = first.t[1]; en.w3 = first.t[2]; en.w4 = first.t[3];
      en.lo = i; en.cnt = e - i;
      en.t[0] = first.t[4]; en.t[1] = first.t[5]; en.t[2] = first.t[6]; en.t[3] = first.t[7];
      g[b].e[slot] = en;
      if (slot < 2) atomicOr(&g[b].dir_a, tag << (16 * slot));
      atomicOr(&g[b].dir_b, (slot == 2 ? tag : 0u) | (lcp << (20 + 3 * slot)));
      return;
    }
    atomicOr(&g[b].dir_b, 1u << 18);
    b = b + 1 == n_buckets ? 0u : b + 1;
  }
}

// Second level.  One thread per child boundary (7-gram boundaries under the 5-gram groups): lo_parent[k] = start of the
// enclosing parent group if that group is deep (more than kDeepGramMin suffixes; its suffixes have at least `parent_rem`
// tokens) and the boundary starts a child group (at least `child_rem` tokens), else kGramEmpty
__global__ void __launch_bounds__(256) k_dv_mark_deep(const uint32_t* __restrict__ bnd_c, uint32_t n_bnd_c, const uint32_t* __restrict__ bnd_p,
                                                      uint32_t n_bnd_p, uint32_t n_suf, const SufRec* __restrict__ suf, uint32_t parent_rem,
                                                      uint32_t child_rem, uint32_t* __restrict__ lo_parent, unsigned long long* __restrict__ count) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  bool is = false;
  if (k < n_bnd_c) {
    const uint32_t j = bnd_c[k];
    uint32_t v = kGramEmpty;
    if (suf[j].rem >= child_rem) {
      // the enclosing parent group: the last parent boundary at or before j
      uint32_t a = 0, b = n_bnd_p;  // bnd_p[a] <= j < bnd_p[b]
      while (b - a > 1) { const uint32_t mid = a + ((b - a) >> 1); if (bnd_p[mid] <= j) a = mid; else b = mid; }
      const uint32_t g_lo = bnd_p[a], g_hi = a + 1 < n_bnd_p ? bnd_p[a + 1] : n_suf;
      if (suf[g_lo].rem >= parent_rem && g_hi - g_lo > kDeepGramMin) { v = g_lo; is = true; }
    }
    lo_parent[k] = v;
  }
  const unsigned long long m = __ballot(is);
  if ((threadIdx.x & 63) == 0 && m) atomicAdd(count, (unsigned long long)__popcll(m));
}

// key_at = index into the suffix key of the child's two distinguishing tokens (7-grams under 5-grams: tokens 5 and 6 = t[4], t[5])
__global__ void __launch_bounds__(256) k_dv_insert_deep(const uint32_t* __restrict__ bnd_c, uint32_t n_bnd_c, uint32_t n_suf,
                                                        const uint32_t* __restrict__ lo_parent, const SufKey* __restrict__ key, uint32_t key_at,
                                                        GramEntry* g, uint32_t mask) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_bnd_c) return;
  const uint32_t gp = lo_parent[k];
  if (gp == kGramEmpty) return;
  const uint32_t j = bnd_c[k];
  const uint32_t e = k + 1 < n_bnd_c ? bnd_c[k + 1] : n_suf;
  const uint32_t wa = key[j].t[key_at], wb = key[j].t[key_at + 1];
  dv_insert(g, mask, gram_hash(gp, wa, wb), wa, wb, j, e - j);
}

}  // namespace fmx
