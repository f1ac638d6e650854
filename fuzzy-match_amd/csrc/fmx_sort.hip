// fmx_sort.hip — the suffix sorter of the index build, hand-written for gfx950.
//
// What is sorted: every token position of every indexed sentence (SuffixArray::add_sentence, src/suffix_array.cc:9-27),
// under SuffixArray::comp (src/suffix_array.cc:235-251): token by token, the shorter suffix first when one is a prefix of the
// other, equal suffixes by sentence id.  In other words a plain suffix sort of the sentence buffer in which every sentence ends
// in a terminator of its own that is smaller than every token and ordered by sentence id — so no comparison ever crosses a
// sentence boundary and no two suffixes are equal.
//
// How: prefix doubling on ranks, every round one LSD radix sort (8-bit digits, written here) of the suffixes that are still tied.
//   round 0   key = the first k tokens of the suffix packed into 64 bits (token + 1; 0 past the sentence's end); k = 64 / bits
//             of the largest token.  Sorted; a suffix's rank is the index of the first suffix with its key; a suffix whose key
//             nobody shares is done.
//   round t   (h = k, 2k, 4k, ... tokens are ranked so far) for the suffixes not done, in text order:
//             key = rank << b | second, second = the sentence id when the suffix has h tokens or fewer (what follows is its own
//             terminator), else n_sent + rank of the suffix h tokens further on.  Sorted; inside a run of equal old ranks the
//             new rank is the old one plus the number of elements in front of the first element with the same key.
//   The ranks live in one array indexed by buffer position (top bit: done), so building the keys is a streaming pass; the only
//   random traffic is the write of the new ranks and the final scatter of (sentence, position) by rank.
// At 5 M sentences (125 M suffixes) this is five rounds that put 2.33 x n elements through the radix sort: 50 ms.
//
// rocPRIM's merge sort under the comparator (what rounds 1-3 used) is kept for indexes the ranks do not fit 31 bits for.
#include "fmx_gpu.hh"

#include <string.h>
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace fmx {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      throw GpuError(std::string(#expr) + ": " + hipGetErrorString(e_));                            \
  } while (0)

namespace {

typedef unsigned long long u64;

constexpr uint32_t kDone = 0x80000000u;      // rank array: the suffix's rank is final
constexpr uint32_t kRankMask = 0x7fffffffu;
constexpr uint32_t kAbsent = 0xffffffffu;

__device__ __forceinline__ uint32_t lane_of() { return threadIdx.x & 63; }

// exclusive sum over the threads of a workgroup of 64 * WAVES threads (s_w: WAVES words); every thread gets the total too
template <int WAVES>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* s_w, uint32_t* total) {
  const uint32_t lane = lane_of(), wv = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t n = __shfl_up(x, d, 64);
    if (lane >= (uint32_t)d) x += n;
  }
  if (lane == 63) s_w[wv] = x;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int j = 0; j < WAVES; j++) {
    const uint32_t c = s_w[j];
    if ((uint32_t)j < wv) base += c;
    tot += c;
  }
  __syncthreads();
  if (total) *total = tot;
  return base + x - v;
}

// ---------------------------------------------------------------------------------------------------------------------------
// LSD radix sort of (64-bit key, 32-bit value) pairs, 8 bits per pass.  A pass is three launches: digit counts per block (a
// block owns a contiguous run of tiles), the counts scanned per digit, and the scatter — stable: a wave ranks its elements of
// equal digit by eight ballots, in index order; the tile is put in digit order in LDS first, so that a digit's elements leave
// for HBM as one contiguous run.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kRsThreads = 256;
constexpr uint32_t kRsWaves = kRsThreads / 64;
#ifndef FMX_RS_ITEMS
#define FMX_RS_ITEMS 8
#endif
constexpr uint32_t kRsItems = FMX_RS_ITEMS;
constexpr uint32_t kRsTile = kRsThreads * kRsItems;
constexpr uint32_t kRsBins = 256;
constexpr uint32_t kRsMaxBlocks = 2048;

__global__ void __launch_bounds__(kRsThreads) k_rs_count(const u64* __restrict__ keys, u64 n, uint32_t shift, uint32_t tiles_per_block,
                                                        uint32_t n_blocks, uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_h[kRsBins];
  s_h[threadIdx.x] = 0;
  __syncthreads();
  const u64 lo = (u64)blockIdx.x * tiles_per_block * kRsTile;
  u64 hi = lo + (u64)tiles_per_block * kRsTile;
  if (hi > n) hi = n;
  u64 i = lo + threadIdx.x;
  for (; i + 3ull * kRsThreads < hi; i += 4ull * kRsThreads) {
    const u64 a = keys[i], b = keys[i + kRsThreads], c = keys[i + 2ull * kRsThreads], d = keys[i + 3ull * kRsThreads];
    atomicAdd(&s_h[(uint32_t)(a >> shift) & 255u], 1u);
    atomicAdd(&s_h[(uint32_t)(b >> shift) & 255u], 1u);
    atomicAdd(&s_h[(uint32_t)(c >> shift) & 255u], 1u);
    atomicAdd(&s_h[(uint32_t)(d >> shift) & 255u], 1u);
  }
  for (; i < hi; i += kRsThreads) atomicAdd(&s_h[(uint32_t)(keys[i] >> shift) & 255u], 1u);
  __syncthreads();
  hist[(u64)threadIdx.x * n_blocks + blockIdx.x] = s_h[threadIdx.x];
}

// one workgroup per digit: the digit's per-block counts -> exclusive prefix (in place), the digit's total
__global__ void __launch_bounds__(kRsThreads) k_rs_rowscan(uint32_t* __restrict__ hist, uint32_t n_blocks, uint32_t* __restrict__ rowtotal) {
  __shared__ uint32_t s_w[kRsWaves];
  uint32_t* row = hist + (u64)blockIdx.x * n_blocks;
  uint32_t carry = 0;
  for (uint32_t b0 = 0; b0 < n_blocks; b0 += kRsThreads) {
    const uint32_t b = b0 + threadIdx.x;
    const uint32_t v = b < n_blocks ? row[b] : 0u;
    uint32_t tot;
    const uint32_t ex = block_excl_sum<kRsWaves>(v, s_w, &tot);
    if (b < n_blocks) row[b] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) rowtotal[blockIdx.x] = carry;
}

__global__ void __launch_bounds__(kRsThreads) k_rs_scatter(const u64* __restrict__ kin, const uint32_t* __restrict__ vin, u64* __restrict__ kout,
                                                          uint32_t* __restrict__ vout, u64 n, uint32_t shift, uint32_t tiles_per_block,
                                                          uint32_t n_blocks, const uint32_t* __restrict__ hist, const uint32_t* __restrict__ rowtotal) {
  __shared__ uint32_t s_cnt[kRsWaves][kRsBins];  // the wave's count of the digit in this tile, then the tile slot of its first element of the digit
  __shared__ uint32_t s_gbase[kRsBins];          // where the block's next element of the digit goes
  __shared__ uint32_t s_gdelta[kRsBins];         // global index = s_gdelta[digit] + tile slot (modulo 2^32)
  __shared__ uint32_t s_w[kRsWaves];
  __shared__ u64 s_key[kRsTile];
  __shared__ uint32_t s_val[kRsTile];
  const uint32_t t = threadIdx.x, lane = t & 63u, wv = t >> 6;
  {
    const uint32_t ex = block_excl_sum<kRsWaves>(rowtotal[t], s_w, nullptr);
    s_gbase[t] = ex + hist[(u64)t * n_blocks + blockIdx.x];
  }
  const u64 lo = (u64)blockIdx.x * tiles_per_block * kRsTile;
  const u64 below = lane ? (~0ull >> (64 - lane)) : 0ull;
  for (uint32_t tile = 0; tile < tiles_per_block; tile++) {
    const u64 tb = lo + (u64)tile * kRsTile;
    if (tb >= n) break;
    const uint32_t valid = (uint32_t)(n - tb < (u64)kRsTile ? n - tb : (u64)kRsTile);
#pragma unroll
    for (uint32_t j = 0; j < kRsWaves; j++) s_cnt[j][t] = 0;
    __syncthreads();
    u64 key[kRsItems];
    uint32_t val[kRsItems], rank[kRsItems];
    const uint32_t wbase = wv * (64u * kRsItems);
#pragma unroll
    for (uint32_t i = 0; i < kRsItems; i++) {
      const uint32_t idx = wbase + i * 64u + lane;
      const bool ok = idx < valid;
      key[i] = ok ? kin[tb + idx] : ~0ull;
      val[i] = ok ? vin[tb + idx] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < kRsItems; i++) {
      const uint32_t idx = wbase + i * 64u + lane;
      const bool ok = idx < valid;
      const uint32_t d = (uint32_t)(key[i] >> shift) & 255u;
      u64 m = __ballot(ok);
#pragma unroll
      for (uint32_t bit = 0; bit < 8; bit++) {
        const bool one = (d >> bit) & 1u;
        const u64 bal = __ballot(one);
        m &= one ? bal : ~bal;
      }
      const uint32_t pre = s_cnt[wv][d];
      rank[i] = pre + (uint32_t)__popcll(m & below);
      if (ok && (m >> lane) == 1ull) s_cnt[wv][d] = pre + (uint32_t)__popcll(m);  // the highest lane of the digit's set
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    {
      const uint32_t c0 = s_cnt[0][t], c1 = s_cnt[1][t], c2 = s_cnt[2][t], c3 = s_cnt[3][t];
      const uint32_t tot = c0 + c1 + c2 + c3;
      const uint32_t st = block_excl_sum<kRsWaves>(tot, s_w, nullptr);
      s_cnt[0][t] = st; s_cnt[1][t] = st + c0; s_cnt[2][t] = st + c0 + c1; s_cnt[3][t] = st + c0 + c1 + c2;
      s_gdelta[t] = s_gbase[t] - st;
      s_gbase[t] += tot;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t i = 0; i < kRsItems; i++) {
      const uint32_t idx = wbase + i * 64u + lane;
      if (idx < valid) {
        const uint32_t d = (uint32_t)(key[i] >> shift) & 255u;
        const uint32_t slot = s_cnt[wv][d] + rank[i];
        s_key[slot] = key[i];
        s_val[slot] = val[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (uint32_t j = 0; j < kRsItems; j++) {
      const uint32_t s = j * kRsThreads + t;
      if (s < valid) {
        const u64 k = s_key[s];
        const uint32_t g = s_gdelta[(uint32_t)(k >> shift) & 255u] + s;
        kout[g] = k;
        vout[g] = s_val[s];
      }
    }
    __syncthreads();
  }
}
static_assert(kRsWaves == 4, "k_rs_scatter spells the four waves out");

// ---------------------------------------------------------------------------------------------------------------------------
// the rounds
// ---------------------------------------------------------------------------------------------------------------------------
constexpr uint32_t kSxThreads = 256;
constexpr uint32_t kSxWaves = kSxThreads / 64;
constexpr uint32_t kSxItems = 8;
constexpr uint32_t kSxTile = kSxThreads * kSxItems;

// per buffer word: its sentence, the tokens left from here to the sentence's end (0: not a token), rank "done" for the framing
// words; the largest token
__global__ void __launch_bounds__(256) k_sx_prepare(const uint32_t* __restrict__ buf, const uint32_t* __restrict__ sent_pos, uint32_t n_sent, u64 n_buf,
                                                    uint32_t* __restrict__ sidw, uint16_t* __restrict__ remw, uint32_t* __restrict__ isa,
                                                    uint32_t* __restrict__ max_tok) {
  const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t mine = 0;
  if (w < n_buf) {
    uint32_t lo = 0, hi = n_sent;  // last sentence whose position is <= w
    while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (sent_pos[mid] <= w) lo = mid; else hi = mid; }
    const uint32_t sp = sent_pos[lo], len = buf[sp];
    const uint32_t pos = (uint32_t)(w - sp);  // 0 = the length word, len + 1 = the terminator
    const bool tok = pos >= 1 && pos <= len;
    sidw[w] = lo;
    remw[w] = tok ? (uint16_t)(len - pos + 1) : (uint16_t)0;
    isa[w] = kDone;
    if (tok) mine = buf[w];
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(mine, d, 64); mine = o > mine ? o : mine; }
  // one atomic per wave on one word is 1.7 M serialised atomics at 5 M sentences (19 ms): ask only when the wave would raise it
  if (lane_of() == 0 && mine > *(volatile uint32_t*)max_tok) atomicMax(max_tok, mine);
}

// round 0: the first k tokens of every suffix, packed; suffixes numbered in buffer order (buffer words minus the two framing
// words of every sentence before)
__global__ void __launch_bounds__(256) k_sx_key0(const uint32_t* __restrict__ buf, const uint32_t* __restrict__ sidw, const uint16_t* __restrict__ remw,
                                                 u64 n_buf, uint32_t bits, uint32_t k, u64* __restrict__ keys, uint32_t* __restrict__ vals) {
  const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_buf) return;
  const uint32_t rem = remw[w];
  if (!rem) return;
  u64 key = 0;
  for (uint32_t j = 0; j < k; j++) key = (key << bits) | (j < rem ? (u64)buf[w + j] + 1ull : 0ull);
  const u64 i = w - 2ull * sidw[w] - 1ull;
  keys[i] = key;
  vals[i] = (uint32_t)w;
}

// Over the sorted (key, position) pairs of a round.  An element starts a run when its old rank (the key's high part) differs
// from its predecessor's, and a group when its key does; new rank = old rank + (index of its group's head - index of its run's
// head).  Three launches: the last run head and group head of every tile, their running maxima over the tiles, the ranks.
template <bool FIRST>
__global__ void __launch_bounds__(kSxThreads) k_sx_heads(const u64* __restrict__ ks, uint32_t m, uint32_t b2, uint2* __restrict__ agg) {
  __shared__ uint32_t s_a[kSxWaves], s_b[kSxWaves];
  const uint32_t e0 = blockIdx.x * kSxTile + threadIdx.x * kSxItems;
  uint32_t rh = 0, nh = 0;
  if (e0 < m) {
    u64 prev = e0 ? ks[e0 - 1] : 0ull;
#pragma unroll
    for (uint32_t j = 0; j < kSxItems; j++) {
      const uint32_t e = e0 + j;
      if (e < m) {
        const u64 k = ks[e];
        if (e == 0 || k != prev) nh = e + 1;
        if (e == 0 || (!FIRST && (k >> b2) != (prev >> b2))) rh = e + 1;
        prev = k;
      }
    }
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    const uint32_t a = __shfl_xor(rh, d, 64), b = __shfl_xor(nh, d, 64);
    rh = a > rh ? a : rh; nh = b > nh ? b : nh;
  }
  if (lane_of() == 0) { s_a[threadIdx.x >> 6] = rh; s_b[threadIdx.x >> 6] = nh; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0, b = 0;
    for (uint32_t j = 0; j < kSxWaves; j++) { a = s_a[j] > a ? s_a[j] : a; b = s_b[j] > b ? s_b[j] : b; }
    agg[blockIdx.x] = make_uint2(a, b);
  }
}

// exclusive running maximum over the tiles (one workgroup)
__global__ void __launch_bounds__(1024) k_sx_heads_scan(uint2* __restrict__ agg, uint32_t n_tiles) {
  __shared__ uint32_t s_a[16], s_b[16];
  const uint32_t per = (n_tiles + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per < n_tiles ? threadIdx.x * per : n_tiles;
  const uint32_t hi = lo + per < n_tiles ? lo + per : n_tiles;
  uint32_t a = 0, b = 0;
  for (uint32_t i = lo; i < hi; i++) { const uint2 v = agg[i]; a = v.x > a ? v.x : a; b = v.y > b ? v.y : b; }
  const uint32_t lane = lane_of(), wv = threadIdx.x >> 6;
  uint32_t xa = a, xb = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t na = __shfl_up(xa, d, 64), nb = __shfl_up(xb, d, 64);
    if (lane >= (uint32_t)d) { xa = na > xa ? na : xa; xb = nb > xb ? nb : xb; }
  }
  if (lane == 63) { s_a[wv] = xa; s_b[wv] = xb; }
  __syncthreads();
  uint32_t ca = 0, cb = 0;  // maximum over the threads in front of this one
  for (uint32_t j = 0; j < wv; j++) { ca = s_a[j] > ca ? s_a[j] : ca; cb = s_b[j] > cb ? s_b[j] : cb; }
  {
    const uint32_t pa = __shfl_up(xa, 1, 64), pb = __shfl_up(xb, 1, 64);
    if (lane) { ca = pa > ca ? pa : ca; cb = pb > cb ? pb : cb; }
  }
  for (uint32_t i = lo; i < hi; i++) {
    const uint2 v = agg[i];
    agg[i] = make_uint2(ca, cb);
    ca = v.x > ca ? v.x : ca; cb = v.y > cb ? v.y : cb;
  }
}

template <bool FIRST>
__global__ void __launch_bounds__(kSxThreads) k_sx_regroup(const u64* __restrict__ ks, const uint32_t* __restrict__ ws, uint32_t m, uint32_t b2,
                                                          const uint2* __restrict__ agg, uint32_t* __restrict__ isa, uint32_t first_shift,
                                                          uint32_t* __restrict__ qva_start, uint32_t n_qva) {
  __shared__ uint32_t s_a[kSxWaves], s_b[kSxWaves];
  const uint32_t e0 = blockIdx.x * kSxTile + threadIdx.x * kSxItems;
  u64 k[kSxItems];
  u64 prev = 0, next = 0;
  uint32_t rh = 0, nh = 0;
  if (e0 < m) {
    prev = e0 ? ks[e0 - 1] : 0ull;
#pragma unroll
    for (uint32_t j = 0; j < kSxItems; j++) k[j] = e0 + j < m ? ks[e0 + j] : 0ull;
    next = e0 + kSxItems < m ? ks[e0 + kSxItems] : 0ull;
    u64 p = prev;
#pragma unroll
    for (uint32_t j = 0; j < kSxItems; j++) {
      const uint32_t e = e0 + j;
      if (e < m) {
        if (e == 0 || k[j] != p) nh = e + 1;
        if (e == 0 || (!FIRST && (k[j] >> b2) != (p >> b2))) rh = e + 1;
        p = k[j];
      }
    }
  }
  // running maxima over the threads in front of this one, and over the tiles in front of this tile
  const uint32_t lane = lane_of(), wv = threadIdx.x >> 6;
  uint32_t xa = rh, xb = nh;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t na = __shfl_up(xa, d, 64), nb = __shfl_up(xb, d, 64);
    if (lane >= (uint32_t)d) { xa = na > xa ? na : xa; xb = nb > xb ? nb : xb; }
  }
  if (lane == 63) { s_a[wv] = xa; s_b[wv] = xb; }
  __syncthreads();
  const uint2 carry = agg[blockIdx.x];
  uint32_t ca = carry.x, cb = carry.y;
  for (uint32_t j = 0; j < wv; j++) { ca = s_a[j] > ca ? s_a[j] : ca; cb = s_b[j] > cb ? s_b[j] : cb; }
  {
    const uint32_t pa = __shfl_up(xa, 1, 64), pb = __shfl_up(xb, 1, 64);
    if (lane) { ca = pa > ca ? pa : ca; cb = pb > cb ? pb : cb; }
  }
  if (e0 >= m) return;
  u64 p = prev;
#pragma unroll
  for (uint32_t j = 0; j < kSxItems; j++) {
    const uint32_t e = e0 + j;
    if (e < m) {
      const bool group_head = e == 0 || k[j] != p;
      if (group_head) cb = e + 1;
      if (e == 0 || (!FIRST && (k[j] >> b2) != (p >> b2))) ca = e + 1;
      const u64 nx = j + 1 < kSxItems ? k[j + 1] : next;
      const bool alone = group_head && (e + 1 == m || nx != k[j]);
      const uint32_t old_rank = FIRST ? 0u : (uint32_t)(k[j] >> b2);
      isa[ws[e]] = (old_rank + (cb - ca)) | (alone ? kDone : 0u);
      if (FIRST) {  // _quickVocabAccess: where the suffixes starting with this token begin (src/suffix_array.cc:76-98)
        const uint32_t w0 = (uint32_t)(k[j] >> first_shift) - 1u, p0 = (uint32_t)(p >> first_shift) - 1u;
        if ((e == 0 || w0 != p0) && w0 < n_qva) qva_start[w0] = e;
      }
      p = k[j];
    }
  }
}

// the suffixes not done, counted per tile of buffer words
__global__ void __launch_bounds__(kSxThreads) k_sx_active_count(const uint32_t* __restrict__ isa, u64 n_buf, uint32_t* __restrict__ cnt) {
  __shared__ uint32_t s_w[kSxWaves];
  const u64 w0 = (u64)blockIdx.x * kSxTile + (u64)threadIdx.x * kSxItems;
  uint32_t c = 0;
#pragma unroll
  for (uint32_t j = 0; j < kSxItems; j++)
    if (w0 + j < n_buf) c += (isa[w0 + j] & kDone) ? 0u : 1u;
  uint32_t tot;
  (void)block_excl_sum<kSxWaves>(c, s_w, &tot);
  if (threadIdx.x == 0) cnt[blockIdx.x] = tot;
}

// exclusive sums over the tiles (one workgroup); the total
__global__ void __launch_bounds__(1024) k_sx_sum_scan(uint32_t* __restrict__ cnt, uint32_t n_tiles, uint32_t* __restrict__ total) {
  __shared__ uint32_t s_w[16];
  const uint32_t per = (n_tiles + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per < n_tiles ? threadIdx.x * per : n_tiles;
  const uint32_t hi = lo + per < n_tiles ? lo + per : n_tiles;
  uint32_t s = 0;
  for (uint32_t i = lo; i < hi; i++) s += cnt[i];
  uint32_t tot;
  uint32_t base = block_excl_sum<16>(s, s_w, &tot);
  for (uint32_t i = lo; i < hi; i++) { const uint32_t v = cnt[i]; cnt[i] = base; base += v; }
  if (threadIdx.x == 0) *total = tot;
}

// round t: keys of the suffixes not done, in text order
__global__ void __launch_bounds__(kSxThreads) k_sx_make_keys(const uint32_t* __restrict__ isa, const uint32_t* __restrict__ sidw, const uint16_t* __restrict__ remw,
                                                            u64 n_buf, uint32_t n_sent, uint32_t h, uint32_t b2, const uint32_t* __restrict__ tile_base,
                                                            u64* __restrict__ keys, uint32_t* __restrict__ vals) {
  __shared__ uint32_t s_w[kSxWaves];
  const u64 w0 = (u64)blockIdx.x * kSxTile + (u64)threadIdx.x * kSxItems;
  uint32_t r[kSxItems];
  uint32_t c = 0;
#pragma unroll
  for (uint32_t j = 0; j < kSxItems; j++) {
    r[j] = w0 + j < n_buf ? isa[w0 + j] : kDone;
    c += (r[j] & kDone) ? 0u : 1u;
  }
  uint32_t o = tile_base[blockIdx.x] + block_excl_sum<kSxWaves>(c, s_w, nullptr);
  if (!c) return;
#pragma unroll
  for (uint32_t j = 0; j < kSxItems; j++) {
    if (r[j] & kDone) continue;
    const u64 w = w0 + j;
    const uint32_t rem = remw[w];
    const uint32_t second = h >= rem ? sidw[w] : n_sent + (isa[w + h] & kRankMask);
    keys[o] = ((u64)r[j] << b2) | (u64)second;
    vals[o] = (uint32_t)w;
    o++;
  }
}

// every suffix to its rank
__global__ void __launch_bounds__(256) k_sx_emit(const uint32_t* __restrict__ isa, const uint32_t* __restrict__ sidw, const uint16_t* __restrict__ remw,
                                                 const uint32_t* __restrict__ sent_pos, u64 n_buf, uint32_t* __restrict__ sid, uint16_t* __restrict__ pos) {
  const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_buf || !remw[w]) return;
  const uint32_t r = isa[w] & kRankMask, s = sidw[w];
  sid[r] = s;
  pos[r] = (uint16_t)(w - sent_pos[s]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// the merge sort of rounds 1-3 (rocPRIM under the comparator): for indexes whose ranks do not fit 31 bits, and FMX_SORTER=merge
// ---------------------------------------------------------------------------------------------------------------------------
struct GpuSortKey {
  u64 prefix;  // tok0 << 32 | tok1
  u64 key;     // sid << 16 | pos
};

struct GpuSuffixLess {
  const uint32_t* buf;
  const uint32_t* sent_pos;
  __device__ bool operator()(const GpuSortKey& a, const GpuSortKey& b) const {
    if (a.prefix != b.prefix) return a.prefix < b.prefix;
    const uint32_t sa = (uint32_t)(a.key >> 16), sb = (uint32_t)(b.key >> 16);
    const uint32_t pa = (uint32_t)(a.key & 0xffff), pb = (uint32_t)(b.key & 0xffff);
    const uint32_t* ba = buf + sent_pos[sa];
    const uint32_t* bb = buf + sent_pos[sb];
    const uint32_t la = ba[0] - (pa - 1), lb = bb[0] - (pb - 1);
    const uint32_t mn = la < lb ? la : lb;
    for (uint32_t i = 2; i < mn; i++) {
      const uint32_t x = ba[pa + i], y = bb[pb + i];
      if (x != y) return x < y;
    }
    if (la != lb) return la < lb;
    return sa < sb;
  }
};

__global__ void k_enumerate_suffixes(const uint32_t* __restrict__ buf, const uint32_t* __restrict__ sent_pos, uint32_t n_sent,
                                     u64 n_buf, GpuSortKey* __restrict__ keys) {
  const u64 w = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_buf) return;
  uint32_t lo = 0, hi = n_sent;
  while (hi - lo > 1) { const uint32_t mid = lo + ((hi - lo) >> 1); if (sent_pos[mid] <= w) lo = mid; else hi = mid; }
  const uint32_t sp = sent_pos[lo], len = buf[sp];
  const uint32_t pos = (uint32_t)(w - sp);
  if (pos == 0 || pos > len) return;
  const u64 i = w - 2ull * lo - 1ull;
  GpuSortKey k;
  k.prefix = ((u64)buf[w] << 32) | (pos < len ? buf[w + 1] : 0u);
  k.key = ((u64)lo << 16) | pos;
  keys[i] = k;
}

// sorted keys -> parallel arrays + where every first token's suffixes begin
__global__ void k_split_sorted(const GpuSortKey* __restrict__ keys, u64 n, uint32_t* __restrict__ sid, uint16_t* __restrict__ pos,
                               uint32_t* __restrict__ qva_start, uint32_t n_qva) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const GpuSortKey k = keys[i];
  sid[i] = (uint32_t)(k.key >> 16);
  pos[i] = (uint16_t)(k.key & 0xffffu);
  const uint32_t w0 = (uint32_t)(k.prefix >> 32);
  if ((i == 0 || (uint32_t)(keys[i - 1].prefix >> 32) != w0) && w0 < n_qva) qva_start[w0] = (uint32_t)i;
}

struct DevBuffers {  // everything the sort allocates and gives back
  std::vector<void*> p;
  template <class T> T* get(size_t n) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    p.push_back(q);
    return (T*)q;
  }
  ~DevBuffers() { for (void* q : p) (void)hipFree(q); }
};

uint32_t bit_length(u64 v) { uint32_t b = 0; while (v) { b++; v >>= 1; } return b; }

struct RadixSorter {
  u64* key[2];
  uint32_t* val[2];
  uint32_t* hist;
  uint32_t* rowtotal;
  // sorts the n pairs in (key[0], val[0]) by the low `bits` bits of the key; returns which of the two buffers holds the result
  int run(u64 n, uint32_t bits) {
    int cur = 0;
    if (n == 0) return cur;
    const u64 n_tiles = (n + kRsTile - 1) / kRsTile;
    const uint32_t tpb = (uint32_t)((n_tiles + kRsMaxBlocks - 1) / kRsMaxBlocks);
    const uint32_t n_blocks = (uint32_t)((n_tiles + tpb - 1) / tpb);
    for (uint32_t shift = 0; shift < bits; shift += 8) {
      hipLaunchKernelGGL(k_rs_count, dim3(n_blocks), dim3(kRsThreads), 0, 0, (const u64*)key[cur], n, shift, tpb, n_blocks, hist);
      hipLaunchKernelGGL(k_rs_rowscan, dim3(kRsBins), dim3(kRsThreads), 0, 0, hist, n_blocks, rowtotal);
      hipLaunchKernelGGL(k_rs_scatter, dim3(n_blocks), dim3(kRsThreads), 0, 0, (const u64*)key[cur], (const uint32_t*)val[cur], key[cur ^ 1], val[cur ^ 1], n,
                         shift, tpb, n_blocks, (const uint32_t*)hist, (const uint32_t*)rowtotal);
      cur ^= 1;
    }
    return cur;
  }
};

}  // namespace

void gpu_sort_resident_free(SortResident* r) {
  if (!r) return;
  if (r->device >= 0) (void)hipSetDevice(r->device);
  hipFree(r->d_buf); hipFree(r->d_sent_pos); hipFree(r->d_sid); hipFree(r->d_pos);
  delete r;
}

// Index build: the suffix array is enumerated, sorted and split on the device from the sentence buffer alone; the sorted arrays
// come home once (the .fmi writer and the host-side views need them) and ALSO stay in HBM for the upload that follows.
SortResident* gpu_suffix_sort(HostIndex& h, int device) {
  gpu_check_device(device);
  HIP_TRY(hipSetDevice(device));
  const uint64_t n = h.suf_sid.size();
  if (n == 0) { h.adopt_sorted(nullptr, 0); return nullptr; }
  const uint64_t n_buf = h.buf.size();
  const uint32_t n_sent = (uint32_t)h.sent_pos.size();
  std::unique_ptr<SortResident, void (*)(SortResident*)> res(new SortResident, gpu_sort_resident_free);
  res->device = device; res->n_buf = n_buf; res->n_sent = n_sent; res->n_suf = n;
  {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, (n_buf + kTokSpare) * sizeof(uint32_t)));
    res->d_buf = (uint32_t*)q;
    HIP_TRY(hipMemset(res->d_buf + n_buf, 0, kTokSpare * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(res->d_buf, h.buf.data(), n_buf * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&q, std::max<size_t>(n_sent, 1) * sizeof(uint32_t)));
    res->d_sent_pos = (uint32_t*)q;
    HIP_TRY(hipMemcpy(res->d_sent_pos, h.sent_pos.data(), (size_t)n_sent * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMalloc(&q, n * sizeof(uint32_t)));
    res->d_sid = (uint32_t*)q;
    HIP_TRY(hipMalloc(&q, n * sizeof(uint16_t)));
    res->d_pos = (uint16_t*)q;
  }
  const uint32_t vs = h.vocab_size;
  const uint32_t n_qva = std::max<uint32_t>(vs, 2);  // _quickVocabAccess: one entry per word id the index knows, plus the total
  const char* which = getenv("FMX_SORTER");
  const bool timing = getenv("FMX_SORT_TIMING") != nullptr;
  const uint32_t b1 = bit_length(n - 1 ? n - 1 : 1), b2 = bit_length(n + n_sent - 1);
  const bool merge = (which && std::string(which) == "merge") || n >= (1ull << 31) || n_buf >= 0xffff0000ull || b1 + b2 > 64;
  DevBuffers mem;
  uint32_t* d_qva = mem.get<uint32_t>(n_qva);
  HIP_TRY(hipMemset(d_qva, 0xff, (size_t)n_qva * 4));
  uint32_t max_tok = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  if (timing) { HIP_TRY(hipEventCreate(&ev0)); HIP_TRY(hipEventCreate(&ev1)); HIP_TRY(hipEventRecord(ev0, 0)); }
  uint32_t rounds = 0;
  u64 sorted_total = 0;
  if (merge) {
    GpuSortKey* d_in = mem.get<GpuSortKey>(n);
    GpuSortKey* d_out = mem.get<GpuSortKey>(n);
    hipLaunchKernelGGL(k_enumerate_suffixes, dim3((unsigned)((n_buf + 255) / 256)), dim3(256), 0, 0, res->d_buf, res->d_sent_pos, n_sent, (u64)n_buf, d_in);
    size_t tmp_bytes = 0;
    GpuSuffixLess less{res->d_buf, res->d_sent_pos};
    HIP_TRY(rocprim::merge_sort(nullptr, tmp_bytes, d_in, d_out, (size_t)n, less, (hipStream_t)0, false));
    void* tmp = mem.get<char>(tmp_bytes);
    HIP_TRY(rocprim::merge_sort(tmp, tmp_bytes, d_in, d_out, (size_t)n, less, (hipStream_t)0, false));
    hipLaunchKernelGGL(k_split_sorted, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const GpuSortKey*)d_out, (u64)n, res->d_sid, res->d_pos, d_qva, n_qva);
    HIP_TRY(hipGetLastError());
    max_tok = 0;  // the merge path learns of tokens beyond the vocabulary from the table below
  } else {
    uint32_t* d_sidw = mem.get<uint32_t>(n_buf);
    uint16_t* d_remw = mem.get<uint16_t>(n_buf);
    uint32_t* d_isa = mem.get<uint32_t>(n_buf);
    RadixSorter rs;
    rs.key[0] = mem.get<u64>(n); rs.key[1] = mem.get<u64>(n);
    rs.val[0] = mem.get<uint32_t>(n); rs.val[1] = mem.get<uint32_t>(n);
    rs.hist = mem.get<uint32_t>((size_t)kRsBins * kRsMaxBlocks);
    rs.rowtotal = mem.get<uint32_t>(kRsBins);
    const uint32_t n_tiles_buf = (uint32_t)((n_buf + kSxTile - 1) / kSxTile);
    const uint32_t n_tiles_suf = (uint32_t)((n + kSxTile - 1) / kSxTile);
    uint2* d_agg = mem.get<uint2>(n_tiles_suf);
    uint32_t* d_cnt = mem.get<uint32_t>(n_tiles_buf);
    uint32_t* d_scalars = mem.get<uint32_t>(2);  // [0] the largest token, [1] suffixes not done
    HIP_TRY(hipMemset(d_scalars, 0, 8));
    hipLaunchKernelGGL(k_sx_prepare, dim3((unsigned)((n_buf + 255) / 256)), dim3(256), 0, 0, res->d_buf, res->d_sent_pos, n_sent, (u64)n_buf, d_sidw, d_remw,
                       d_isa, d_scalars);
    HIP_TRY(hipMemcpy(&max_tok, d_scalars, 4, hipMemcpyDeviceToHost));
    const uint32_t bits = std::max<uint32_t>(1, bit_length((u64)max_tok + 1ull));
    const uint32_t k = std::min<uint32_t>(16, std::max<uint32_t>(1, 64 / bits));
    hipLaunchKernelGGL(k_sx_key0, dim3((unsigned)((n_buf + 255) / 256)), dim3(256), 0, 0, res->d_buf, d_sidw, d_remw, (u64)n_buf, bits, k, rs.key[0], rs.val[0]);
    int cur = rs.run(n, bits * k);
    sorted_total += n;
    auto regroup = [&](bool first, uint32_t m) {
      const uint32_t tiles = (m + kSxTile - 1) / kSxTile;
      if (first) {
        hipLaunchKernelGGL(k_sx_heads<true>, dim3(tiles), dim3(kSxThreads), 0, 0, (const u64*)rs.key[cur], m, b2, d_agg);
        hipLaunchKernelGGL(k_sx_heads_scan, dim3(1), dim3(1024), 0, 0, d_agg, tiles);
        hipLaunchKernelGGL(k_sx_regroup<true>, dim3(tiles), dim3(kSxThreads), 0, 0, (const u64*)rs.key[cur], (const uint32_t*)rs.val[cur], m, b2, (const uint2*)d_agg,
                           d_isa, bits * (k - 1), d_qva, n_qva);
      } else {
        hipLaunchKernelGGL(k_sx_heads<false>, dim3(tiles), dim3(kSxThreads), 0, 0, (const u64*)rs.key[cur], m, b2, d_agg);
        hipLaunchKernelGGL(k_sx_heads_scan, dim3(1), dim3(1024), 0, 0, d_agg, tiles);
        hipLaunchKernelGGL(k_sx_regroup<false>, dim3(tiles), dim3(kSxThreads), 0, 0, (const u64*)rs.key[cur], (const uint32_t*)rs.val[cur], m, b2,
                           (const uint2*)d_agg, d_isa, 0u, d_qva, 0u);
      }
    };
    regroup(true, (uint32_t)n);
    rounds = 1;
    for (u64 hh = k;; hh *= 2) {
      hipLaunchKernelGGL(k_sx_active_count, dim3(n_tiles_buf), dim3(kSxThreads), 0, 0, (const uint32_t*)d_isa, (u64)n_buf, d_cnt);
      hipLaunchKernelGGL(k_sx_sum_scan, dim3(1), dim3(1024), 0, 0, d_cnt, n_tiles_buf, d_scalars + 1);
      uint32_t active = 0;
      HIP_TRY(hipMemcpy(&active, d_scalars + 1, 4, hipMemcpyDeviceToHost));
      if (!active) break;
      if (rounds > 40) throw GpuError("suffix sort: the rounds do not end");
      if (cur != 0) { std::swap(rs.key[0], rs.key[1]); std::swap(rs.val[0], rs.val[1]); cur = 0; }
      const uint32_t h32 = (uint32_t)std::min<u64>(hh, 0x7fffffffull);
      hipLaunchKernelGGL(k_sx_make_keys, dim3(n_tiles_buf), dim3(kSxThreads), 0, 0, (const uint32_t*)d_isa, (const uint32_t*)d_sidw, (const uint16_t*)d_remw,
                         (u64)n_buf, n_sent, h32, b2, (const uint32_t*)d_cnt, rs.key[0], rs.val[0]);
      cur = rs.run(active, b1 + b2);
      regroup(false, active);
      sorted_total += active;
      rounds++;
    }
    hipLaunchKernelGGL(k_sx_emit, dim3((unsigned)((n_buf + 255) / 256)), dim3(256), 0, 0, (const uint32_t*)d_isa, (const uint32_t*)d_sidw, (const uint16_t*)d_remw,
                       (const uint32_t*)res->d_sent_pos, (u64)n_buf, res->d_sid, res->d_pos);
    HIP_TRY(hipGetLastError());
  }
  if (timing) {
    HIP_TRY(hipEventRecord(ev1, 0));
    HIP_TRY(hipEventSynchronize(ev1));
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, ev0, ev1));
    fprintf(stderr, "[fmx sort] %llu suffixes of %u sentences sorted on the device in %.2f ms (%s; %u rounds, %.2f x n elements through the radix sort)\n",
            (unsigned long long)n, n_sent, ms, merge ? "merge sort under the comparator" : "prefix doubling", rounds, (double)sorted_total / (double)n);
    (void)hipEventDestroy(ev0); (void)hipEventDestroy(ev1);
  }
  std::vector<uint32_t> sid(n), start(n_qva);
  std::vector<uint16_t> pos(n);
  HIP_TRY(hipMemcpy(sid.data(), res->d_sid, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(pos.data(), res->d_pos, n * 2, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(start.data(), d_qva, (size_t)n_qva * 4, hipMemcpyDeviceToHost));
  // word ids above the declared vocabulary (bulk corpora may carry them): the host's rule grows the vocabulary to fit
  bool beyond = !merge && max_tok >= n_qva;
  if (merge) {  // the first token of the last suffix is the largest one
    const uint32_t last = h.buf[(size_t)h.sent_pos[sid[n - 1]] + pos[n - 1]];
    beyond = last >= n_qva;
  }
  if (beyond) {
    std::vector<uint64_t> flat(n);
    for (uint64_t i = 0; i < n; i++) flat[i] = ((uint64_t)sid[i] << 16) | pos[i];
    h.adopt_sorted(flat.data(), n);
  } else {
    std::vector<uint32_t> qva((size_t)n_qva + 1, 0);  // first suffix per first-token id, [vocab] = total; absent ids: where the next one starts
    qva[n_qva] = (uint32_t)n;
    for (uint32_t w = n_qva; w-- > 0;) qva[w] = start[w] == kAbsent ? qva[w + 1] : start[w];
    h.adopt_sorted_arrays(std::move(sid), std::move(pos), std::move(qva));
  }
  return res.release();
}

}  // namespace fmx
