// fmx_vocab.hip.hh — VocabIndexer::getIndex (src/vocab_indexer.cc:52-60; the per-pattern host prep of match(),
// src/fuzzy_match.cc:468-470) for a whole batch ON THE DEVICE: the batch arrives as token SPELLINGS — NUL-terminated, back to
// back, what a tokenizer writes — and leaves as vocabulary ids in HBM, in token order.  One kernel: every thread takes
// kVocabChunk bytes of the text, marks the token starts in them, the tile scans the counts (look-back over the tiles: the
// rank of a token is the number of starts in front of it) and every start is looked up in the device copy of the host's
// spelling table (HostIndex::FastVocab: open addressing, the first eight bytes beside every slot, the spellings in an arena;
// same hash).  Unknown spellings get VOCAB_UNK (1), like getIndex.
#pragma once
#include "fmx_kernels.hip.hh"

namespace fmx {

constexpr uint32_t kVocabChunk = 16;  // bytes of text per thread (64 bytes: thirteen look-ups in a row per thread, 0.60 ms per million queries; 16 or 32: 0.41)

struct DevVocab {
  const ulonglong2*    entry;  // {offset into arena << 32 | id (~0 = empty), the spelling's first eight bytes zero padded}: one load per probe
  const unsigned char* arena;  // NUL-terminated spellings
  unsigned long long   mask;
};

// text: n_bytes bytes followed by at least kVocabChunk + 16 zero bytes, 16-byte aligned.  ids: n_tok words.  total: the
// number of tokens found (the host compares it with the offsets it was given).
__global__ void __launch_bounds__(kScanThreads) k_vocab_lookup(const unsigned char* __restrict__ text, unsigned long long n_bytes,
                                                              unsigned long long n_tok, DevVocab v, uint32_t* __restrict__ ids,
                                                              unsigned long long* state, uint32_t* ticket, unsigned long long* total) {
  const uint32_t tile = scan_ticket_take(ticket);
  const unsigned long long b0 = ((unsigned long long)tile * kScanThreads + threadIdx.x) * kVocabChunk;
  unsigned long long starts = 0;
  if (b0 < n_bytes) {
    unsigned char prev = b0 ? text[b0 - 1] : (unsigned char)0;
#pragma unroll
    for (uint32_t w = 0; w < kVocabChunk / 16; w++) {
      const uint4 x = *reinterpret_cast<const uint4*>(text + b0 + 16 * w);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) {
        const unsigned char c = (unsigned char)(xs[k >> 2] >> (8 * (k & 3)));
        if (c != 0 && prev == 0 && b0 + 16 * w + k < n_bytes) starts |= 1ull << (16 * w + k);
        prev = c;
      }
    }
  }
  unsigned long long sum[1] = {(unsigned long long)__popcll(starts)}, ex[1];
  scan_tile_lb<1>(tile, sum, ex, state);
  unsigned long long rank = ex[0];
  if (b0 < n_bytes && b0 + kVocabChunk >= n_bytes) *total = rank + sum[0];  // the thread that holds the last byte
  while (starts) {
    const uint32_t i = (uint32_t)__builtin_ctzll(starts);
    starts &= starts - 1ull;
    const unsigned long long pos = b0 + i;
    const unsigned char* t = text + pos;
    // the spelling's first sixteen bytes by two aligned 16-byte loads and a byte shift (byte-by-byte loads are a chain of
    // dependent cache latencies: 0.73 ms per million queries); longer spellings continue byte by byte
    const uint4* al = reinterpret_cast<const uint4*>(text + (pos & ~15ull));
    const uint4 lo = al[0], hi = al[1];
    const uint32_t w[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    const uint32_t sh = (uint32_t)(pos & 15ull), ws = sh >> 2, bs = (sh & 3u) * 8u;
    uint32_t tw[4];
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
      uint32_t a = 0, b2 = 0;
#pragma unroll
      for (uint32_t j = 0; j < 5; j++) {  // w[ws + k], w[ws + k + 1] without a dynamic register index
        if (ws + k == j + k) a = w[j + k];
        if (ws + k + 1 == j + k + 1 && j + k + 1 < 8) b2 = w[j + k + 1];
      }
      tw[k] = bs ? ((a >> bs) | (b2 << (32u - bs))) : a;
    }
    // HostIndex's spell_hash: FNV-1a over the bytes, the first eight of them kept as the head
    unsigned long long h = 1469598103934665603ull, hd = 0;
    uint32_t len = 0;
    bool ended = false;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      const unsigned char c = (unsigned char)(tw[k >> 2] >> (8 * (k & 3)));
      ended = ended || c == 0;
      if (!ended) {
        h ^= c; h *= 1099511628211ull;
        if (k < 8) hd |= (unsigned long long)c << (8 * k);
        len = k + 1;
      }
    }
    if (!ended)
      for (unsigned char c = t[len]; c != 0; c = t[++len]) { h ^= c; h *= 1099511628211ull; }
    h = (h ^ (h >> 29)) & v.mask;
    uint32_t id = 1u;  // VOCAB_UNK
    for (;;) {
      const ulonglong2 e = v.entry[h];
      if (e.x == ~0ull) break;
      if (e.y == hd) {
        bool same = true;
        if (len >= 8) {
          const unsigned char* a = v.arena + (e.x >> 32);
          for (uint32_t k = 8; k <= len && same; k++) same = a[k] == t[k];  // up to and including the terminator
        }
        if (same) { id = (uint32_t)e.x; break; }
      }
      h = (h + 1) & v.mask;
    }
    if (rank < n_tok) ids[rank] = id;
    rank++;
  }
}

}  // namespace fmx
