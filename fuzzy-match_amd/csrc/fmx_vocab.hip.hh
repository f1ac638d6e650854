// fmx_vocab.hip.hh — VocabIndexer::getIndex (src/vocab_indexer.cc:52-60; the per-pattern host prep of match(),
// src/fuzzy_match.cc:468-470) for a whole batch ON THE DEVICE: the batch arrives as token SPELLINGS — NUL-terminated, back to
// back, what a tokenizer writes — and leaves as vocabulary ids in HBM, in token order.  Every thread takes
// kVocabChunk bytes of the text and marks the token starts in them; the rank of a token is the number of starts in front of
// it (counts per tile, a scan of the tile counts, a scan inside the tile); every start is looked up in the device copy of the host's
// spelling table (HostIndex::FastVocab: open addressing, the first eight bytes beside every slot, the spellings in an arena;
// same hash).  Unknown spellings get VOCAB_UNK (1), like getIndex.  The same kernel interns a batch's REAL tokens against the
// index's real-token table (HostIndex::real_table) for indexes whose real tokens differ from the normalised ones.
#pragma once
#include "fmx_kernels.hip.hh"

namespace fmx {

constexpr uint32_t kVocabChunk = 16;  // bytes of text per thread (64 bytes: thirteen look-ups in a row per thread, 0.60 ms per million queries; 16 or 32: 0.41)

struct DevVocab {
  const ulonglong2*    entry;  // {offset into arena << 32 | id (~0 = empty), the spelling's first eight bytes zero padded}: one load per probe
  const unsigned char* arena;  // NUL-terminated spellings
  unsigned long long   mask;
};

// Token starts of the kVocabChunk bytes at b0 (bit i = a spelling starts at byte b0 + i)
__device__ __forceinline__ uint32_t vocab_starts(const unsigned char* __restrict__ text, unsigned long long b0, unsigned long long n_bytes) {
  uint32_t starts = 0;
  if (b0 < n_bytes) {
    unsigned char prev = b0 ? text[b0 - 1] : (unsigned char)0;
#pragma unroll
    for (uint32_t w = 0; w < kVocabChunk / 16; w++) {
      const uint4 x = *reinterpret_cast<const uint4*>(text + b0 + 16 * w);
      const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) {
        const unsigned char c = (unsigned char)(xs[k >> 2] >> (8 * (k & 3)));
        if (c != 0 && prev == 0 && b0 + 16 * w + k < n_bytes) starts |= 1u << (16 * w + k);
        prev = c;
      }
    }
  }
  return starts;
}

// Pass 1: spellings per tile of kScanThreads x kVocabChunk bytes.  (A single-pass look-back scan over the ~5000 tiles of a
// million queries was a chain of ~0.1 us per tile — 0.41 ms; three short kernels without a chain take a third of that.)
__global__ void __launch_bounds__(kScanThreads) k_vocab_count(const unsigned char* __restrict__ text, unsigned long long n_bytes,
                                                             uint32_t* __restrict__ tile_count) {
  __shared__ uint32_t s_w[kScanThreads / kWave];
  const unsigned long long b0 = ((unsigned long long)blockIdx.x * kScanThreads + threadIdx.x) * kVocabChunk;
  uint32_t n = wave_sum((uint32_t)__popc(vocab_starts(text, b0, n_bytes)));
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0;
    for (uint32_t k = 0; k < kScanThreads / kWave; k++) t += s_w[k];
    tile_count[blockIdx.x] = t;
  }
}

// Pass 2: exclusive scan of the tile counts by ONE workgroup (in place); total[0] = the number of spellings.
// ids (optional): the id array the look-up fills — entries [spellings found, n_tok) get `fill` here, so that a text with fewer
// spellings than the offsets count leaves VOCAB_UNK behind and not what the buffer held (normally no entry at all: the array used
// to be memset whole, 100 MB per million-query call, before every look-up).
__global__ void __launch_bounds__(kScanThreads) k_vocab_scan(uint32_t* tile_count, uint32_t n_tiles, unsigned long long* total,
                                                             uint32_t* ids = nullptr, unsigned long long n_tok = 0, uint32_t fill = 0) {
  __shared__ uint32_t s_w[kScanThreads / kWave];
  __shared__ uint32_t s_run;
  if (threadIdx.x == 0) s_run = 0;
  __syncthreads();
  for (uint32_t t0 = 0; t0 < n_tiles; t0 += kScanThreads) {
    const uint32_t t = t0 + threadIdx.x;
    const uint32_t v = t < n_tiles ? tile_count[t] : 0u;
    const uint32_t incl = wave_incl_scan(v);
    if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t wbase = 0;
    for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) wbase += s_w[k];
    const uint32_t run = s_run;
    if (t < n_tiles) tile_count[t] = run + wbase + incl - v;
    __syncthreads();
    if (threadIdx.x == kScanThreads - 1) s_run = run + wbase + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = s_run;
  if (ids)
    for (unsigned long long i = (unsigned long long)s_run + threadIdx.x; i < n_tok; i += kScanThreads) ids[i] = fill;
}

// Pass 3: text: n_bytes bytes followed by at least kVocabChunk + 16 zero bytes, 16-byte aligned.  ids: n_tok words.
__global__ void __launch_bounds__(kScanThreads) k_vocab_lookup(const unsigned char* __restrict__ text, unsigned long long n_bytes,
                                                              unsigned long long n_tok, DevVocab v, uint32_t* __restrict__ ids,
                                                              const uint32_t* __restrict__ tile_base, uint32_t unknown_id) {
  __shared__ uint32_t s_w[kScanThreads / kWave];
  const unsigned long long b0 = ((unsigned long long)blockIdx.x * kScanThreads + threadIdx.x) * kVocabChunk;
  unsigned long long starts = vocab_starts(text, b0, n_bytes);
  const uint32_t cnt = (uint32_t)__popcll(starts);
  const uint32_t incl = wave_incl_scan(cnt);
  if ((threadIdx.x & 63) == 63) s_w[threadIdx.x >> 6] = incl;
  __syncthreads();
  uint32_t wbase = 0;
  for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) wbase += s_w[k];
  unsigned long long rank = (unsigned long long)tile_base[blockIdx.x] + wbase + incl - cnt;
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
      if ((k & 3) == 0 && k != 0 && __ballot(!ended) == 0ull) break;  // every spelling of the wave is through (most are short)
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
    uint32_t id = unknown_id;  // VOCAB_UNK (1) for the vocabulary; "no such real token" for the real-token table
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

// ---------------------------------------------------------------------------------------------
// Pre-tokenised LINES as they sit in a file (fmx_match_batch_lines): tokens separated by blanks, one pattern per
// '\n'-terminated line.  The device cuts them: token starts and line ends per 16 bytes, two scans of the tile counts, then
// every line end writes the pattern's end offset (q_off[line + 1] = token starts in front of it) and the blanks become
// terminators in place — what is left is the packed text k_vocab_lookup takes.  The host never looks at a token.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool text_blank(unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == 0; }

// starts: bit i = a token starts at byte b0 + i; nls: bit i = byte b0 + i is '\n'; conv = the 16 bytes with blanks zeroed
__device__ __forceinline__ void line_masks(const unsigned char* __restrict__ text, unsigned long long b0, unsigned long long n_bytes,
                                           uint32_t& starts, uint32_t& nls, uint4& conv) {
  starts = 0; nls = 0; conv = make_uint4(0u, 0u, 0u, 0u);
  if (b0 >= n_bytes) return;
  bool prev_blank = b0 ? text_blank(text[b0 - 1]) : true;  // (a neighbour may have zeroed it already: still a blank)
  const uint4 x = *reinterpret_cast<const uint4*>(text + b0);
  uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    const unsigned char c = (unsigned char)(xs[k >> 2] >> (8 * (k & 3)));
    const bool in = b0 + k < n_bytes;
    const bool bl = text_blank(c) || !in;
    if (!bl && prev_blank) starts |= 1u << k;
    if (in && c == '\n') nls |= 1u << k;
    if (bl) xs[k >> 2] &= ~(0xffu << (8 * (k & 3)));
    prev_blank = bl;
  }
  conv = make_uint4(xs[0], xs[1], xs[2], xs[3]);
}

static_assert(kVocabChunk == 16, "line_masks handles one 16-byte word per thread");

__global__ void __launch_bounds__(kScanThreads) k_lines_count(const unsigned char* __restrict__ text, unsigned long long n_bytes,
                                                             uint32_t* __restrict__ tile_tok, uint32_t* __restrict__ tile_nl) {
  __shared__ uint32_t s_t[kScanThreads / kWave], s_n[kScanThreads / kWave];
  const unsigned long long b0 = ((unsigned long long)blockIdx.x * kScanThreads + threadIdx.x) * kVocabChunk;
  uint32_t st, nl;
  uint4 conv;
  line_masks(text, b0, n_bytes, st, nl, conv);
  const uint32_t a = wave_sum((uint32_t)__popc(st)), b = wave_sum((uint32_t)__popc(nl));
  if ((threadIdx.x & 63) == 0) { s_t[threadIdx.x >> 6] = a; s_n[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0, n = 0;
    for (uint32_t k = 0; k < kScanThreads / kWave; k++) { t += s_t[k]; n += s_n[k]; }
    tile_tok[blockIdx.x] = t;
    tile_nl[blockIdx.x] = n;
  }
}

// tile_tok / tile_nl: exclusive scans of the tile counts.  q_off: n_lines + 1 entries.
__global__ void __launch_bounds__(kScanThreads) k_lines_emit(unsigned char* __restrict__ text, unsigned long long n_bytes,
                                                            const uint32_t* __restrict__ tile_tok, const uint32_t* __restrict__ tile_nl,
                                                            uint32_t* __restrict__ q_off, unsigned long long n_lines) {
  __shared__ uint32_t s_t[kScanThreads / kWave], s_n[kScanThreads / kWave];
  const unsigned long long b0 = ((unsigned long long)blockIdx.x * kScanThreads + threadIdx.x) * kVocabChunk;
  uint32_t st, nl;
  uint4 conv;
  line_masks(text, b0, n_bytes, st, nl, conv);
  const uint32_t ct = (uint32_t)__popc(st), cn = (uint32_t)__popc(nl);
  const uint32_t it = wave_incl_scan(ct), in = wave_incl_scan(cn);
  if ((threadIdx.x & 63) == 63) { s_t[threadIdx.x >> 6] = it; s_n[threadIdx.x >> 6] = in; }
  __syncthreads();
  uint32_t bt = 0, bn = 0;
  for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) { bt += s_t[k]; bn += s_n[k]; }
  const unsigned long long tok0 = (unsigned long long)tile_tok[blockIdx.x] + bt + it - ct;  // token starts in front of this thread's bytes
  unsigned long long line = (unsigned long long)tile_nl[blockIdx.x] + bn + in - cn;         // line ends in front of them
  if (blockIdx.x == 0 && threadIdx.x == 0) q_off[0] = 0u;
  uint32_t m = nl;
  while (m) {
    const uint32_t i = (uint32_t)__builtin_ctz(m);
    m &= m - 1u;
    if (line < n_lines) q_off[line + 1] = (uint32_t)(tok0 + __popc(st & ((1u << i) - 1u)));
    line++;
  }
  if (b0 < n_bytes) *reinterpret_cast<uint4*>(text + b0) = conv;  // blanks -> terminators (bytes beyond n_bytes are zero already)
}


// ---------------------------------------------------------------------------------------------
// RAW text lines (fmx_match_batch_raw_lines, round 5): what the reference CLI reads (cli/src/FuzzyMatch-cli.cc:219-233) and
// FuzzyMatch::match(const std::string&, ...) tokenises (src/fuzzy_match.cc:393-412 -> _tokenize_and_normalize, :108-193).  For
// lines made of ASCII letters, digits, punctuation and blanks — the alphabet the reference's fixtures pin the tokenizer on — the
// aggressive tokenizer is a byte-class machine (fmx_text.cc: tokenize_aggressive): a run of letters is a token, a run of digits
// is a token, every other printable character is a token of its own, white space separates; with the `cas` penalty token a word
// is lower-cased and its real token is the case feature (L / C / U / M), with `nbr` a number's normalised token is the
// placeholder and its real token the digits (src/fuzzy_match.cc:131-135, 160-170).  The device runs that machine on the bytes as
// they sit in the file: token starts and line ends per 16 bytes, two scans of the tile counts, then every token start computes
// its run, hashes it (lower-cased where the index says so) and looks its vocabulary id — and its real-token id — up; every line
// end writes its pattern's end offset.  No spelling is ever written.  A line with any other byte (UTF-8, control characters) is
// FLAGGED: the host tokenises those lines (fmx_text.cc, the full front-end) and answers them through a text batch; whatever the
// device made of them is overwritten.
// ---------------------------------------------------------------------------------------------
enum : uint32_t { kRawSep = 0, kRawLetter = 1, kRawDigit = 2, kRawOther = 3, kRawBad = 4 };
__device__ __forceinline__ uint32_t raw_class(unsigned char c) {
  const unsigned char l = c | 0x20;
  if (l >= 'a' && l <= 'z') return kRawLetter;
  if (c >= '0' && c <= '9') return kRawDigit;
  if (c == ' ' || (c >= 9 && c <= 13) || c == 0) return kRawSep;   // onmt::unicode::is_separator on ASCII (0: the padding behind the text)
  if (c > 32 && c < 128) return kRawOther;
  return kRawBad;
}

struct RawTok {
  uint32_t cas, nbr;        // penalty tokens of the index that change a word / a number (fmx_text.hh: kPtCas, kPtNbr)
  uint32_t num_id;          // vocabulary id of the number placeholder (VOCAB_UNK when the index never saw a number)
  uint32_t feat_real[4];    // real-token ids of the case features L, C, U, M (0x7ffffffe: the index holds no such real token)
};

// starts: bit i = a token starts at byte b0 + i; nls: line ends; bad: bytes outside the alphabet
__device__ __forceinline__ void raw_masks(const unsigned char* __restrict__ text, unsigned long long b0, unsigned long long n_bytes,
                                          uint32_t& starts, uint32_t& nls, uint32_t& bad) {
  starts = 0; nls = 0; bad = 0;
  if (b0 >= n_bytes) return;
  uint32_t prev = b0 ? raw_class(text[b0 - 1]) : (uint32_t)kRawSep;
  if (prev == kRawBad) prev = kRawSep;
  const uint4 x = *reinterpret_cast<const uint4*>(text + b0);
  const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    const unsigned char c = (unsigned char)(xs[k >> 2] >> (8 * (k & 3)));
    const bool in = b0 + k < n_bytes;
    uint32_t cl = in ? raw_class(c) : (uint32_t)kRawSep;
    // (a NUL byte INSIDE the text is not a separator for the host front-end, which drops code points below 32 without ending the
    // token: such a line is the host's; only the padding behind the text reads as separators)
    if (cl == kRawBad || (in && c == 0)) { bad |= 1u << k; cl = kRawSep; }
    if (cl != kRawSep && (cl == kRawOther || cl != prev)) starts |= 1u << k;
    if (in && c == '\n') nls |= 1u << k;
    prev = cl;
  }
}

__global__ void __launch_bounds__(kScanThreads) k_raw_count(const unsigned char* __restrict__ text, unsigned long long n_bytes,
                                                           uint32_t* __restrict__ tile_tok, uint32_t* __restrict__ tile_nl) {
  __shared__ uint32_t s_t[kScanThreads / kWave], s_n[kScanThreads / kWave];
  const unsigned long long b0 = ((unsigned long long)blockIdx.x * kScanThreads + threadIdx.x) * kVocabChunk;
  uint32_t st, nl, bad;
  raw_masks(text, b0, n_bytes, st, nl, bad);
  const uint32_t a = wave_sum((uint32_t)__popc(st)), b = wave_sum((uint32_t)__popc(nl));
  if ((threadIdx.x & 63) == 0) { s_t[threadIdx.x >> 6] = a; s_n[threadIdx.x >> 6] = b; }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t t = 0, n = 0;
    for (uint32_t k = 0; k < kScanThreads / kWave; k++) { t += s_t[k]; n += s_n[k]; }
    tile_tok[blockIdx.x] = t;
    tile_nl[blockIdx.x] = n;
  }
}

// the id of the `len` bytes at t in table v (HostIndex::FastVocab: FNV-1a, the first eight bytes beside every slot); lower: upper-case
// ASCII letters count as their lower-case counterparts (onmt::CaseModifier on ASCII)
__device__ __forceinline__ uint32_t raw_find(const DevVocab& v, const unsigned char* __restrict__ t, uint32_t len, bool lower, uint32_t unknown_id) {
  unsigned long long h = 1469598103934665603ull, hd = 0;
  for (uint32_t k = 0; k < len; k++) {
    unsigned char c = t[k];
    if (lower && c >= 'A' && c <= 'Z') c |= 0x20;
    h ^= c; h *= 1099511628211ull;
    if (k < 8) hd |= (unsigned long long)c << (8 * k);
  }
  h = (h ^ (h >> 29)) & v.mask;
  for (;;) {
    const ulonglong2 e = v.entry[h];
    if (e.x == ~0ull) return unknown_id;
    if (e.y == hd) {
      bool same = true;
      if (len >= 8) {
        const unsigned char* a = v.arena + (e.x >> 32);
        for (uint32_t k = 8; k <= len && same; k++) {
          unsigned char c = k < len ? t[k] : (unsigned char)0;
          if (lower && c >= 'A' && c <= 'Z') c |= 0x20;
          same = a[k] == c;
        }
      }
      if (same) return (uint32_t)e.x;
    }
    h = (h + 1) & v.mask;
  }
}

// tile_tok / tile_nl: exclusive scans of the tile counts.  q_off: n_lines + 1 entries; ids / real: one per token (real may be
// null: the index's real tokens are its normalised tokens); line_bad: one byte per line, zeroed by the caller; n_bad: lines flagged.
__global__ void __launch_bounds__(kScanThreads) k_raw_emit(const unsigned char* __restrict__ text, unsigned long long n_bytes,
                                                          const uint32_t* __restrict__ tile_tok, const uint32_t* __restrict__ tile_nl,
                                                          uint32_t* __restrict__ q_off, unsigned long long n_lines, unsigned long long n_tok,
                                                          DevVocab vocab, DevVocab real_vocab, RawTok rt, uint32_t* __restrict__ ids,
                                                          uint32_t* __restrict__ real, unsigned char* __restrict__ line_bad,
                                                          unsigned long long* __restrict__ n_bad) {
  __shared__ uint32_t s_t[kScanThreads / kWave], s_n[kScanThreads / kWave];
  const unsigned long long b0 = ((unsigned long long)blockIdx.x * kScanThreads + threadIdx.x) * kVocabChunk;
  uint32_t st, nl, bad;
  raw_masks(text, b0, n_bytes, st, nl, bad);
  const uint32_t ct = (uint32_t)__popc(st), cn = (uint32_t)__popc(nl);
  const uint32_t it = wave_incl_scan(ct), in = wave_incl_scan(cn);
  if ((threadIdx.x & 63) == 63) { s_t[threadIdx.x >> 6] = it; s_n[threadIdx.x >> 6] = in; }
  __syncthreads();
  uint32_t bt = 0, bn = 0;
  for (uint32_t k = 0; k < (threadIdx.x >> 6); k++) { bt += s_t[k]; bn += s_n[k]; }
  const unsigned long long tok0 = (unsigned long long)tile_tok[blockIdx.x] + bt + it - ct;  // token starts in front of this thread's bytes
  const unsigned long long line0 = (unsigned long long)tile_nl[blockIdx.x] + bn + in - cn;   // line ends in front of them
  if (blockIdx.x == 0 && threadIdx.x == 0) q_off[0] = 0u;
  {  // every line end writes its pattern's end offset
    uint32_t m = nl;
    unsigned long long line = line0;
    while (m) {
      const uint32_t i = (uint32_t)__builtin_ctz(m);
      m &= m - 1u;
      if (line < n_lines) q_off[line + 1] = (uint32_t)(tok0 + __popc(st & ((1u << i) - 1u)));
      line++;
    }
  }
  {  // a byte outside the alphabet flags its line (the line of byte i: line ends in front of it)
    uint32_t m = bad;
    while (m) {
      const uint32_t i = (uint32_t)__builtin_ctz(m);
      m &= m - 1u;
      const unsigned long long line = line0 + __popc(nl & ((1u << i) - 1u));
      if (line < n_lines && line_bad[line] == 0) {
        // (several threads may flag one line: the byte is written by all of them, the count by the first to swap its word)
        unsigned int* wptr = reinterpret_cast<unsigned int*>(line_bad + (line & ~3ull));
        const unsigned int bit = 1u << (8 * (line & 3ull));
        const unsigned int old = atomicOr(wptr, bit);
        if ((old & bit) == 0) atomicAdd(n_bad, 1ull);
      }
    }
  }
  unsigned long long rank = tok0;
  uint32_t m = st;
  while (m) {
    const uint32_t i = (uint32_t)__builtin_ctz(m);
    m &= m - 1u;
    const unsigned char* t = text + b0 + i;
    const uint32_t cl = raw_class(t[0]);
    uint32_t len = 1;
    if (cl != kRawOther) while (raw_class(t[len]) == cl) len++;   // (the text is followed by zero bytes: a run ends there at the latest)
    uint32_t id, rid = 0x7ffffffeu;
    const bool word = cl == kRawLetter && rt.cas != 0;
    if (cl == kRawDigit && rt.nbr != 0) id = rt.num_id;
    else id = raw_find(vocab, t, len, word, 1u);   // VOCAB_UNK for spellings the index never saw (VocabIndexer::getIndex)
    if (real != nullptr) {
      if (word) {  // onmt::CaseModifier::extract_case_type over the run
        uint32_t n_up = 0;
        for (uint32_t k = 0; k < len; k++) n_up += (t[k] >= 'A' && t[k] <= 'Z') ? 1u : 0u;
        const bool first_up = t[0] >= 'A' && t[0] <= 'Z';
        const uint32_t f = n_up == 0 ? 0u : ((first_up && n_up == 1) ? 1u : ((n_up == len && len >= 2) ? 2u : 3u));
        rid = rt.feat_real[f];
      } else {
        rid = raw_find(real_vocab, t, len, false, 0x7ffffffeu);
      }
    }
    if (rank < n_tok) { ids[rank] = id; if (real != nullptr) real[rank] = rid; }
    rank++;
  }
}

}  // namespace fmx
