// fmx_host.hh — host-side index model: what SuffixArrayIndex + VocabIndexer + SuffixArray hold in
// the reference (include/fuzzy/suffix_array_index.hh:50-54), plus the derived SoA arrays that
// are uploaded to HBM.  Pure C++17, no HIP.
#pragma once
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "fmx_core.hh"

namespace fmx {

unsigned usable_cpus();  // min(affinity mask, cgroup quota): what thread pools are sized by

// fuzzy::Sentence (include/fuzzy/sentence.hh:24-48)
struct RealTok {
  std::string tokstring;                               // real tokens joined by '\t'
  std::vector<std::pair<uint64_t, std::string>> itoks;  // "index of next real token" -> penalty string
  std::vector<std::string> split() const;               // Sentence::operator Tokens (src/sentence.cc:7-22)
};

class HostIndex {
 public:
  // FuzzyMatch::_pt, SuffixArrayIndex::_max_tokens_in_pattern
  int      pt = 0;
  uint64_t max_tokens = 300;

  // VocabIndexer (include/fuzzy/vocab_indexer.hh:38-41).  For id-only corpora (bulk builder)
  // forms/form2index stay empty and only vocab_size/sfreq are meaningful.
  std::vector<std::string>                  forms;
  std::vector<uint32_t>                     sfreq;
  std::unordered_map<std::string, uint32_t> form2index;
  uint32_t                                  vocab_size = 0;

  // SuffixArray (include/fuzzy/suffix_array.hh:50-61)
  bool                  sorted = false;
  std::vector<uint32_t> suf_sid;
  std::vector<uint16_t> suf_pos;
  std::vector<uint32_t> buf;       // _sentence_buffer: [len, tok x len, 0] per sentence
  std::vector<uint32_t> sent_pos;  // _sentence_pos
  std::vector<uint32_t> qva;       // _quickVocabAccess

  // What a loaded .fmi does not determine from the arrays alone: archive / class versions and the iteration order and
  // bucket counts of its unordered_maps (implementation-defined in the writer's C++ library).  Kept so that
  // load -> save reproduces a reference-written file byte for byte; a freshly built index writes the current versions.
  struct FmiMeta {
    bool     loaded = false;
    uint16_t lvt = 14;            // Boost archive library version
    uint32_t fm_version = 1;      // FuzzyMatch class version
    uint32_t sai_version = 1;     // SuffixArrayIndex class version (1 = _max_tokens_in_pattern trailer)
    std::vector<uint32_t> map_order;     // form2index values in file order
    uint64_t map_buckets = 0;
    std::vector<uint64_t> itok_buckets;  // per sentence
  } fmi;

  // SuffixArrayIndex::_ids / _real_tokens — empty for id-only corpora
  std::vector<std::string> ids;
  std::vector<RealTok>     real;

  // ---- derived on demand by derive_heavy(): the product builds these on the device (fmx_derive.hip.hh) ----
  bool                  heavy_ready = false;
  std::vector<SufRec>   suf_rec;   // one 8-byte record per suffix, suffix-array order
  std::vector<uint64_t> suf_sig;   // sentence length (8 bits, clamped) << 56 | 56-bit word signature of the suffix's sentence
  std::vector<SufKey>   suf_key;   // the eight tokens after each suffix's first token (0-padded)
  std::vector<SufRecB>  suf_b;     // phase-B record: SufRec + previous token + sentence id
  std::vector<GramEntry> gram;     // trigram -> suffix range hash table (power-of-two size, empty if tiny index)
  std::vector<G5Bucket>  gram5;    // every distinct 5-gram -> suffix range + the tokens its group shares (128-byte buckets, fmx_core.hh)
  std::vector<GramEntry> gram7;    // (5-gram range start, w5, w6) -> 7-gram range, for 5-grams with > kDeepGramMin suffixes
  void build_gram_table(int nthreads);
  // ---- derived by derive() ----
  std::vector<float>    idf;       // logf(N / sfreq[w]) per word id (src/fuzzy_match.cc:384)
  float                 idf_max = 0.01f;
  bool                  real_differs = false;  // some sentence has real tokens != its normalised tokens, or itoks
  std::vector<uint32_t> real_ids;  // per token slot of buf: interned real-token id | LUMC flag (bit 31)
  std::vector<uint32_t> itok_ids;  // per slot of buf (slot sent_pos+i <-> itok index i in [0, len])
  std::unordered_map<std::string, uint32_t> real_intern;
  std::vector<std::string>                  itok_strings;  // id 0 = ""
  std::unordered_map<std::string, uint32_t> itok_intern;
  // byte-Levenshtein distance of every pair of penalty strings.  itok_tab is the working copy with a row stride of itok_cap
  // (a power of two: a new string costs one row and one column, the table is re-laid out only when the stride doubles);
  // itok_dist is the compact n x n copy the kernels index (n = itok_strings.size()), rebuilt by sync_itok_dist() when stale.
  std::vector<uint16_t>                     itok_tab;
  uint32_t                                  itok_cap = 0;
  mutable std::vector<uint16_t>             itok_dist;
  mutable uint64_t                          itok_dist_version = ~0ull;
  // itok_sync_mu guards itok_strings / itok_intern / itok_tab / itok_cap / itok_dist: intern_itok() takes it for its whole
  // body, so a match() that interns a new string cannot re-lay out the table under another thread's snapshot
  mutable std::mutex                        itok_sync_mu;
  void sync_itok_dist() const;
  // a consistent copy of the compact table with the version and string count it belongs to (what a device replica uploads)
  void snapshot_itok_dist(std::vector<uint16_t>& out, uint32_t& n, uint64_t& version) const;
  std::atomic<uint64_t>                     itok_version{0};

  HostIndex();

 private:
  struct FastVocab {
    std::vector<char>     arena;   // NUL-terminated spellings
    std::vector<uint64_t> slot;    // offset into arena << 32 | id, ~0 = empty
    std::vector<uint64_t> head;    // beside every slot: the spelling's first eight bytes (zero-padded) — spellings of up to eight
                                   // bytes are decided without touching the arena (one cache miss fewer per look-up)
    uint64_t              mask = 0;
    std::atomic<uint64_t> built_for{0};   // vocabulary generation it was built from (0 = not built)
  };
  // bumped by everything that changes a spelling or an id (add_word, set_forms, load_fmi): the spelling table here and its
  // device copy are keyed on it, not on forms.size() (set_forms can change spellings without changing the size)
  std::atomic<uint64_t> vocab_gen_{1};
  mutable FastVocab  fast_;
  mutable FastVocab  real_fast_;     // the same kind of table over the interned real tokens (real_table())
  uint64_t           real_gen_ = 1;  // bumped by derive() (real_intern is rebuilt there)
  mutable std::mutex fast_mu_;
  const FastVocab& fast_vocab() const;

 public:

  // VocabIndexer::addWord / getIndex (src/vocab_indexer.cc:37-60)
  uint32_t add_word(const std::string& w);
  uint32_t get_index(const std::string& w) const;
  // getIndex for a stream of tokens (the per-query host prep of match(), src/fuzzy_match.cc:468-470) without a std::string
  // per token: an open-addressing table over the packed spellings (built on first use, rebuilt when words were added),
  // spread over `nthreads` host threads
  void get_indices(const char* const* toks, uint64_t n, uint32_t* ids, int nthreads) const;
  // the spelling table behind get_indices, for its device copy (fmx_vocab.hip.hh); built_for = the vocabulary generation it was built from
  struct VocabTable { const uint64_t* slot; const uint64_t* head; const char* arena; uint64_t n_slots, arena_bytes, built_for; };
  VocabTable vocab_table() const;
  VocabTable real_table() const;   // interned real-token spellings -> ids (indexes whose real tokens differ from the normalised ones)
  // attach spellings to an id-only index (ids 2 .. n - 1; bulk-built corpora): vocabulary lookups then work on it
  void set_forms(const char* const* spellings, uint64_t n);

  // SuffixArrayIndex::add_tm (src/suffix_array_index.cc:10-30): returns the new sentence id or -1
  // when skipped (empty real tokens, or more than max_tokens normalised tokens).
  int64_t add_tm(const std::string& id, const RealTok& real_tokens, const std::vector<std::string>& norm);
  // Adopt the reference's SuffixArray members as they are (include/fuzzy/suffix_array.hh:50-61): an already sorted suffix
  // array, the sentence buffer, sentence positions and (optionally) _quickVocabAccess / VocabIndexer::sfreq.  Validates
  // the structure (not the sort order of every pair: adjacent suffixes of a sample are compared).  Id-only index.
  void adopt_arrays(const uint32_t* sid, const uint16_t* pos, uint64_t n_suf, const uint32_t* buffer, uint64_t n_buf,
                    const uint32_t* spos, uint64_t n_sent, const uint32_t* qva_in, uint64_t n_qva, const uint32_t* sfreq_in,
                    uint32_t vocab_size_);
  // id-level bulk add (synthetic corpora): CSR token stream with ids >= 2 already assigned.
  void add_ids_bulk(const uint32_t* off, const uint32_t* toks, uint64_t n_sent, uint32_t vocab_size_);

  // SuffixArray::sort (src/suffix_array.cc:58-102) on the CPU; result identical to the reference's.
  void sort(int nthreads);
  // adopt a suffix order computed elsewhere (the GPU builder) — keys = sid << 16 | pos, sorted.
  void adopt_sorted(const uint64_t* keys, uint64_t n);
  // the same when the builder hands over the parallel arrays and _quickVocabAccess it computed itself (sizes must match)
  void adopt_sorted_arrays(std::vector<uint32_t>&& sid, std::vector<uint16_t>&& pos, std::vector<uint32_t>&& qva_in);
  // unsorted (sid<<16|pos) keys + first-token bucket offsets, for the GPU builder
  void bucket_keys(std::vector<uint64_t>& keys, std::vector<uint64_t>& bucket_off) const;

  void derive();        // idf table, interned real tokens / penalty tokens (what the host side of the product needs)
  void derive_heavy();  // + the per-suffix accelerator arrays and n-gram tables on the host (tests, CPU emulation)
  uint32_t num_sentences() const { return (uint32_t)sent_pos.size(); }
  uint64_t num_suffixes() const { return suf_sid.size(); }

  // penalty-string interning for query batches; grows the distance table (thread safe: serialised on itok_sync_mu)
  uint32_t intern_itok(const std::string& s);
  uint32_t lookup_real(const std::string& s) const;  // 0x7ffffffe when absent

  // .fmi (src/fuzzy_matcher_binarization.cc:16-51 + the per-class save/load templates)
  void load_fmi(const std::string& path);
  void save_fmi(const std::string& path) const;
};

// _edit_distance_char (include/fuzzy/edit_distance.hxx:7-35)
int edit_distance_char(const char* s1, int n1, const char* s2, int n2);

}  // namespace fmx
