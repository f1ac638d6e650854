// fuzzy/fuzzy_match.hh — C++ facade that re-creates the reference's public class on top of the C ABI
// (include/fmx.h).  Same names, argument meaning, defaults and error behaviour as
//   include/fuzzy/fuzzy_match.hh:17-147           class fuzzy::FuzzyMatch, Match, penalty_token, ContrastReduce
//   include/fuzzy/costs.hh:7-29                   EditCosts
//   include/fuzzy/sentence.hh:22-48               Tokens, Sentence
//   include/fuzzy/fuzzy_matcher_binarization.hh   import_/export_binarized_fuzzy_matcher
// of SYSTRAN/fuzzy-match, so a caller of libFuzzyMatch.so re-links against libfmx.so and keeps its
// source.  Header only; every call goes through the extern "C" entry points.
//
// add_tm(..., sort = true) does what the reference does — further add_tm calls stay legal: the suffix sort, the
// derived arrays and the upload to HBM happen lazily at the first match()/export after a change, and an add_tm
// after that thaws the index again (fmx_index_reopen).  match() is const and re-entrant like the reference's:
// every call goes through fmx_match_one, which coalesces concurrent callers into GPU micro-batches.
#pragma once
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <iostream>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/fmx.h"

namespace fuzzy {

typedef std::vector<std::string> Tokens;

constexpr size_t DEFAULT_MAX_TOKENS_IN_PATTERN = 300;

struct EditCosts {
  const float insert_cost;
  const float delete_cost;
  const float replace_cost;
  EditCosts() : insert_cost(1), delete_cost(1), replace_cost(1) {}
  EditCosts(float i, float d, float r) : insert_cost(i), delete_cost(d), replace_cost(r) {}
  bool is_null() const { return insert_cost == 0. && delete_cost == 0. && replace_cost == 0.; }
};

// include/fuzzy/sentence.hh:24-48 — real tokens joined by '\t' plus sparse penalty tokens
class Sentence {
 public:
  Sentence() {}
  Sentence(const Tokens& s) {
    for (size_t i = 0; i < s.size(); i++) { if (i) _tokstring += '\t'; _tokstring += s[i]; }
  }
  operator Tokens() const {
    Tokens parts;
    std::string accu;
    for (const char c : _tokstring) {
      if (c == '\t') { parts.emplace_back(std::move(accu)); accu.clear(); } else accu += c;
    }
    if (!accu.empty()) parts.emplace_back(std::move(accu));
    return parts;
  }
  void push_back(const std::string& s) { if (!_tokstring.empty()) _tokstring += "\t"; _tokstring += s; }
  void reserve(size_t n) { _tokstring.reserve(n); }
  bool empty() const { return _tokstring.empty(); }
  void set_itok(size_t idx, const std::string& itok) { _itoks[idx] += itok; }
  const std::string& tokstring() const { return _tokstring; }
  const std::unordered_map<size_t, std::string>& itoks() const { return _itoks; }

 private:
  std::string _tokstring;
  std::unordered_map<size_t, std::string> _itoks;
};

enum class ContrastReduce { MEAN, MAX };

class FuzzyMatch {
 public:
  enum penalty_token {
    pt_none = 0, pt_tag = 1 << 0, pt_pct = 1 << 1, pt_sep = 1 << 2, pt_jnr = 1 << 3, pt_nbr = 1 << 4, pt_cas = 1 << 5,
  };

  struct Match {
    Match(const unsigned* seq, int length) : length(length), s(seq) {}
    Match() {}
    float score;
    float penalty;
    int max_subseq;
    unsigned s_id;
    std::string id;
    int length;
    const unsigned* s;  // borrows the index's sentence buffer; valid while the FuzzyMatch lives
  };

  FuzzyMatch(int pt = pt_none, size_t max_tokens_in_pattern = DEFAULT_MAX_TOKENS_IN_PATTERN, int device = 0)
      : _pt(pt), _max_tokens(max_tokens_in_pattern), _device(device),
        _ix(fmx_index_create(pt, max_tokens_in_pattern)) {
    if (!_ix) throw std::bad_alloc();
  }
  ~FuzzyMatch() { fmx_index_free(_ix); }
  FuzzyMatch(const FuzzyMatch&) = delete;
  FuzzyMatch& operator=(const FuzzyMatch&) = delete;

  bool add_tm(const std::string& id, const Tokens& norm, bool sort = true) {
    return add_tm(id, Sentence(norm), norm, sort);
  }
  // SuffixArrayIndex::add_tm (src/suffix_array_index.cc:10-30): legal at any time, also after sort() / match()
  bool add_tm(const std::string& id, const Sentence& source, const Tokens& norm, bool sort = true) {
    std::lock_guard<std::mutex> lk(_mu);
    if (_ready) { check(fmx_index_reopen(_ix)); _ready = false; }
    std::vector<const char*> toks;
    for (const auto& t : norm) toks.push_back(t.c_str());
    std::vector<uint64_t> idx;
    std::vector<const char*> strs;
    for (const auto& kv : source.itoks()) { idx.push_back(kv.first); strs.push_back(kv.second.c_str()); }
    const int64_t r = fmx_index_add_tm(_ix, id.c_str(), toks.data(), (uint32_t)toks.size(), source.tokstring().c_str(),
                                       (uint32_t)idx.size(), idx.data(), strs.data());
    if (r < 0 && r != FMX_TM_SKIPPED) throw std::runtime_error(fmx_last_error());
    (void)sort;  // the sort is lazy: see ensure_ready()
    return true;
  }
  // add_tm(id, sentence) — src/fuzzy_match.cc:213-224: tokenised by the library's restricted text front-end
  // (include/fmx.h, fmx_text_*); an empty segment is not indexed (warning on stderr, false)
  bool add_tm(const std::string& id, const std::string& sentence, bool sort = true) {
    std::lock_guard<std::mutex> lk(_mu);
    if (_ready) { check(fmx_index_reopen(_ix)); _ready = false; }
    const int64_t r = fmx_index_add_text(_ix, id.c_str(), sentence.data(), sentence.size());
    if (r < 0 && r != FMX_TM_SKIPPED) throw std::runtime_error(fmx_last_error());
    (void)sort;
    if (r == FMX_TM_SKIPPED && text_is_empty(sentence)) {
      std::cerr << "WARNING: cannot index empty segment: " << sentence << " (" << id << ")" << std::endl;
      return false;
    }
    return true;
  }

  // SuffixArray::sort (src/suffix_array.cc:58-102): here a promise — the work happens in ensure_ready()
  void sort() {}
  // (not in the reference) GPUs to replicate the index on; call before the first match().  Default: the constructor's
  // device, or the FMX_DEVICES environment variable ("0,1,2,3").
  void set_devices(const std::vector<int>& devices) { _devices = devices; }
  // match(const Tokens&, ...) — include/fuzzy/fuzzy_match.hh:59-69
  bool match(const Tokens& pattern, float fuzzy, unsigned number_of_matches, std::vector<Match>& matches,
             int min_subseq_length = 2, float min_subseq_ratio = 0, float vocab_idf_penalty = 0,
             const EditCosts& edit_costs = EditCosts(), float contrastive_factor = 0,
             ContrastReduce reduce = ContrastReduce::MEAN, int contrast_buffer = -1) const {
    return match(Sentence(pattern), pattern, fuzzy, number_of_matches, false, matches, min_subseq_length,
                 min_subseq_ratio, vocab_idf_penalty, edit_costs, contrastive_factor, reduce, contrast_buffer);
  }

  // match(const Sentence& real, const Tokens& pattern, ...) — include/fuzzy/fuzzy_match.hh:70-82,
  // src/fuzzy_match.cc:435-681.  Appends to `matches` (never clears it); returns matches.size() > 0.
  bool match(const Sentence& real, const Tokens& pattern, float fuzzy, unsigned number_of_matches, bool no_perfect,
             std::vector<Match>& matches, int min_subseq_length = 3, float min_subseq_ratio = 0.3,
             float vocab_idf_penalty = 0, const EditCosts& edit_costs = EditCosts(), float contrastive_factor = 0,
             ContrastReduce reduce = ContrastReduce::MEAN, int contrast_buffer = -1) const {
    ensure_ready();
    const uint32_t p = (uint32_t)pattern.size();
    if (p == 0 || p > _max_tokens) return false;  // src/fuzzy_match.cc:455-461
    // `matches` is appended to, never cleared, and entries already in it count towards number_of_matches
    // (src/fuzzy_match.cc:633,670)
    const size_t existing = matches.size();
    if (number_of_matches != 0 && existing >= number_of_matches) return true;
    std::vector<const char*> toks;
    for (const auto& t : pattern) toks.push_back(t.c_str());
    std::vector<uint32_t> wids(p), q_real, q_itok;
    check(fmx_vocab_lookup(_ix, toks.data(), p, wids.data()));
    real_ids(real, p, q_real, q_itok);
    fmx_params prm;
    prm.fuzzy = fuzzy;
    prm.nmatch = number_of_matches ? number_of_matches - (unsigned)existing : 0;
    prm.no_perfect = no_perfect;
    prm.min_subseq_length = min_subseq_length; prm.min_subseq_ratio = min_subseq_ratio;
    prm.idf_penalty = vocab_idf_penalty;
    prm.insert_cost = edit_costs.insert_cost; prm.delete_cost = edit_costs.delete_cost;
    prm.replace_cost = edit_costs.replace_cost;
    prm.contrast = contrastive_factor; prm.reduce = reduce == ContrastReduce::MAX ? 1 : 0;
    // "-1 => number_of_matches" is taken on the caller's number (src/fuzzy_match.cc:451-452)
    prm.contrast_buffer = contrast_buffer == -1 ? (int)number_of_matches : contrast_buffer;
    uint32_t cap = prm.nmatch ? prm.nmatch : 64;
    std::vector<fmx_match> out;
    uint32_t count = 0, total = 0;
    // entries already in `matches` take part in the contrastive rerank: every round penalises the candidates against
    // ALL of `matches`, the prior entries first (src/fuzzy_match.cc:622-650); their token sequences go along
    // (and their s_ids: the reference's memo of pair costs is keyed on ids, src/fuzzy_match.cc:637)
    std::vector<uint32_t> prior_off, prior_tok, prior_sid;
    const bool with_prior = existing != 0 && contrastive_factor > 0;
    if (with_prior) {
      prior_off.push_back(0);
      for (const Match& m : matches) {
        for (int k = 0; k < m.length; k++) prior_tok.push_back(m.s[k]);
        prior_off.push_back((uint32_t)prior_tok.size());
        prior_sid.push_back(m.s_id);
      }
      if (prior_tok.empty()) prior_tok.push_back(0);
    }
    for (;;) {  // number_of_matches == 0 means "all": grow until everything fits
      out.resize(cap);
      if (with_prior)
        check(fmx_match_one_prior_ids(_ix, &prm, wids.data(), p, q_real.empty() ? nullptr : q_real.data(),
                                      q_itok.empty() ? nullptr : q_itok.data(), prior_off.data(), prior_tok.data(), prior_sid.data(),
                                      (uint32_t)existing, out.data(), cap, &count, &total));
      else
        check(fmx_match_one(_ix, &prm, wids.data(), p, q_real.empty() ? nullptr : q_real.data(),
                            q_itok.empty() ? nullptr : q_itok.data(), out.data(), cap, &count, &total));
      if (total <= cap) break;
      cap = total;
    }
    for (uint32_t k = 0; k < count; k++) {
      Match m(_buffer + _sent_pos[out[k].s_id] + 1, (int)_buffer[_sent_pos[out[k].s_id]]);
      m.score = out[k].score; m.penalty = out[k].penalty; m.max_subseq = out[k].max_subseq;
      m.s_id = out[k].s_id; m.id = fmx_index_sentence_id(_ix, out[k].s_id);
      matches.push_back(m);
    }
    return matches.size() > 0;
  }

  // match(const std::string&, ...) — include/fuzzy/fuzzy_match.hh:85-97, src/fuzzy_match.cc:393-412: tokenise +
  // normalise (restricted text front-end of the library), then the Sentence/Tokens overload
  bool match(const std::string& sentence, float fuzzy, unsigned number_of_matches, bool no_perfect,
             std::vector<Match>& matches, int min_subseq_length = 3, float min_subseq_ratio = 0.3,
             float vocab_idf_penalty = 0, const EditCosts& edit_costs = EditCosts(), float contrastive_factor = 0,
             ContrastReduce reduce = ContrastReduce::MEAN, int contrast_buffer = -1) const {
    TextHandle t(_pt, sentence);
    Sentence real;
    Tokens norm;
    t.fill(real, norm);
    return match(real, norm, fuzzy, number_of_matches, no_perfect, matches, min_subseq_length, min_subseq_ratio,
                 vocab_idf_penalty, edit_costs, contrastive_factor, reduce, contrast_buffer);
  }
  // subsequence() — include/fuzzy/fuzzy_match.hh:99-105, src/fuzzy_match.cc:250-365: the sub-n-gram of the sentence
  // with the largest weight that occurs in the memory, and the closest of up to number_of_matches sentences holding
  // it; Match::id = the sentence's id + '\t' + the detokenised sub-n-gram
  bool subsequence(const std::string& sentence, unsigned number_of_matches, bool no_perfect, std::vector<Match>& matches,
                   int min_subseq_length = 3, float min_subseq_ratio = 0.3, bool idf_weighting = false) const {
    TextHandle t(_pt, sentence);
    Sentence real;
    Tokens norm;
    t.fill(real, norm);
    uint32_t pos = 0;
    Match m;
    if (!subsequence_tokens(real, norm, number_of_matches, no_perfect, min_subseq_length, min_subseq_ratio, idf_weighting, m, pos))
      return false;
    m.id += "\t";
    m.id += fmx_text_detokenize(t.h, pos, pos + (uint32_t)m.max_subseq);
    matches.push_back(m);
    return true;
  }
  // the same on a tokenised pattern (no text to detokenise: the sub-n-gram's tokens are joined by blanks)
  bool subsequence(const Tokens& pattern, unsigned number_of_matches, bool no_perfect, std::vector<Match>& matches,
                   int min_subseq_length = 3, float min_subseq_ratio = 0.3, bool idf_weighting = false) const {
    uint32_t pos = 0;
    Match m;
    if (!subsequence_tokens(Sentence(pattern), pattern, number_of_matches, no_perfect, min_subseq_length, min_subseq_ratio,
                            idf_weighting, m, pos))
      return false;
    m.id += "\t";
    for (int k = 0; k < m.max_subseq; k++) { if (k) m.id += ' '; m.id += pattern[pos + k]; }
    matches.push_back(m);
    return true;
  }

  // _tokenize_and_normalize — include/fuzzy/fuzzy_match.hh:106-118, src/fuzzy_match.cc:93-193 (both public in the
  // reference): NFC normalisation, the aggressive tokenizer configured from the penalty-token mask, then the split into
  // normalised tokens (`pattern`), real tokens and penalty strings (`real`)
  void _tokenize_and_normalize(const std::string& sentence, Sentence& real, Tokens& pattern) const {
    std::vector<unsigned> map_tokens;
    std::vector<std::string> tokens;
    std::vector<std::vector<std::string>> features;
    _tokenize_and_normalize(sentence, real, pattern, map_tokens, tokens, features);
  }
  // ... also the tokenizer's own tokens and features and, per normalised token, the raw-token index behind it
  // (map_tokens[0] = 0).  `features` holds one stream (the case feature) under pt_cas and none otherwise, like the
  // tokenizer's output for the flags the reference sets (src/fuzzy_match.cc:70-80).
  void _tokenize_and_normalize(const std::string& sentence, Sentence& real, Tokens& pattern, std::vector<unsigned>& map_tokens,
                               std::vector<std::string>& tokens, std::vector<std::vector<std::string>>& features) const {
    TextHandle t(_pt, sentence);
    t.fill(real, pattern);
    const uint32_t n_raw = fmx_text_num_raw(t.h), n = fmx_text_num_tokens(t.h);
    for (uint32_t i = 0; i < n_raw; i++) tokens.emplace_back(fmx_text_raw(t.h, i));
    if (_pt & pt_cas) {
      features.emplace_back();
      for (uint32_t i = 0; i < n_raw; i++) features.back().emplace_back(fmx_text_feature(t.h, i));
    }
    for (uint32_t k = 0; k <= n; k++) map_tokens.push_back(fmx_text_map(t.h, k));
  }

  size_t max_tokens_in_pattern() const { return _max_tokens; }
  fmx_index* handle() const { return _ix; }

 private:
  friend void import_binarized_fuzzy_matcher(const std::string&, FuzzyMatch&);
  friend void export_binarized_fuzzy_matcher(const std::string&, const FuzzyMatch&);
  static void check(int rc) { if (rc != FMX_OK) throw std::runtime_error(fmx_last_error()); }
  struct TextHandle {  // fmx_text of one sentence (RAII)
    fmx_text* h;
    TextHandle(int pt, const std::string& s) : h(fmx_text_tokenize(pt, s.data(), s.size())) { if (!h) throw std::bad_alloc(); }
    ~TextHandle() { fmx_text_free(h); }
    TextHandle(const TextHandle&) = delete;
    TextHandle& operator=(const TextHandle&) = delete;
    void fill(Sentence& real, Tokens& norm) const {
      const uint32_t n = fmx_text_num_tokens(h);
      for (uint32_t i = 0; i < n; i++) { norm.emplace_back(fmx_text_norm(h, i)); real.push_back(fmx_text_real(h, i)); }
      for (uint32_t k = 0; k < fmx_text_num_itoks(h); k++) real.set_itok(fmx_text_itok_index(h, k), fmx_text_itok(h, k));
    }
  };
  bool text_is_empty(const std::string& sentence) const {
    TextHandle t(_pt, sentence);
    return fmx_text_num_tokens(t.h) == 0;
  }
  // ids of the real tokens / penalty strings of `real` for a pattern of p tokens (the `real` argument of match())
  void real_ids(const Sentence& real, uint32_t p, std::vector<uint32_t>& q_real, std::vector<uint32_t>& q_itok) const {
    if (_real_differs) {
      const Tokens rt = (Tokens)real;
      q_real.resize(p);
      for (uint32_t j = 0; j < p; j++) {
        const std::string r = j < rt.size() ? rt[j] : std::string();
        q_real[j] = fmx_index_lookup_real(_ix, r.data(), r.size());
      }
    }
    if (!real.itoks().empty()) {
      q_itok.assign(p + 1, 0);
      for (const auto& kv : real.itoks())
        if (kv.first <= p) q_itok[kv.first] = fmx_index_intern_itok(_ix, kv.second.data(), kv.second.size());
    }
  }
  bool subsequence_tokens(const Sentence& real, const Tokens& pattern, unsigned number_of_matches, bool no_perfect,
                          int min_subseq_length, float min_subseq_ratio, bool idf_weighting, Match& m, uint32_t& pos) const {
    ensure_ready();
    const uint32_t p = (uint32_t)pattern.size();
    if (p == 0) return false;
    std::vector<const char*> toks;
    for (const auto& t : pattern) toks.push_back(t.c_str());
    std::vector<uint32_t> wids(p), q_real, q_itok;
    check(fmx_vocab_lookup(_ix, toks.data(), p, wids.data()));
    real_ids(real, p, q_real, q_itok);
    const uint32_t off[2] = {0, p};
    fmx_batch b;
    b.nq = 1; b.q_off = off; b.q_tok = wids.data();
    b.q_real = q_real.empty() ? nullptr : q_real.data();
    b.q_itok = q_itok.empty() ? nullptr : q_itok.data();
    b.max_len = p;
    fmx_match rec;
    uint32_t found = 0;
    check(fmx_subsequence_batch(_ix, &b, number_of_matches, no_perfect ? 1 : 0, min_subseq_length, min_subseq_ratio,
                                idf_weighting ? 1 : 0, &rec, &found, &pos));
    if (!found) return false;
    m = Match(_buffer + _sent_pos[rec.s_id] + 1, (int)_buffer[_sent_pos[rec.s_id]]);
    m.score = rec.score; m.penalty = 0; m.max_subseq = rec.max_subseq; m.s_id = rec.s_id;
    m.id = fmx_index_sentence_id(_ix, rec.s_id);
    return true;
  }
  // sort + derive + upload, once after any change (match() is const in the reference: the state is mutable here)
  void ensure_ready() const {
    if (_ready.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lk(_mu);
    if (_ready.load(std::memory_order_relaxed)) return;
    check(fmx_index_sort(_ix, _device, 0));
    const_cast<FuzzyMatch*>(this)->make_ready();
  }
  void make_ready() {
    check(fmx_index_finalize(_ix));
    // one GPU by default; FMX_DEVICES="0,1,2,3" (or set_devices() before the first match) replicates the index and
    // lets libfmx.so cut every batch into one block per GPU (include/fmx.h: fmx_index_to_devices)
    std::vector<int> devs = _devices;
    if (devs.empty()) {
      if (const char* e = getenv("FMX_DEVICES")) {
        for (const char* p = e; *p;) {
          char* end = nullptr;
          const long v = strtol(p, &end, 10);
          if (end == p) break;
          devs.push_back((int)v);
          p = *end == ',' ? end + 1 : end;
        }
      }
    }
    if (devs.size() > 1) check(fmx_index_to_devices(_ix, devs.data(), (int)devs.size()));
    else check(fmx_index_to_device(_ix, devs.empty() ? _device : devs[0]));
    fmx_index_info info;
    check(fmx_index_get_info(_ix, &info));
    _max_tokens = info.max_tokens_in_pattern;
    _pt = info.pt;
    _real_differs = info.real_differs != 0;
    const void* p; uint64_t n;
    check(fmx_index_array(_ix, FMX_A_BUFFER, &p, &n)); _buffer = (const unsigned*)p;
    check(fmx_index_array(_ix, FMX_A_SENT_POS, &p, &n)); _sent_pos = (const unsigned*)p;
    _ready.store(true, std::memory_order_release);
  }
  int _pt;
  size_t _max_tokens;
  int _device;
  std::vector<int> _devices;
  fmx_index* _ix;
  mutable std::mutex _mu;
  std::atomic<bool> _ready{false};
  bool _real_differs = false;
  const unsigned* _buffer = nullptr;
  const unsigned* _sent_pos = nullptr;
};

// include/fuzzy/fuzzy_matcher_binarization.hh:8-10; throws std::exception subclasses like the reference
inline void import_binarized_fuzzy_matcher(const std::string& path, FuzzyMatch& fm) {
  char err[512];
  fmx_index* ix = fmx_index_load(path.c_str(), err, sizeof(err));
  if (!ix) throw std::runtime_error(err);
  fmx_index_free(fm._ix);
  fm._ix = ix;
  fm._ready = false;
  fm.ensure_ready();  // the reference's import leaves a usable matcher; errors surface here, like there
}

inline void export_binarized_fuzzy_matcher(const std::string& path, const FuzzyMatch& fm) {
  fm.ensure_ready();  // the file holds the sorted suffix array
  if (fmx_index_save(fm._ix, path.c_str()) != FMX_OK) throw std::runtime_error(fmx_last_error());
}

}  // namespace fuzzy
