// fuzzy/fuzzy_match.hh — C++ facade that re-creates the reference's public class on top of the C ABI
// (include/fmx.h).  Same names, argument meaning, defaults and error behaviour as
//   include/fuzzy/fuzzy_match.hh:17-147           class fuzzy::FuzzyMatch, Match, penalty_token, ContrastReduce
//   include/fuzzy/costs.hh:7-29                   EditCosts
//   include/fuzzy/sentence.hh:22-48               Tokens, Sentence
//   include/fuzzy/fuzzy_matcher_binarization.hh   import_/export_binarized_fuzzy_matcher
// of SYSTRAN/fuzzy-match, so a caller of libFuzzyMatch.so re-links against libfmx.so and keeps its
// source.  Header only; every call goes through the extern "C" entry points.
//
// Not provided (out of scope of the match() hot path): the std::string overloads
// that run the OpenNMT tokenizer (add_tm(id, string), match(string, ...)) and subsequence().  They exist
// so that callers compile, and throw std::logic_error.
#pragma once
#include <stdint.h>

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
  bool add_tm(const std::string& id, const Sentence& source, const Tokens& norm, bool sort = true) {
    std::vector<const char*> toks;
    for (const auto& t : norm) toks.push_back(t.c_str());
    std::vector<uint64_t> idx;
    std::vector<const char*> strs;
    for (const auto& kv : source.itoks()) { idx.push_back(kv.first); strs.push_back(kv.second.c_str()); }
    const int64_t r = fmx_index_add_tm(_ix, id.c_str(), toks.data(), (uint32_t)toks.size(), source.tokstring().c_str(),
                                       (uint32_t)idx.size(), idx.data(), strs.data());
    if (r < -1) throw std::runtime_error(fmx_last_error());
    if (sort) this->sort();
    return true;
  }
  bool add_tm(const std::string&, const std::string&, bool = true) {
    throw std::logic_error("add_tm(id, text): the tokenizer front-end is outside the MI355X match() build");
  }

  // SuffixArray::sort then upload: the index becomes immutable (the reference allows further add_tm
  // calls after sort(); this build freezes the index at the first sort()).
  void sort() {
    if (_ready) return;
    check(fmx_index_sort(_ix, _device, 0));
    make_ready();
  }

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
    if (!_ready) throw std::logic_error("FuzzyMatch::match before sort()/import");
    const uint32_t p = (uint32_t)pattern.size();
    if (p == 0 || p > _max_tokens) return false;  // src/fuzzy_match.cc:455-461
    std::vector<const char*> toks;
    for (const auto& t : pattern) toks.push_back(t.c_str());
    std::vector<uint32_t> wids(p), q_real, q_itok;
    check(fmx_vocab_lookup(_ix, toks.data(), p, wids.data()));
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
    fmx_params prm;
    prm.fuzzy = fuzzy; prm.nmatch = number_of_matches; prm.no_perfect = no_perfect;
    prm.min_subseq_length = min_subseq_length; prm.min_subseq_ratio = min_subseq_ratio;
    prm.idf_penalty = vocab_idf_penalty;
    prm.insert_cost = edit_costs.insert_cost; prm.delete_cost = edit_costs.delete_cost;
    prm.replace_cost = edit_costs.replace_cost;
    prm.contrast = contrastive_factor; prm.reduce = reduce == ContrastReduce::MAX ? 1 : 0;
    prm.contrast_buffer = contrast_buffer;
    const uint32_t q_off[2] = {0, p};
    fmx_batch b;
    b.nq = 1; b.q_off = q_off; b.q_tok = wids.data();
    b.q_real = q_real.empty() ? nullptr : q_real.data();
    b.q_itok = q_itok.empty() ? nullptr : q_itok.data();
    b.max_len = p;
    uint32_t cap = number_of_matches ? number_of_matches : 64;
    std::vector<fmx_match> out;
    uint32_t count = 0, total = 0;
    for (;;) {  // number_of_matches == 0 means "all": grow until everything fits
      out.resize(cap);
      check(fmx_match_batch(_ix, &prm, &b, out.data(), &count, &total, cap, nullptr));
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

  bool match(const std::string&, float, unsigned, bool, std::vector<Match>&, int = 3, float = 0.3, float = 0,
             const EditCosts& = EditCosts(), float = 0, ContrastReduce = ContrastReduce::MEAN, int = -1) const {
    throw std::logic_error("match(text, ...): the tokenizer front-end is outside the MI355X match() build");
  }
  bool subsequence(const std::string&, unsigned, bool, std::vector<Match>&, int = 3, float = 0.3, bool = false) const {
    throw std::logic_error("subsequence(): outside the MI355X match() build");
  }

  size_t max_tokens_in_pattern() const { return _max_tokens; }
  fmx_index* handle() const { return _ix; }

 private:
  friend void import_binarized_fuzzy_matcher(const std::string&, FuzzyMatch&);
  friend void export_binarized_fuzzy_matcher(const std::string&, const FuzzyMatch&);
  static void check(int rc) { if (rc != FMX_OK) throw std::runtime_error(fmx_last_error()); }
  void make_ready() {
    check(fmx_index_finalize(_ix));
    check(fmx_index_to_device(_ix, _device));
    fmx_index_info info;
    check(fmx_index_get_info(_ix, &info));
    _max_tokens = info.max_tokens_in_pattern;
    _pt = info.pt;
    _real_differs = info.real_differs != 0;
    const void* p; uint64_t n;
    check(fmx_index_array(_ix, FMX_A_BUFFER, &p, &n)); _buffer = (const unsigned*)p;
    check(fmx_index_array(_ix, FMX_A_SENT_POS, &p, &n)); _sent_pos = (const unsigned*)p;
    _ready = true;
  }
  int _pt;
  size_t _max_tokens;
  int _device;
  fmx_index* _ix;
  bool _ready = false, _real_differs = false;
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
  if (fmx_index_sort(ix, fm._device, 0) != FMX_OK) throw std::runtime_error(fmx_last_error());  // no-op when sorted
  fm.make_ready();
}

inline void export_binarized_fuzzy_matcher(const std::string& path, const FuzzyMatch& fm) {
  if (fmx_index_save(fm._ix, path.c_str()) != FMX_OK) throw std::runtime_error(fmx_last_error());
}

}  // namespace fuzzy
