// fmx_text.hh — the text front-end of FuzzyMatch: _tokenize_and_normalize (src/fuzzy_match.cc:108-193) over a
// RESTRICTED restatement of the tokenizer the reference configures in _update_tokenizer (src/fuzzy_match.cc:70-91):
// OpenNMT Tokenizer v1.20, Mode::Aggressive, flags SegmentAlphabetChange | NoSubstitution | SupportPriorJoiners
// (+ CaseFeature under pt_cas, + JoinerNew | JoinerAnnotate under pt_jnr, + SpacerNew | SpacerAnnotate under pt_sep),
// segmenting Han / Kanbun / Katakana / Hiragana / Thai / Myanmar.  That library is third-party code absent
// from this image, so its published behaviour is restated for the character classes listed in fmx_text.cc
// (Latin, Greek, Cyrillic, Armenian, Hebrew, Arabic, Indic, Thai, Myanmar, Georgian, Hangul, kana, Han, digits,
// ASCII / Latin-1 / general punctuation).  ICU's NFC normalisation (src/fuzzy_match.cc:43-58) IS applied, through the
// system's libicuuc bound at run time (fmx_text.cc: icu(); without one a warning is printed once and the input is taken
// as NFC).  It is pinned
// by the reference's own text fixtures (test/data/tm1, test-tm1, test-tm1-{pct,sep,jnr,nonbr-nocas}, tm2.en.gz +
// test-tm2: tests/test_text_frontend.py); text outside those classes is parity-unpinned.
#pragma once
#include <stdint.h>

#include <string>
#include <utility>
#include <vector>

namespace fmx {

enum PenaltyToken : int {  // FuzzyMatch::penalty_token, include/fuzzy/fuzzy_match.hh:20-28
  kPtNone = 0, kPtTag = 1 << 0, kPtPct = 1 << 1, kPtSep = 1 << 2, kPtJnr = 1 << 3, kPtNbr = 1 << 4, kPtCas = 1 << 5,
};

struct TextTokens {
  std::vector<std::string> raw;       // the tokenizer's tokens (joiner / spacer markers included)
  std::vector<std::string> feat;      // case feature per raw token ("N" everywhere without pt_cas)
  std::vector<std::string> norm;      // `pattern`: the normalised tokens
  std::vector<std::string> real;      // the real tokens of `real` (same count as norm)
  std::vector<std::pair<uint64_t, std::string>> itoks;  // `real`'s penalty strings by real-token index, ascending
  std::vector<uint32_t>    map_tokens;  // raw-token index after the k-th normalised token (k = 0: 0)
};

// onmt::Tokenizer::tokenize for the configuration above: raw tokens + case features
void tokenize_aggressive(int pt, const std::string& text, std::vector<std::string>& tokens, std::vector<std::string>& feats);
// FuzzyMatch::_tokenize_and_normalize
void tokenize_and_normalize(int pt, const std::string& sentence, TextTokens& out);
// ... into packed storage (bulk ingestion): normalised tokens NUL-terminated appended to `norm`, the real tokens tab-joined appended
// to `real` (Sentence::_tokstring), the penalty strings by real-token index
bool tokenize_and_normalize_packed(int pt, const char* sentence, size_t len, std::string& norm, uint32_t& n_tok, std::string& real,
                                   std::vector<std::pair<uint64_t, std::string>>& itoks);
// ... without the fast path for ASCII sentences (what the tests compare that path with)
void tokenize_and_normalize_general(int pt, const std::string& sentence, TextTokens& out);
// onmt::Tokenizer::detokenize(tokens, features) for the same configuration (subsequence() appends it to Match::id)
std::string detokenize(int pt, const std::vector<std::string>& tokens, const std::vector<std::string>& feats);

}  // namespace fmx
