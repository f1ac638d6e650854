// Exercises the C++ facade (fuzzy-match_amd/include/fuzzy/fuzzy_match.hh) the way the reference's gtest file
// uses its class: import a .fmi, match through the Sentence / Tokens overloads, print "ID score id ...".
// usage: facade_test <tm1.fmi> <cases.tsv>   (cases: id \t fuzzy \t no_perfect \t nmatch \t norm(space) \t real(space) \t itoks(idx=str,...))
#include <fuzzy/fuzzy_match.hh>

#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>

static std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  std::string cur;
  std::istringstream is(s);
  while (std::getline(is, cur, sep)) out.push_back(cur);
  return out;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  fuzzy::FuzzyMatch fm;
  try {
    fuzzy::FuzzyMatch missing;
    fuzzy::import_binarized_fuzzy_matcher(std::string(argv[1]) + ".does-not-exist", missing);
    std::puts("MISSING_FILE_DID_NOT_THROW");
    return 4;
  } catch (const std::exception&) {
    std::puts("MISSING_FILE_THROWS");  // test/test.cc:22-26
  }
  try {
    fuzzy::import_binarized_fuzzy_matcher(argv[1], fm);
  } catch (const std::exception& e) {
    std::printf("IMPORT_FAILED %s\n", e.what());
    return 3;
  }
  std::printf("MAX_TOKENS %zu\n", fm.max_tokens_in_pattern());
  std::ifstream in(argv[2]);
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty()) continue;
    const auto f = split(line, '\t');
    const float fuzzy = std::stof(f[1]);
    const bool np = f[2] == "1";
    const unsigned nmatch = (unsigned)std::stoul(f[3]);
    const fuzzy::Tokens norm = split(f[4], ' ');
    fuzzy::Sentence real;
    for (const auto& r : split(f[5], ' ')) real.push_back(r);
    if (f.size() > 6 && !f[6].empty())
      for (const auto& kv : split(f[6], ',')) {
        const auto eq = kv.find('=');
        real.set_itok(std::stoul(kv.substr(0, eq)), kv.substr(eq + 1));
      }
    std::vector<fuzzy::FuzzyMatch::Match> matches;
    const bool any = fm.match(real, norm, fuzzy, nmatch, np, matches, 2, 0);
    std::printf("CASE %s %d", f[0].c_str(), any ? 1 : 0);
    for (const auto& m : matches) std::printf("\t%.9g\t%s\t%d\t%u", m.score, m.id.c_str(), m.length, m.s[0]);
    std::printf("\n");
  }
  // Tokens overload + appending semantics (matches is never cleared; test/test.cc:288-301 relies on it)
  std::vector<fuzzy::FuzzyMatch::Match> matches;
  fm.match(fuzzy::Tokens{"aa", "bb", "cc", "dd"}, 0.6f, 0, matches);
  const size_t first = matches.size();
  fm.match(fuzzy::Tokens{"aa", "bb", "cc", "dd"}, 0.6f, 0, matches);
  std::printf("APPEND %zu %zu\n", first, matches.size());
  return 0;
}
