// fmi_fuzz.cc — the .fmi reader / writer and the host side of the C ABI under AddressSanitizer + UBSan (CPU build, see
// gpu_stub.cc): the reference's fixtures round-trip, then N mutated copies of a fixture — bit flips, byte overwrites, 64-bit
// fields blown up, truncations, extensions — go through fmx_index_load; whatever loads is finalized, queried for vocabulary
// ids and saved again.  A corrupt file must be refused (NULL + message) or load into something every later call survives:
// the sanitizers end the run on any out-of-bounds access, overflow or leak.
//   usage: fmi_fuzz <fixture.fmi> <other.fmi ...> -- <n_mutations> <scratch dir> [text fixture ...]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <string>
#include <vector>

#include "fmx.h"

static std::vector<char> slurp(const std::string& p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}
static void spit(const std::string& p, const std::vector<char>& d) {
  std::ofstream f(p, std::ios::binary);
  f.write(d.data(), (std::streamsize)d.size());
}
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {  // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int exercise(fmx_index* ix, const std::string& scratch) {
  // everything a caller does with a freshly loaded index short of matching (there is no device here)
  fmx_index_info info;
  if (fmx_index_get_info(ix, &info) != FMX_OK) return 1;
  (void)fmx_index_finalize(ix);  // may refuse (unsorted flag flipped, ...): must not crash
  const char* toks[] = {"aa", "bb", "zz", ".", "", "\xef\xbd\x9fnum\xef\xbd\xa0"};
  uint32_t ids[6];
  (void)fmx_vocab_lookup(ix, toks, 6, ids);
  (void)fmx_vocab_lookup_mt(ix, toks, 6, ids, 2);
  for (uint32_t s = 0; s < 4; s++) (void)fmx_index_sentence_id(ix, s);
  for (int which = 0; which < 16; which++) {
    const void* data = nullptr;
    uint64_t count = 0;
    (void)fmx_index_array(ix, which, &data, &count);
  }
  const std::string out = scratch + "/resaved.fmi";
  if (fmx_index_save(ix, out.c_str()) == FMX_OK) {
    char err[256];
    fmx_index* again = fmx_index_load(out.c_str(), err, sizeof err);
    if (!again) { fprintf(stderr, "a file this build wrote does not load: %s\n", err); return 1; }
    fmx_index_free(again);
  }
  return 0;
}

int main(int argc, char** argv) {
  std::vector<std::string> fixtures;
  int k = 1;
  for (; k < argc && strcmp(argv[k], "--") != 0; k++) fixtures.push_back(argv[k]);
  if (fixtures.empty() || k + 2 >= argc) { fprintf(stderr, "usage: fmi_fuzz <fixture.fmi>... -- <n> <scratch dir>\n"); return 2; }
  const long n_mut = atol(argv[k + 1]);
  const std::string scratch = argv[k + 2];
  char err[512];
  // 1. the fixtures themselves: load, exercise, save, load again; the current-layout fixture re-exports byte for byte
  for (const auto& f : fixtures) {
    fmx_index* ix = fmx_index_load(f.c_str(), err, sizeof err);
    if (!ix) { fprintf(stderr, "%s does not load: %s\n", f.c_str(), err); return 1; }
    const std::string out = scratch + "/copy.fmi";
    if (fmx_index_save(ix, out.c_str()) != FMX_OK) { fprintf(stderr, "save failed: %s\n", fmx_last_error()); return 1; }
    if (&f == &fixtures[0] && slurp(out) != slurp(f)) { fprintf(stderr, "%s: load -> save is not byte-identical\n", f.c_str()); return 1; }
    if (exercise(ix, scratch)) return 1;
    fmx_index_free(ix);
  }
  // 1b. the text front-end and the index builder under the sanitizers: every line of the text fixtures handed over (argv after
  // the scratch dir) tokenised in each penalty-token mode — also cut at every byte (truncated UTF-8 sequences) — every accessor
  // read, and an index built from the lines, sorted on the host, finalized, saved and loaded again
  for (int f = k + 3; f < argc; f++) {
    std::ifstream in(argv[f]);
    std::vector<std::string> lines;
    for (std::string line; std::getline(in, line);) lines.push_back(line);
    const int32_t pts[] = {0, 1, 2, 4, 8, 16, 32, 49, 63};
    for (int32_t pt : pts) {
      fmx_index* ix = fmx_index_create(pt, 300);
      size_t n_line = 0;
      for (const auto& line : lines) {
        for (size_t cut = (n_line < 40 ? 0 : line.size()); cut <= line.size(); cut++) {
          fmx_text* t = fmx_text_tokenize(pt, line.data(), cut);
          if (!t) continue;
          const uint32_t n = fmx_text_num_tokens(t);
          size_t sink = 0;
          for (uint32_t i = 0; i < n; i++) { sink += strlen(fmx_text_norm(t, i)) + strlen(fmx_text_real(t, i)); (void)fmx_text_map(t, i); }
          (void)fmx_text_map(t, n);
          for (uint32_t i = 0; i < fmx_text_num_itoks(t); i++) { sink += strlen(fmx_text_itok(t, i)); (void)fmx_text_itok_index(t, i); }
          for (uint32_t i = 0; i < fmx_text_num_raw(t); i++) sink += strlen(fmx_text_raw(t, i)) + strlen(fmx_text_feature(t, i));
          if (n) { const char* d = fmx_text_detokenize(t, 0, n); sink += d ? strlen(d) : 0; }
          (void)sink;
          fmx_text_free(t);
        }
        const std::string id = std::to_string(n_line++);
        (void)fmx_index_add_text(ix, id.c_str(), line.data(), line.size());
      }
      if (fmx_index_sort(ix, -1, 2) != FMX_OK) { fprintf(stderr, "host sort failed: %s\n", fmx_last_error()); return 1; }
      if (fmx_index_finalize(ix) != FMX_OK) { fprintf(stderr, "finalize failed: %s\n", fmx_last_error()); return 1; }
      if (exercise(ix, scratch)) return 1;
      fmx_index_free(ix);
    }
  }
  // 2. mutated copies
  long loaded = 0, refused = 0;
  for (long it = 0; it < n_mut; it++) {
    std::vector<char> d = slurp(fixtures[(size_t)it % fixtures.size()]);
    const int kind = (int)(rnd() % 6);
    if (kind == 0) {            // a few bit flips anywhere
      for (int b = 1 + (int)(rnd() % 8); b > 0; b--) d[rnd() % d.size()] ^= (char)(1u << (rnd() % 8));
    } else if (kind == 1) {     // truncation
      d.resize(rnd() % d.size());
    } else if (kind == 2) {     // one byte overwritten
      d[rnd() % d.size()] = (char)rnd();
    } else if (kind == 3) {     // an aligned-or-not 64-bit field blown up (collection sizes, string lengths)
      const size_t at = rnd() % (d.size() - 8);
      const uint64_t v = (rnd() % 3 == 0) ? ~0ull : (1ull << (20 + rnd() % 40)) + (rnd() & 0xffff);
      memcpy(&d[at], &v, 8);
    } else if (kind == 4) {     // a 32-bit field set to a large / boundary value
      const size_t at = rnd() % (d.size() - 4);
      const uint32_t v = (rnd() % 2) ? 0xffffffffu : (uint32_t)rnd();
      memcpy(&d[at], &v, 4);
    } else {                    // bytes appended, or a slice duplicated in place
      if (rnd() % 2) for (int b = 1 + (int)(rnd() % 64); b > 0; b--) d.push_back((char)rnd());
      else { const size_t a = rnd() % d.size(), n = rnd() % 64; d.insert(d.begin() + (long)a, d.begin() + (long)a, d.begin() + (long)std::min(d.size(), a + n)); }
    }
    const std::string p = scratch + "/mut.fmi";
    spit(p, d);
    err[0] = 0;
    fmx_index* ix = fmx_index_load(p.c_str(), err, sizeof err);
    if (!ix) { refused++; if (!err[0]) { fprintf(stderr, "mutation %ld: refused without a message\n", it); return 1; } continue; }
    loaded++;
    if (exercise(ix, scratch)) { fprintf(stderr, "mutation %ld (kind %d) broke a later call\n", it, kind); return 1; }
    fmx_index_free(ix);
  }
  printf("fmi_fuzz ok: %zu fixtures, %ld mutations (%ld refused, %ld loaded and exercised)\n", fixtures.size(), n_mut, refused, loaded);
  return 0;
}
