// fmx_cli.cc — CLI driver with the reference's flags, output format and timing lines
// (cli/src/FuzzyMatch-cli.cc:287-343 options, :219-233 output, :21-27 TICK, :450-467 NMATCH), on top of
// the C ABI.  Differences: input lines must already be tokenised (tokens separated by blanks: the
// OpenNMT tokenizer front-end is out of scope), and lines are matched in batches on the
// GPU instead of one per worker thread (-N is accepted and ignored).  Output order = input order.
//
//   fmx-cli -i tm.fmi -a match -f 0.7 -n 5 < patterns.tok > matches.tsv
//   fmx-cli -c corpus.tok -a index            # builds corpus.tok.fmi (ids = line numbers, or "id<TAB>tokens")
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../include/fmx.h"

static std::chrono::steady_clock::time_point g_start, g_period;
static void tick(const std::string& msg) {  // TICK — cli/src/FuzzyMatch-cli.cc:21-27
  auto now = std::chrono::steady_clock::now();
  const double el = std::chrono::duration<double>(now - g_period).count();
  const double tot = std::chrono::duration<double>(now - g_start).count();
  std::cerr << "STEP\t" << msg << "\tELAPSE\t" << el << "\tTOTAL\t" << tot << std::endl;
  g_period = now;
}

static std::vector<std::string> split_blank(const std::string& s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string t;
  while (is >> t) out.push_back(t);
  return out;
}

static int usage(int rc) {
  std::cerr << "FuzzyMatch cli options (MI355X build; input must be pre-tokenised, blank-separated):\n"
               "  -h [ --help ]\n  -a [ --action ] arg (=index)   index|match\n  -i [ --index ] arg             index file (.fmi)\n"
               "  -c [ --corpus ] arg            tokenised corpus to index: one sentence per line, optional 'id<TAB>' prefix\n"
               "  -f [ --fuzzy ] arg (=0.8)\n  --ml arg (=3)\n  --mr arg (=0.3)\n  -n [ --nmatch ] arg (=5)\n  -P [ --no-perfect ]\n"
               "  -I [ --idf-penalty ] arg (=0)\n  --insert-cost arg (=1)  --delete-cost arg (=1)  --replace-cost arg (=1)\n"
               "  --max-tokens-in-pattern arg (=300)\n  --contrast arg (=0)  --contrast-reduce arg (=mean)  --contrast-buffer arg (=-1)\n"
               "  -N [ --nthreads ] arg          accepted, ignored (batches run on the GPU)\n"
               "  --batch arg (=65536)           patterns per GPU batch\n  --device arg (=0)\n";
  return rc;
}

int main(int argc, char** argv) {
  g_start = g_period = std::chrono::steady_clock::now();
  std::string action = "index", index_file, corpus, reduce = "mean";
  fmx_params prm = {0.8f, 5, 0, 3, 0.3f, 0.f, 1.f, 1.f, 1.f, 0.f, 0, -1};
  uint64_t max_tokens = 300;
  size_t batch = 65536;
  int device = 0;
  for (int i = 1; i < argc; i++) {
    const std::string a = argv[i];
    auto val = [&]() -> const char* {
      if (i + 1 >= argc) { std::cerr << "ERROR: missing value for " << a << std::endl; exit(usage(2)); }
      return argv[++i];
    };
    if (a == "-h" || a == "--help") return usage(0);
    else if (a == "-a" || a == "--action") action = val();
    else if (a == "-i" || a == "--index") index_file = val();
    else if (a == "-c" || a == "--corpus") corpus = val();
    else if (a == "-f" || a == "--fuzzy") prm.fuzzy = (float)atof(val());
    else if (a == "--ml") prm.min_subseq_length = atoi(val());
    else if (a == "--mr") prm.min_subseq_ratio = (float)atof(val());
    else if (a == "-n" || a == "--nmatch") prm.nmatch = (uint32_t)atoi(val());
    else if (a == "-P" || a == "--no-perfect") prm.no_perfect = 1;
    else if (a == "-I" || a == "--idf-penalty") prm.idf_penalty = (float)atof(val());
    else if (a == "--insert-cost") prm.insert_cost = (float)atof(val());
    else if (a == "--delete-cost") prm.delete_cost = (float)atof(val());
    else if (a == "--replace-cost") prm.replace_cost = (float)atof(val());
    else if (a == "--max-tokens-in-pattern") max_tokens = strtoull(val(), nullptr, 10);
    else if (a == "--contrast") prm.contrast = (float)atof(val());
    else if (a == "--contrast-reduce") reduce = val();
    else if (a == "--contrast-buffer") prm.contrast_buffer = atoi(val());
    else if (a == "-N" || a == "--nthreads") (void)val();
    else if (a == "-p" || a == "--penalty-tokens") (void)val();  // fixed by the tokenised input
    else if (a == "--batch") batch = (size_t)strtoull(val(), nullptr, 10);
    else if (a == "--device") device = atoi(val());
    else { std::cerr << "ERROR: unknown option " << a << std::endl; return usage(2); }
  }
  prm.reduce = reduce == "max" ? 1 : 0;

  fmx_index* ix = nullptr;
  if (!index_file.empty()) {
    tick("Loading index");
    char err[512];
    ix = fmx_index_load(index_file.c_str(), err, sizeof(err));
    if (!ix) { std::cerr << "ERROR: " << err << std::endl; return 1; }
  } else if (!corpus.empty()) {
    tick("Indexing");
    ix = fmx_index_create(0, max_tokens);
    std::ifstream in(corpus);
    if (!in) { std::cerr << "ERROR: cannot open " << corpus << std::endl; return 1; }
    std::string line;
    size_t count = 0;
    while (std::getline(in, line)) {
      ++count;
      std::string id = std::to_string(count);
      const size_t tab = line.find('\t');
      if (tab != std::string::npos) { id = line.substr(0, tab); line = line.substr(tab + 1); }
      const auto toks = split_blank(line);
      if (toks.empty()) { std::cerr << "WARNING: cannot index empty segment: (" << id << ")" << std::endl; continue; }
      std::vector<const char*> ptr;
      for (const auto& t : toks) ptr.push_back(t.c_str());
      if (fmx_index_add_tm(ix, id.c_str(), ptr.data(), (uint32_t)ptr.size(), nullptr, 0, nullptr, nullptr) < -1) {
        std::cerr << "ERROR: " << fmx_last_error() << std::endl;
        return 1;
      }
    }
    tick("Sorting");
    if (fmx_index_sort(ix, device, 0) != FMX_OK) { std::cerr << "ERROR: " << fmx_last_error() << std::endl; return 1; }
    const std::string out = corpus + ".fmi";
    tick("Dump: " + out);
    if (fmx_index_save(ix, out.c_str()) != FMX_OK) { std::cerr << "ERROR: " << fmx_last_error() << std::endl; return 1; }
  } else {
    std::cerr << "ERROR: index file or corpus needs to be provided";
    return 3;
  }

  if (action == "match") {
    if (fmx_index_sort(ix, device, 0) != FMX_OK || fmx_index_finalize(ix) != FMX_OK || fmx_index_to_device(ix, device) != FMX_OK) {
      std::cerr << "ERROR: " << fmx_last_error() << std::endl;
      return 1;
    }
    tick("Matching");
    uint32_t cap = prm.nmatch ? prm.nmatch : 64;
    size_t n_total = 0, n_nonempty = 0;
    std::vector<std::string> lines;
    std::string line;
    bool eof = false;
    while (!eof) {
      lines.clear();
      while (lines.size() < batch && std::getline(std::cin, line)) lines.push_back(line);
      eof = lines.size() < batch;
      if (lines.empty()) break;
      std::vector<uint32_t> q_off(1, 0), wids;
      std::vector<std::string> toks_all;
      std::vector<const char*> ptr;
      for (const auto& l : lines) {
        const auto toks = split_blank(l);
        for (const auto& t : toks) toks_all.push_back(t);
        q_off.push_back((uint32_t)toks_all.size());
      }
      for (const auto& t : toks_all) ptr.push_back(t.c_str());
      wids.resize(ptr.size());
      if (fmx_vocab_lookup(ix, ptr.data(), ptr.size(), wids.data()) != FMX_OK) { std::cerr << "ERROR: " << fmx_last_error() << std::endl; return 1; }
      fmx_batch b = {lines.size(), q_off.data(), wids.data(), nullptr, nullptr, 0};
      std::vector<fmx_match> out;
      std::vector<uint32_t> cnt(lines.size()), tot(lines.size());
      for (;;) {  // nmatch == 0 (no limit): grow the per-query capacity until nothing is truncated
        out.resize(lines.size() * (size_t)cap);
        if (fmx_match_batch(ix, &prm, &b, out.data(), cnt.data(), tot.data(), cap, nullptr) != FMX_OK) {
          std::cerr << "ERROR: " << fmx_last_error() << std::endl;
          return 1;
        }
        uint32_t mx = 0;
        for (uint32_t t : tot) mx = t > mx ? t : mx;
        if (mx <= cap) break;
        cap = mx;
      }
      std::string buf;
      char num[64];
      for (size_t q = 0; q < lines.size(); q++) {
        for (uint32_t k = 0; k < cnt[q]; k++) {
          const fmx_match& m = out[q * cap + k];
          snprintf(num, sizeof(num), "%.9g", m.score);  // boost::lexical_cast<std::string>(float), cli:230
          if (k) buf += '\t';
          buf += num;
          buf += '\t';
          buf += fmx_index_sentence_id(ix, m.s_id);
        }
        buf += '\n';
        n_total++;
        if (cnt[q]) n_nonempty++;
      }
      std::cout << buf;
    }
    std::cerr << "NMATCH\t" << n_nonempty << "\t/\t" << n_total << std::endl;
  } else if (action != "index") {
    std::cerr << "ERROR: action '" << action << "' is not available in this build (match and index only)" << std::endl;
    return 2;
  }
  tick("Done");
  fmx_index_free(ix);
  return 0;
}
