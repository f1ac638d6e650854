// fmx_cli.cc — CLI driver with the reference's flags, output format and timing lines
// (cli/src/FuzzyMatch-cli.cc:287-343 options, :219-248 output, :21-27 TICK, :450-467 NMATCH), on top of
// the C ABI.  Like the reference it takes raw sentences (tokenised by the library's restricted text front-end,
// include/fmx.h fmx_text_*; penalty tokens from -p when indexing, from the .fmi when matching); --tokenised
// switches to blank-separated pre-tokenised input.  Lines are matched in batches on the GPU instead of one per
// worker thread; -N sets the helper threads that tokenise and format around it.
// Matching is a streaming pipeline (reader -> parser -> GPU -> writer).  Output order = input order.
//
//   fmx-cli -c tm.en.gz,tm.fr.gz --add-target          # builds tm.en.gz.fmi (ids "<line>=<target>", cli:29-81)
//   fmx-cli -i tm.en.gz.fmi -a match -f 0.7 -n 5 < patterns.txt > matches.tsv
//   fmx-cli -i tm.en.gz.fmi -a subseq -w < patterns.txt # score, length and text of the best sub-n-gram (cli:234-248)
//   fmx-cli --tokenised -c corpus.tok -a index          # ids = line numbers, or "id<TAB>tokens"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <zlib.h>

#include "../include/fmx.h"

static std::chrono::steady_clock::time_point g_start, g_period;
static void tick(const std::string& msg) {  // TICK — cli/src/FuzzyMatch-cli.cc:21-27
  auto now = std::chrono::steady_clock::now();
  const double el = std::chrono::duration<double>(now - g_period).count();
  const double tot = std::chrono::duration<double>(now - g_start).count();
  std::cerr << "STEP\t" << msg << "\tELAPSE\t" << el << "\tTOTAL\t" << tot << std::endl;
  g_period = now;
}

// bounded single-producer / single-consumer hand-over between pipeline stages
template <class T>
class Channel {
 public:
  explicit Channel(size_t cap) : cap_(cap) {}
  bool push(T v) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return q_.size() < cap_ || closed_; });
    if (closed_) return false;
    q_.push_back(std::move(v));
    cv_.notify_all();
    return true;
  }
  bool pop(T& v) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !q_.empty() || closed_; });
    if (q_.empty()) return false;
    v = std::move(q_.front());
    q_.pop_front();
    cv_.notify_all();
    return true;
  }
  void close() {
    std::lock_guard<std::mutex> lk(mu_);
    closed_ = true;
    cv_.notify_all();
  }

 private:
  std::mutex mu_;
  std::condition_variable cv_;
  std::deque<T> q_;
  size_t cap_;
  bool closed_ = false;
};

// [0, n) split into `parts` contiguous chunks, one helper thread each: f(part, lo, hi)
template <class F>
static void parallel_for(unsigned parts, size_t n, F f) {
  if (parts <= 1 || n < 2 * parts) { f(0u, (size_t)0, n); return; }
  std::vector<std::thread> th;
  for (unsigned t = 0; t < parts; t++) th.emplace_back([&, t] { f(t, n * t / parts, n * (t + 1) / parts); });
  for (auto& x : th) x.join();
}

static std::vector<std::string> split_blank(const std::string& s) {
  std::vector<std::string> out;
  std::istringstream is(s);
  std::string t;
  while (is >> t) out.push_back(t);
  return out;
}

// lines of a plain or gzipped file (import_tm reads both, cli:29-47); zlib's gz* functions read plain files as they are
class LineFile {
 public:
  explicit LineFile(const std::string& path) : f_(gzopen(path.c_str(), "rb")) { if (f_) gzbuffer(f_, 1 << 20); }
  ~LineFile() { if (f_) gzclose(f_); }
  bool ok() const { return f_ != nullptr; }
  bool getline(std::string& out) {
    out.clear();
    char buf[65536];
    bool any = false;
    while (gzgets(f_, buf, sizeof(buf))) {
      any = true;
      const size_t n = strlen(buf);
      if (n && buf[n - 1] == '\n') { out.append(buf, n - 1); return true; }
      out.append(buf, n);
    }
    return any;
  }

 private:
  gzFile f_;
};

static int usage(int rc) {
  std::cerr << "FuzzyMatch cli options (MI355X build):\n"
               "  -h [ --help ]\n  -a [ --action ] arg (=index)   index|match|subseq\n  -i [ --index ] arg             index file (.fmi)\n"
               "  -c [ --corpus ] arg            corpus to index: bitext (src<TAB>tgt) or 2 files comma-separated, plain or .gz\n"
               "  --add-target                   ids become '<line>=<target>'\n  --add-target-no-index          ids become '<target>'\n"
               "  -p [ --penalty-tokens ] arg (=nbr,tag,cas)  none, or a list of cas,nbr,tag,sep|jnr,pct (when indexing)\n"
               "  -w [ --subseq-idf-weighting ]  idf weighting in finding the longest subsequence\n"
               "  --tokenised                    input is blank-separated tokens (corpus lines: optional 'id<TAB>' prefix)\n"
               "  -f [ --fuzzy ] arg (=0.8)\n  --ml arg (=3)\n  --mr arg (=0.3)\n  -n [ --nmatch ] arg (=5)\n  -P [ --no-perfect ]\n"
               "  -I [ --idf-penalty ] arg (=0)\n  --insert-cost arg (=1)  --delete-cost arg (=1)  --replace-cost arg (=1)\n"
               "  --max-tokens-in-pattern arg (=300)\n  --contrast arg (=0)  --contrast-reduce arg (=mean)  --contrast-buffer arg (=-1)\n"
               "  -N [ --nthreads ] arg          helper threads for parsing / formatting (matching runs on the GPU)\n"
               "  --batch arg (=65536)           patterns per GPU batch\n  --device arg (=0)\n"
               "  --devices a,b,...              replicate the index on several GPUs: every batch is cut into one block per GPU\n";
  return rc;
}

int main(int argc, char** argv) {
  g_start = g_period = std::chrono::steady_clock::now();
  std::ios::sync_with_stdio(false);  // std::getline on std::cin is several times faster without the stdio lock-step
  std::cin.tie(nullptr);
  std::string action = "index", index_file, corpus, reduce = "mean", penalty_tokens = "nbr,tag,cas";
  fmx_params prm = {0.8f, 5, 0, 3, 0.3f, 0.f, 1.f, 1.f, 1.f, 0.f, 0, -1};
  uint64_t max_tokens = 300;
  size_t batch = 65536;
  unsigned nthreads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
  int device = 0;
  std::vector<int> devices;
  bool tokenised = false, add_target = false, add_target_no_index = false, idf_weighting = false;
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
    else if (a == "-N" || a == "--nthreads") nthreads = (unsigned)std::max(1, atoi(val()));
    else if (a == "-p" || a == "--penalty-tokens") penalty_tokens = val();
    else if (a == "-w" || a == "--subseq-idf-weighting") idf_weighting = true;
    else if (a == "--add-target") add_target = true;
    else if (a == "--add-target-no-index") add_target_no_index = true;
    else if (a == "--tokenised" || a == "--tokenized") tokenised = true;
    else if (a == "--batch") batch = (size_t)strtoull(val(), nullptr, 10);
    else if (a == "--device") device = atoi(val());
    else if (a == "--devices") {
      std::string v = val();
      for (size_t i = 0; i <= v.size();) {
        size_t j = v.find(',', i);
        if (j == std::string::npos) j = v.size();
        if (j > i) devices.push_back(atoi(v.substr(i, j - i).c_str()));
        i = j + 1;
      }
      if (!devices.empty()) device = devices[0];
    }
    else { std::cerr << "ERROR: unknown option " << a << std::endl; return usage(2); }
  }
  prm.reduce = reduce == "max" ? 1 : 0;
  // --penalty-tokens (cli/src/FuzzyMatch-cli.cc:371-396)
  int pt = 0;
  if (penalty_tokens.empty()) { std::cerr << "ERROR: the argument for option '--penalty-tokens' is invalid"; return 1; }
  if (penalty_tokens != "none") {
    std::istringstream is(penalty_tokens);
    std::string tok;
    while (std::getline(is, tok, ',')) {
      const int apt = tok == "tag" ? 1 : tok == "pct" ? 2 : tok == "sep" ? 4 : tok == "jnr" ? 8 : tok == "nbr" ? 16 : tok == "cas" ? 32 : 0;
      if (!apt || (pt & apt)) { std::cerr << "ERROR: the argument ('" << tok << "') for option '--penalty-tokens' is invalid"; return 1; }
      pt |= apt;
    }
    if ((pt & 4) && (pt & 8)) { std::cerr << "ERROR: the argument ('sep/jnr') for option '--penalty-tokens' is invalid"; return 1; }
  }

  fmx_index* ix = nullptr;
  if (!index_file.empty()) {
    tick("Loading index_file: " + index_file);
    char err[512];
    ix = fmx_index_load(index_file.c_str(), err, sizeof(err));
    if (!ix) { std::cerr << "ERROR: " << err << std::endl; return 1; }
  } else if (!corpus.empty()) {
    tick("Importing TM: " + corpus);
    ix = fmx_index_create(tokenised ? 0 : pt, max_tokens);
    // import_tm (cli/src/FuzzyMatch-cli.cc:29-81): bitext, or two files separated by a comma
    std::string tgt_file;
    const size_t comma = corpus.find(',');
    if (comma != std::string::npos) { tgt_file = corpus.substr(comma + 1); corpus.erase(comma); }
    LineFile in(corpus);
    if (!in.ok()) { std::cerr << "ERROR: cannot open " << corpus << std::endl; return 1; }
    std::unique_ptr<LineFile> tin;
    if (!tgt_file.empty()) {
      tin.reset(new LineFile(tgt_file));
      if (!tin->ok()) { std::cerr << "ERROR: cannot open " << tgt_file << std::endl; return 1; }
    }
    std::string line, tgt;
    size_t count = 0;
    while (in.getline(line)) {
      ++count;
      std::string id = std::to_string(count);
      int64_t sid;
      if (tokenised) {
        const size_t tab = line.find('\t');
        if (tab != std::string::npos) { id = line.substr(0, tab); line = line.substr(tab + 1); }
        const auto toks = split_blank(line);
        if (toks.empty()) { std::cerr << "WARNING: cannot index empty segment: (" << id << ")" << std::endl; continue; }
        std::vector<const char*> ptr;
        for (const auto& t : toks) ptr.push_back(t.c_str());
        sid = fmx_index_add_tm(ix, id.c_str(), ptr.data(), (uint32_t)ptr.size(), nullptr, 0, nullptr, nullptr);
      } else {
        tgt.clear();
        if (tin) tin->getline(tgt);
        else {
          const size_t tab = line.find('\t');
          if (tab != std::string::npos) { tgt = line.substr(tab + 1); line.erase(tab); }
        }
        if (add_target) id += "=" + tgt;
        if (add_target_no_index) id = tgt;
        sid = fmx_index_add_text(ix, id.c_str(), line.data(), line.size());
        if (sid == FMX_TM_SKIPPED) std::cerr << "WARNING: cannot index empty segment: " << line << " (" << id << ")" << std::endl;
      }
      if (sid < 0 && sid != FMX_TM_SKIPPED) {
        std::cerr << "ERROR: " << fmx_last_error() << std::endl;
        return 1;
      }
    }
    tick("Sorting Index");
    if (fmx_index_sort(ix, device, 0) != FMX_OK) { std::cerr << "ERROR: " << fmx_last_error() << std::endl; return 1; }
    if (action == "index") {
      const std::string out = corpus + ".fmi";
      tick("Dump: " + out);
      if (fmx_index_save(ix, out.c_str()) != FMX_OK) { std::cerr << "ERROR: " << fmx_last_error() << std::endl; return 1; }
    }
  } else {
    std::cerr << "ERROR: index file or corpus needs to be provided";
    return 3;
  }

  if (action == "match" || action == "subseq") {
    const bool subseq = action == "subseq";
    if (fmx_index_sort(ix, device, 0) != FMX_OK || fmx_index_finalize(ix) != FMX_OK ||
        (devices.size() > 1 ? fmx_index_to_devices(ix, devices.data(), (int)devices.size()) : fmx_index_to_device(ix, device)) != FMX_OK) {
      std::cerr << "ERROR: " << fmx_last_error() << std::endl;
      return 1;
    }
    fmx_index_info info;
    if (fmx_index_get_info(ix, &info) != FMX_OK) { std::cerr << "ERROR: " << fmx_last_error() << std::endl; return 1; }
    const int ix_pt = info.pt;  // penalty tokens are a property of the index (src/fuzzy_matcher_binarization.cc:32-51)
    tick(subseq ? "Subsequencing" : "Matching");
    // Streaming pipeline (the reference keeps 1000 futures in flight over a worker pool, cli:112-193): batches
    // of lines flow reader -> parser (tokens + vocabulary ids, `nthreads` helpers) -> GPU -> formatter/writer
    // through bounded queues, so reading and parsing batch i + 1 and printing batch i - 1 overlap the GPU work
    // of batch i.  Output order = input order.
    struct Job {
      std::vector<std::string> lines;
      std::vector<uint32_t> q_off, wids, q_real, q_itok, cnt, tot, pos;
      std::vector<fmx_text*> text;  // subseq on raw text: kept for the detokenised sub-n-gram
      std::vector<fmx_match> out;
      uint32_t cap = 0;
      bool has_real = false, has_itok = false;
      ~Job() { for (fmx_text* t : text) fmx_text_free(t); }
    };
    using JobPtr = std::unique_ptr<Job>;
    Channel<JobPtr> to_parse(2), to_gpu(2), to_write(2);
    std::atomic<bool> failed{false};
    std::string fail_msg;
    std::mutex fail_mu;
    auto fail = [&](const std::string& msg) {
      std::lock_guard<std::mutex> lk(fail_mu);
      if (!failed.exchange(true)) fail_msg = msg;
    };
    const unsigned helpers = std::max(1u, nthreads);

    std::thread parser([&] {
      JobPtr job;
      while (to_parse.pop(job)) {
        const size_t n = job->lines.size();
        // per helper: ids, real-token ids, (slot, penalty-string id) pairs, token counts
        std::vector<std::vector<uint32_t>> ids(helpers), reals(helpers), counts(helpers);
        std::vector<std::vector<std::pair<uint64_t, uint32_t>>> itoks(helpers);  // (line << 32 | position, id)
        if (!tokenised && subseq) job->text.assign(n, nullptr);
        parallel_for(helpers, n, [&](unsigned t, size_t lo, size_t hi) {
          std::vector<const char*> ptr;
          for (size_t i = lo; i < hi; i++) {
            std::string& l = job->lines[i];
            ptr.clear();
            fmx_text* th = nullptr;
            if (tokenised) {
              // tokens in place: blanks become terminators
              size_t k = 0, len = l.size();
              while (k < len) {
                while (k < len && (l[k] == ' ' || l[k] == '\t' || l[k] == '\r')) l[k++] = '\0';
                if (k >= len) break;
                ptr.push_back(l.data() + k);
                while (k < len && l[k] != ' ' && l[k] != '\t' && l[k] != '\r') k++;
              }
            } else {
              th = fmx_text_tokenize(ix_pt, l.data(), l.size());
              if (!th) { fail("out of memory"); counts[t].push_back(0); continue; }
              const uint32_t nt = fmx_text_num_tokens(th);
              for (uint32_t j = 0; j < nt; j++) ptr.push_back(fmx_text_norm(th, j));
            }
            const size_t base = ids[t].size();
            ids[t].resize(base + ptr.size());
            // (the batch lookup: an open-addressing table over the packed spellings, no std::string per token)
            if (!ptr.empty() && fmx_vocab_lookup_mt(ix, ptr.data(), ptr.size(), ids[t].data() + base, 1) != FMX_OK) fail(fmx_last_error());
            counts[t].push_back((uint32_t)ptr.size());
            if (th) {
              if (info.real_differs)
                for (uint32_t j = 0; j < ptr.size(); j++) {
                  const char* r = fmx_text_real(th, j);
                  reals[t].push_back(fmx_index_lookup_real(ix, r, strlen(r)));
                }
              for (uint32_t k = 0; k < fmx_text_num_itoks(th); k++) {
                const char* is = fmx_text_itok(th, k);
                const uint64_t at = fmx_text_itok_index(th, k);
                if (at <= ptr.size()) itoks[t].emplace_back(((uint64_t)i << 32) | at, fmx_index_intern_itok(ix, is, strlen(is)));
              }
              if (subseq) job->text[i] = th; else fmx_text_free(th);
            }
          }
        });
        job->q_off.assign(1, 0);
        job->q_off.reserve(n + 1);
        size_t total = 0, n_itok = 0;
        for (unsigned t = 0; t < helpers; t++) { total += ids[t].size(); n_itok += itoks[t].size(); }
        job->wids.reserve(total);
        job->has_real = !tokenised && info.real_differs;
        job->has_itok = n_itok != 0;
        for (unsigned t = 0; t < helpers; t++) {
          for (uint32_t c : counts[t]) job->q_off.push_back(job->q_off.back() + c);
          job->wids.insert(job->wids.end(), ids[t].begin(), ids[t].end());
          if (job->has_real) job->q_real.insert(job->q_real.end(), reals[t].begin(), reals[t].end());
        }
        if (job->has_itok) {  // slot of (line q, position j) = q_off[q] + q + j
          job->q_itok.assign(total + n, 0);
          for (unsigned t = 0; t < helpers; t++)
            for (const auto& kv : itoks[t]) {
              const uint64_t q = kv.first >> 32, j = kv.first & 0xffffffffu;
              job->q_itok[job->q_off[q] + q + j] = kv.second;
            }
        }
        if (!to_gpu.push(std::move(job))) break;
      }
      to_gpu.close();
      to_parse.close();  // unblocks the reader if this stage stopped early
    });

    std::thread gpu([&] {
      JobPtr job;
      uint32_t cap = subseq ? 1u : (prm.nmatch ? prm.nmatch : 64);
      while (to_gpu.pop(job)) {
        const size_t n = job->lines.size();
        fmx_batch b = {n, job->q_off.data(), job->wids.data(), job->has_real ? job->q_real.data() : nullptr,
                       job->has_itok ? job->q_itok.data() : nullptr, 0};
        job->cnt.resize(n);
        job->tot.resize(n);
        if (subseq) {
          job->out.resize(n);
          job->pos.resize(n);
          if (fmx_subsequence_batch(ix, &b, prm.nmatch, prm.no_perfect, prm.min_subseq_length, prm.min_subseq_ratio,
                                    idf_weighting ? 1 : 0, job->out.data(), job->cnt.data(), job->pos.data()) != FMX_OK)
            fail(fmx_last_error());
        } else {
          for (;;) {  // nmatch == 0 (no limit): grow the per-query capacity until nothing is truncated
            job->out.resize(n * (size_t)cap);
            if (fmx_match_batch(ix, &prm, &b, job->out.data(), job->cnt.data(), job->tot.data(), cap, nullptr) != FMX_OK) {
              fail(fmx_last_error());
              break;
            }
            uint32_t mx = 0;
            for (uint32_t t : job->tot) mx = t > mx ? t : mx;
            if (mx <= cap) break;
            cap = mx;
          }
        }
        job->cap = cap;
        if (failed || !to_write.push(std::move(job))) break;
      }
      to_write.close();
      to_gpu.close();  // unblocks the parser if the GPU stage stopped early
    });

    size_t n_total = 0, n_nonempty = 0;
    std::thread writer([&] {
      JobPtr job;
      while (to_write.pop(job)) {
        const size_t n = job->lines.size();
        std::vector<std::string> text(helpers);
        std::vector<size_t> nonempty(helpers, 0);
        parallel_for(helpers, n, [&](unsigned t, size_t lo, size_t hi) {
          std::string& buf = text[t];
          char num[64];
          for (size_t q = lo; q < hi; q++) {
            for (uint32_t k = 0; k < job->cnt[q]; k++) {
              const fmx_match& m = job->out[q * job->cap + k];
              snprintf(num, sizeof(num), "%.9g", m.score);  // boost::lexical_cast<std::string>(float), cli:230
              if (k) buf += '\t';
              buf += num;
              buf += '\t';
              if (subseq) {  // score, max_subseq, id, and the sub-n-gram itself appended to the id (cli:241-247, src/fuzzy_match.cc:348)
                buf += std::to_string(m.max_subseq);
                buf += '\t';
              }
              buf += fmx_index_sentence_id(ix, m.s_id);
              if (subseq) {
                buf += '\t';
                if (!job->text.empty() && job->text[q]) buf += fmx_text_detokenize(job->text[q], job->pos[q], job->pos[q] + (uint32_t)m.max_subseq);
                else {  // tokenised input: the tokens of the sub-n-gram (terminated in place by the parser)
                  const char* w = job->lines[q].data();
                  const char* end = w + job->lines[q].size();
                  uint32_t idx = 0;
                  bool first = true;
                  while (w < end) {
                    if (*w == '\0') { w++; continue; }
                    const size_t len = strlen(w);
                    if (idx >= job->pos[q] && idx < job->pos[q] + (uint32_t)m.max_subseq) { if (!first) buf += ' '; buf.append(w, len); first = false; }
                    idx++;
                    w += len;
                  }
                }
              }
            }
            buf += '\n';
            if (job->cnt[q]) nonempty[t]++;
          }
        });
        for (unsigned t = 0; t < helpers; t++) {
          fwrite(text[t].data(), 1, text[t].size(), stdout);
          n_nonempty += nonempty[t];
        }
        n_total += n;
      }
      fflush(stdout);
    });

    {  // reader: this thread
      std::string line;
      bool eof = false;
      while (!eof && !failed) {
        JobPtr job(new Job);
        job->lines.reserve(batch);
        while (job->lines.size() < batch && std::getline(std::cin, line)) job->lines.push_back(line);
        eof = job->lines.size() < batch;
        if (job->lines.empty()) break;
        if (!to_parse.push(std::move(job))) break;
      }
      to_parse.close();
    }
    parser.join();
    gpu.join();
    writer.join();
    if (failed) {
      std::cerr << "ERROR: " << fail_msg << std::endl;
      return 1;
    }
    std::cerr << "NMATCH\t" << n_nonempty << "\t/\t" << n_total << std::endl;
  } else if (action != "index") {
    std::cerr << "ERROR: unknown action '" << action << "' (index|match|subseq)" << std::endl;
    return 2;
  }
  tick("Done");
  fmx_index_free(ix);
  return 0;
}
