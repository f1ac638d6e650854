// fmx_cli.cc — CLI driver with the reference's flags, output format and timing lines
// (cli/src/FuzzyMatch-cli.cc:287-343 options, :219-233 output, :21-27 TICK, :450-467 NMATCH), on top of
// the C ABI.  Differences: input lines must already be tokenised (tokens separated by blanks: the
// OpenNMT tokenizer front-end is out of scope), and lines are matched in batches on the
// GPU instead of one per worker thread; -N sets the helper threads that parse and format around it.
// Matching is a streaming pipeline (reader -> parser -> GPU -> writer).  Output order = input order.
//
//   fmx-cli -i tm.fmi -a match -f 0.7 -n 5 < patterns.tok > matches.tsv
//   fmx-cli -c corpus.tok -a index            # builds corpus.tok.fmi (ids = line numbers, or "id<TAB>tokens")
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

static int usage(int rc) {
  std::cerr << "FuzzyMatch cli options (MI355X build; input must be pre-tokenised, blank-separated):\n"
               "  -h [ --help ]\n  -a [ --action ] arg (=index)   index|match\n  -i [ --index ] arg             index file (.fmi)\n"
               "  -c [ --corpus ] arg            tokenised corpus to index: one sentence per line, optional 'id<TAB>' prefix\n"
               "  -f [ --fuzzy ] arg (=0.8)\n  --ml arg (=3)\n  --mr arg (=0.3)\n  -n [ --nmatch ] arg (=5)\n  -P [ --no-perfect ]\n"
               "  -I [ --idf-penalty ] arg (=0)\n  --insert-cost arg (=1)  --delete-cost arg (=1)  --replace-cost arg (=1)\n"
               "  --max-tokens-in-pattern arg (=300)\n  --contrast arg (=0)  --contrast-reduce arg (=mean)  --contrast-buffer arg (=-1)\n"
               "  -N [ --nthreads ] arg          helper threads for parsing / formatting (matching runs on the GPU)\n"
               "  --batch arg (=65536)           patterns per GPU batch\n  --device arg (=0)\n";
  return rc;
}

int main(int argc, char** argv) {
  g_start = g_period = std::chrono::steady_clock::now();
  std::ios::sync_with_stdio(false);  // std::getline on std::cin is several times faster without the stdio lock-step
  std::cin.tie(nullptr);
  std::string action = "index", index_file, corpus, reduce = "mean";
  fmx_params prm = {0.8f, 5, 0, 3, 0.3f, 0.f, 1.f, 1.f, 1.f, 0.f, 0, -1};
  uint64_t max_tokens = 300;
  size_t batch = 65536;
  unsigned nthreads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
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
    else if (a == "-N" || a == "--nthreads") nthreads = (unsigned)std::max(1, atoi(val()));
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
      const int64_t sid = fmx_index_add_tm(ix, id.c_str(), ptr.data(), (uint32_t)ptr.size(), nullptr, 0, nullptr, nullptr);
      if (sid < 0 && sid != FMX_TM_SKIPPED) {
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
    // Streaming pipeline (the reference keeps 1000 futures in flight over a worker pool, cli:112-193): batches
    // of lines flow reader -> parser (tokens + vocabulary ids, `nthreads` helpers) -> GPU -> formatter/writer
    // through bounded queues, so reading and parsing batch i + 1 and printing batch i - 1 overlap the GPU work
    // of batch i.  Output order = input order.
    struct Job {
      std::vector<std::string> lines;
      std::vector<uint32_t> q_off, wids, cnt, tot;
      std::vector<fmx_match> out;
      uint32_t cap = 0;
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
        std::vector<std::vector<uint32_t>> ids(helpers), counts(helpers);
        parallel_for(helpers, n, [&](unsigned t, size_t lo, size_t hi) {
          std::vector<const char*> ptr;
          for (size_t i = lo; i < hi; i++) {
            std::string& l = job->lines[i];
            ptr.clear();
            // tokens in place: blanks become terminators
            size_t k = 0, len = l.size();
            while (k < len) {
              while (k < len && (l[k] == ' ' || l[k] == '\t' || l[k] == '\r')) l[k++] = '\0';
              if (k >= len) break;
              ptr.push_back(l.data() + k);
              while (k < len && l[k] != ' ' && l[k] != '\t' && l[k] != '\r') k++;
            }
            const size_t base = ids[t].size();
            ids[t].resize(base + ptr.size());
            if (!ptr.empty() && fmx_vocab_lookup(ix, ptr.data(), ptr.size(), ids[t].data() + base) != FMX_OK) fail(fmx_last_error());
            counts[t].push_back((uint32_t)ptr.size());
          }
        });
        job->q_off.assign(1, 0);
        job->q_off.reserve(n + 1);
        size_t total = 0;
        for (unsigned t = 0; t < helpers; t++) total += ids[t].size();
        job->wids.reserve(total);
        for (unsigned t = 0; t < helpers; t++) {
          for (uint32_t c : counts[t]) job->q_off.push_back(job->q_off.back() + c);
          job->wids.insert(job->wids.end(), ids[t].begin(), ids[t].end());
        }
        if (!to_gpu.push(std::move(job))) break;
      }
      to_gpu.close();
      to_parse.close();  // unblocks the reader if this stage stopped early
    });

    std::thread gpu([&] {
      JobPtr job;
      uint32_t cap = prm.nmatch ? prm.nmatch : 64;
      while (to_gpu.pop(job)) {
        const size_t n = job->lines.size();
        fmx_batch b = {n, job->q_off.data(), job->wids.data(), nullptr, nullptr, 0};
        job->cnt.resize(n);
        job->tot.resize(n);
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
              buf += fmx_index_sentence_id(ix, m.s_id);
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
    std::cerr << "ERROR: action '" << action << "' is not available in this build (match and index only)" << std::endl;
    return 2;
  }
  tick("Done");
  fmx_index_free(ix);
  return 0;
}
