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
#include <array>
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
    // raw text: lines are handed to the library a block at a time (fmx_index_add_text_bulk: _tokenize_and_normalize on every usable
    // CPU, sentences added in order) instead of one call per line on this thread
    std::vector<std::string> b_ids, b_lines;
    auto flush_block = [&]() -> bool {
      if (b_lines.empty()) return true;
      std::vector<const char*> pi(b_lines.size()), pl(b_lines.size());
      std::vector<size_t> ll(b_lines.size());
      std::vector<int64_t> sids(b_lines.size());
      for (size_t k = 0; k < b_lines.size(); k++) { pi[k] = b_ids[k].c_str(); pl[k] = b_lines[k].data(); ll[k] = b_lines[k].size(); }
      if (fmx_index_add_text_bulk(ix, pi.data(), pl.data(), ll.data(), b_lines.size(), sids.data()) < 0) {
        std::cerr << "ERROR: " << fmx_last_error() << std::endl;
        return false;
      }
      for (size_t k = 0; k < b_lines.size(); k++)
        if (sids[k] == FMX_TM_SKIPPED) std::cerr << "WARNING: cannot index empty segment: " << b_lines[k] << " (" << b_ids[k] << ")" << std::endl;
      b_ids.clear(); b_lines.clear();
      return true;
    };
    while (in.getline(line)) {
      ++count;
      std::string id = std::to_string(count);
      int64_t sid = 0;
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
        b_ids.push_back(std::move(id));
        b_lines.push_back(line);
        if (b_lines.size() >= (1u << 18) && !flush_block()) return 1;
      }
      if (sid < 0 && sid != FMX_TM_SKIPPED) {
        std::cerr << "ERROR: " << fmx_last_error() << std::endl;
        return 1;
      }
    }
    if (!flush_block()) return 1;
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

  std::shared_ptr<void> after_done;  // what is released after the last STEP line (the blocks' pinned buffers: set up before the clock started, too)
  bool done_ticked = false;
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
    // Before the clock of the matching phase starts (like the index load): the blocks' pinned buffers and the library's call
    // contexts (streams, events) — one tiny call per slot.
    const unsigned workers = std::max(1u, nthreads);
    constexpr int kGpuCalls = 4;     // synchronous library calls in flight (each in its own context; their kernels and copies overlap)
    // -a match: the block goes to the library as it sits in the file — pre-tokenised lines (fmx_match_batch_lines) or, like the
    // reference CLI reads them, RAW lines (fmx_match_batch_raw_lines: the device tokenises what its byte-class machine covers, the
    // host front-end the rest); the host-side tokenisation below is the subsequence action's
    const bool lines_path = !subseq;
    const uint32_t cap0 = subseq ? 1u : (prm.nmatch ? prm.nmatch : 64);

    const size_t n_jobs = std::min<size_t>(workers, 16) + 8;
    {
      // one call per slot, at once, with a block's worth of the memory's OWN sentences as patterns: every context of the library
      // creates its streams and sizes its workspace (segments, queues, work lists: learned from what a pass needed) on realistic
      // hit and candidate counts here, instead of re-running and re-allocating inside the first real calls of a short run
      const void* buf_v = nullptr; const void* pos_v = nullptr;
      uint64_t n_buf = 0, n_pos = 0;
      (void)fmx_index_array(ix, FMX_A_BUFFER, &buf_v, &n_buf);
      (void)fmx_index_array(ix, FMX_A_SENT_POS, &pos_v, &n_pos);
      const uint32_t* sbuf = (const uint32_t*)buf_v;
      const uint32_t* spos = (const uint32_t*)pos_v;
      const size_t nw = std::min<size_t>(batch, 1u << 20);
      std::vector<uint32_t> off(nw + 1, 0), w1;
      if (n_pos) {
        w1.reserve(nw * 24);
        for (size_t q = 0; q < nw; q++) {
          const uint32_t sp = spos[(q * 7919u) % n_pos], len = sbuf[sp];   // spread over the memory
          w1.insert(w1.end(), sbuf + sp + 1, sbuf + sp + 1 + len);
          off[q + 1] = (uint32_t)w1.size();
        }
      }
      if (w1.empty()) w1.push_back(1u);
      std::vector<std::thread> warm;
      for (int k = 0; k < kGpuCalls; k++)
        warm.emplace_back([&] {
          std::vector<fmx_match> out(nw * cap0);
          std::vector<uint32_t> c1(nw), t1(nw);
          fmx_batch b1 = {nw, off.data(), w1.data(), nullptr, nullptr, 0};
          (void)fmx_match_batch(ix, &prm, &b1, out.data(), c1.data(), t1.data(), cap0, nullptr);
          if (lines_path) {  // ... and the line path's own buffers and kernels: a block of the same shape, of words no memory holds
            std::string dummy;
            dummy.reserve(nw * 120);
            for (size_t q = 0; q < nw; q++) { for (int k = 0; k < 24; k++) dummy += tokenised ? "?w?w " : "qzqz "; dummy.back() = '\n'; }
            if (tokenised) (void)fmx_match_batch_lines(ix, &prm, dummy.data(), dummy.size(), nw, out.data(), c1.data(), t1.data(), cap0, nullptr);
            else (void)fmx_match_batch_raw_lines(ix, &prm, dummy.data(), dummy.size(), nw, out.data(), c1.data(), t1.data(), cap0, nullptr);
          }
        });
      for (auto& t : warm) t.join();
    }
    // Streaming pipeline (the reference keeps 1000 futures in flight over a worker pool, cli:112-193): the reader cuts stdin into
    // blocks of whole lines (no per-line strings), `nthreads` workers take a block each — at most kGpuCalls of them inside the
    // library at once —, format its records and hand it to the writer, which prints the blocks in input order.
    //   * --tokenised -a match: the block goes to the GPU AS IT IS (fmx_match_batch_lines): the device cuts it into patterns and
    //     tokens and looks the ids up; the host never touches a token.
    //   * raw text: the library's text front-end tokenises each line (host), the tokens travel as spellings
    //     (fmx_match_batch_text: normalised + real tokens interned on the device, penalty tokens as pairs).
    //   * -a subseq: vocabulary ids on the host (fmx_subsequence_batch takes ids).
    struct Job {
      uint64_t seq = 0;
      char*    text = nullptr;      // pinned: the block, every line ended by '\n'
      size_t   text_cap = 0, n_bytes = 0, n_lines = 0;
      fmx_match* out = nullptr;     // pinned: n_lines x cap records
      uint32_t *cnt = nullptr, *tot = nullptr;
      size_t   out_cap = 0, cnt_cap = 0;
      std::string printed;          // the block's output as one string (subseq), or ...
      std::vector<std::string> parts;               // ... in ranges of kFmtLines lines, formatted by whichever workers are free
      std::vector<size_t> part_nonempty;
      std::atomic<uint32_t> parts_left{0};
      uint32_t cap_used = 0;
      size_t   nonempty = 0;
      void need_text(size_t n) {
        if (n <= text_cap) return;
        char* p = (char*)fmx_host_alloc(n + n / 2);
        if (!p) { fprintf(stderr, "ERROR: out of pinned memory\n"); exit(1); }
        if (n_bytes) memcpy(p, text, n_bytes);
        if (text) fmx_host_free(text);
        text = p; text_cap = n + n / 2;
      }
      void need_out(size_t lines, uint32_t cap) {
        if (lines * cap > out_cap) { if (out) fmx_host_free(out); out_cap = lines * cap + lines * cap / 4; out = (fmx_match*)fmx_host_alloc(out_cap * sizeof(fmx_match)); }
        if (lines > cnt_cap) {
          if (cnt) fmx_host_free(cnt);
          if (tot) fmx_host_free(tot);
          cnt_cap = lines + lines / 4;
          cnt = (uint32_t*)fmx_host_alloc(cnt_cap * 4); tot = (uint32_t*)fmx_host_alloc(cnt_cap * 4);
        }
        if (!out || !cnt || !tot) { fprintf(stderr, "ERROR: out of pinned memory\n"); exit(1); }
      }
      ~Job() { if (text) fmx_host_free(text); if (out) fmx_host_free(out); if (cnt) fmx_host_free(cnt); if (tot) fmx_host_free(tot); }
    };
    std::vector<std::unique_ptr<Job>> pool(n_jobs);
    constexpr size_t kFmtLines = 8192;   // lines per formatting task
    struct Task { Job* job; int part; };  // part < 0: the block's GPU call; else: format lines [part * kFmtLines, ...)
    Channel<Job*> free_jobs(n_jobs + 1), to_write(n_jobs + 1);
    Channel<Task> to_work((n_jobs + 1) * (batch / kFmtLines + 2));  // never full: a worker that pushes format tasks must not block
    for (auto& j : pool) {
      j.reset(new Job);
      j->need_text(std::min<size_t>(batch * 160, 48u << 20));   // (grown on demand: a block is `batch` lines, whatever their length)
      j->need_out(batch, cap0);
      free_jobs.push(j.get());
    }
    // the ids of the memory's sentences in sixteen bytes each — up to fifteen characters in place, longer ones by pointer —: the
    // formatters append an id with one cache miss, without a call and a strlen per match
    struct SidEntry { char s[15]; uint8_t len; };   // len == 255: s holds a pointer and a 32-bit length
    std::vector<SidEntry> sid_tab(info.n_sentences);
    parallel_for(std::min(workers, 8u), sid_tab.size(), [&](unsigned, size_t lo, size_t hi) {
      for (size_t k = lo; k < hi; k++) {
        const char* c = fmx_index_sentence_id(ix, (uint32_t)k);
        const size_t n = strlen(c);
        SidEntry& e = sid_tab[k];
        if (n <= 15) { memcpy(e.s, c, n); memset(e.s + n, 0, 15 - n); e.len = (uint8_t)n; }
        else { const uint32_t n32 = (uint32_t)n; memcpy(e.s, &c, sizeof c); memcpy(e.s + 8, &n32, 4); e.len = 255; }
      }
    });
    auto sid_len = [&](uint32_t sid) -> uint32_t {
      if (sid >= sid_tab.size()) return 0u;
      const SidEntry& e = sid_tab[sid];
      if (e.len != 255) return e.len;
      uint32_t n32; memcpy(&n32, e.s + 8, 4); return n32;
    };
    // boost::lexical_cast<std::string>(float) of a score (cli:230) = "%.9g"; scores are k / 10000 (src/fuzzy_match.cc:598): tabulated
    std::vector<float> score_f(10001);
    std::vector<std::array<char, 16>> score_s(10001);
    std::vector<uint8_t> score_n(10001);
    for (int k = 0; k <= 10000; k++) {
      score_f[k] = (float)((double)k / 10000.0);
      score_n[k] = (uint8_t)snprintf(score_s[k].data(), 16, "%.9g", score_f[k]);
    }
    const bool timing = getenv("FMX_CLI_TIMING") != nullptr;
    std::atomic<uint64_t> us_gpu{0}, us_fmt{0}, us_read{0}, us_write{0}, us_wait_job{0}, n_retries{0}, us_gpu_max{0};
    std::atomic<uint64_t> t_gpu_first_end{0}, t_gpu_last_end{0}, t_read_end{0}, t_fmt_last_end{0};
    auto now_us = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    uint64_t t_tick = 0;   // the clock of the matching phase starts when the pipeline stands (below): workers and writer wait for input
    std::atomic<bool> failed{false};
    std::string fail_msg;
    std::mutex fail_mu;
    auto fail = [&](const std::string& msg) {
      std::lock_guard<std::mutex> lk(fail_mu);
      if (!failed.exchange(true)) fail_msg = msg;
    };
    std::mutex gpu_mu;
    std::condition_variable gpu_cv;
    int gpu_busy = 0;
    struct GpuSlot {
      std::mutex& mu; std::condition_variable& cv; int& busy;
      GpuSlot(std::mutex& m, std::condition_variable& c, int& b) : mu(m), cv(c), busy(b) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return busy < kGpuCalls; }); busy++; }
      ~GpuSlot() { { std::lock_guard<std::mutex> lk(mu); busy--; } cv.notify_one(); }
    };
    auto put_score = [&](std::string& buf, float v) {
      const int k = (int)(v * 10000.f + 0.5f);
      if (k >= 0 && k <= 10000 && score_f[k] == v) { buf.append(score_s[k].data(), score_n[k]); return; }
      char num[64];
      buf.append(num, (size_t)snprintf(num, sizeof(num), "%.9g", v));
    };
    std::atomic<uint32_t> cap_now{cap0};

    auto work = [&](Job* job) {
      const size_t n = job->n_lines;
      uint32_t cap = cap_now.load();
      std::vector<const char*> line_at;           // generic path: start of every line (lines are terminated in place)
      std::vector<uint32_t> q_off, wids, pos;
      std::vector<fmx_text*> texts;
      struct FreeTexts { std::vector<fmx_text*>& v; ~FreeTexts() { for (fmx_text* t : v) fmx_text_free(t); } } free_texts{texts};
      if (lines_path) {
        for (;;) {  // nmatch == 0 (no limit): grow the per-query capacity until nothing is truncated
          job->need_out(n, cap);
          int rc;
          { GpuSlot slot(gpu_mu, gpu_cv, gpu_busy); const uint64_t a = now_us(); fmx_stats fs; memset(&fs, 0, sizeof fs); rc = tokenised ? fmx_match_batch_lines(ix, &prm, job->text, job->n_bytes, n, job->out, job->cnt, job->tot, cap, timing ? &fs : nullptr) : fmx_match_batch_raw_lines(ix, &prm, job->text, job->n_bytes, n, job->out, job->cnt, job->tot, cap, timing ? &fs : nullptr); const uint64_t d = now_us() - a; us_gpu += d; n_retries += fs.n_retries; { uint64_t z = 0; t_gpu_first_end.compare_exchange_strong(z, a + d - t_tick); t_gpu_last_end.store(a + d - t_tick); } uint64_t mx = us_gpu_max.load(); while (d > mx && !us_gpu_max.compare_exchange_weak(mx, d)) {} }
          if (rc != FMX_OK) { fail(fmx_last_error()); return; }
          uint32_t mx = 0;
          for (size_t q = 0; q < n; q++) mx = job->tot[q] > mx ? job->tot[q] : mx;
          if (mx <= cap) break;
          cap = mx;
        }
      } else {
        // lines -> tokens on the host
        line_at.reserve(n);
        char* p = job->text;
        char* const end = job->text + job->n_bytes;
        while (p < end) { line_at.push_back(p); char* nl = (char*)memchr(p, '\n', (size_t)(end - p)); *nl = '\0'; p = nl + 1; }
        q_off.assign(1, 0);
        std::string norm_text, real_text;
        std::vector<uint64_t> itok_slot;
        std::vector<uint32_t> itok_id;
        std::vector<const char*> ptr;
        if (!tokenised && subseq) texts.assign(n, nullptr);
        for (size_t i = 0; i < n; i++) {
          char* l = const_cast<char*>(line_at[i]);
          const size_t len = strlen(l);
          ptr.clear();
          fmx_text* th = nullptr;
          if (tokenised) {  // tokens in place: blanks become terminators
            size_t k = 0;
            while (k < len) {
              while (k < len && (l[k] == ' ' || l[k] == '\t' || l[k] == '\r')) l[k++] = '\0';
              if (k >= len) break;
              ptr.push_back(l + k);
              while (k < len && l[k] != ' ' && l[k] != '\t' && l[k] != '\r') k++;
            }
          } else {
            th = fmx_text_tokenize(ix_pt, l, len);
            if (!th) { fail("out of memory"); return; }
            const uint32_t nt = fmx_text_num_tokens(th);
            for (uint32_t j = 0; j < nt; j++) ptr.push_back(fmx_text_norm(th, j));
          }
          if (subseq) {  // fmx_subsequence_batch takes vocabulary ids
            const size_t base = wids.size();
            wids.resize(base + ptr.size());
            if (!ptr.empty() && fmx_vocab_lookup_mt(ix, ptr.data(), ptr.size(), wids.data() + base, 1) != FMX_OK) { fail(fmx_last_error()); return; }
          } else {       // the tokens as spellings: ids (and real tokens) are interned on the device
            bool empty_real = false;
            for (uint32_t j = 0; j < ptr.size(); j++) {
              norm_text.append(ptr[j]); norm_text.push_back('\0');
              if (th && info.real_differs) { const char* r = fmx_text_real(th, j); empty_real = empty_real || !*r; real_text.append(r); real_text.push_back('\0'); }
            }
            if (empty_real) { fail("a pattern with an empty real token cannot travel as a text batch"); return; }
            if (th)
              for (uint32_t k = 0; k < fmx_text_num_itoks(th); k++) {
                const char* is = fmx_text_itok(th, k);
                const uint64_t at = fmx_text_itok_index(th, k);
                if (at <= ptr.size()) { itok_slot.push_back((uint64_t)q_off.back() + i + at); itok_id.push_back(fmx_index_intern_itok(ix, is, strlen(is))); }
              }
          }
          q_off.push_back(q_off.back() + (uint32_t)ptr.size());
          if (th) { if (subseq) texts[i] = th; else fmx_text_free(th); }
        }
        if (subseq) {
          job->need_out(n, 1);
          pos.resize(n);
          fmx_batch b = {n, q_off.data(), wids.data(), nullptr, nullptr, 0};
          GpuSlot slot(gpu_mu, gpu_cv, gpu_busy);
          if (fmx_subsequence_batch(ix, &b, prm.nmatch, prm.no_perfect, prm.min_subseq_length, prm.min_subseq_ratio, idf_weighting ? 1 : 0, job->out,
                                    job->cnt, pos.data()) != FMX_OK) { fail(fmx_last_error()); return; }
          cap = 1;
        } else {
          fmx_text_batch tb = {n, q_off.data(), norm_text.data(), norm_text.size(), nullptr, 0, 0, nullptr, nullptr};
          if (!tokenised && info.real_differs) {
            tb.real_text = real_text.data(); tb.n_real_bytes = real_text.size();
            tb.n_itoks = itok_slot.size(); tb.itok_slot = itok_slot.data(); tb.itok_id = itok_id.data();
          }
          for (;;) {
            job->need_out(n, cap);
            int rc;
            { GpuSlot slot(gpu_mu, gpu_cv, gpu_busy); rc = fmx_match_batch_text(ix, &prm, &tb, job->out, job->cnt, job->tot, cap, nullptr); }
            if (rc != FMX_OK) { fail(fmx_last_error()); return; }
            uint32_t mx = 0;
            for (size_t q = 0; q < n; q++) mx = job->tot[q] > mx ? job->tot[q] : mx;
            if (mx <= cap) break;
            cap = mx;
          }
        }
      }
      if (cap > cap_now.load()) cap_now.store(cap);
      job->cap_used = cap;
      job->parts.clear();
      if (!subseq) return;  // match: the records are formatted in ranges of kFmtLines lines by whichever workers are free (format_part)
      // records -> text (cli:219-248): score <TAB> id pairs joined by tabs, an empty line when nothing matched
      const uint64_t t_fmt = now_us();
      std::string& buf = job->printed;
      buf.clear();
      buf.reserve(n * 48);
      job->nonempty = 0;
      for (size_t q = 0; q < n; q++) {
        for (uint32_t k = 0; k < job->cnt[q]; k++) {
          const fmx_match& mt = job->out[q * cap + k];
          if (k) buf += '\t';
          put_score(buf, mt.score);
          buf += '\t';
          if (subseq) {  // score, max_subseq, id, and the sub-n-gram itself appended to the id (cli:241-247, src/fuzzy_match.cc:348)
            buf += std::to_string(mt.max_subseq);
            buf += '\t';
          }
          buf += fmx_index_sentence_id(ix, mt.s_id);
          if (subseq) {
            buf += '\t';
            if (!texts.empty() && texts[q]) buf += fmx_text_detokenize(texts[q], pos[q], pos[q] + (uint32_t)mt.max_subseq);
            else {  // tokenised input: the tokens of the sub-n-gram (terminated in place above)
              const char* w = line_at[q];
              const char* end = q + 1 < n ? line_at[q + 1] - 1 : job->text + job->n_bytes - 1;
              uint32_t idx = 0;
              bool first = true;
              while (w < end) {
                if (*w == '\0') { w++; continue; }
                const size_t len = strlen(w);
                if (idx >= pos[q] && idx < pos[q] + (uint32_t)mt.max_subseq) { if (!first) buf += ' '; buf.append(w, len); first = false; }
                idx++;
                w += len;
              }
            }
          }
        }
        buf += '\n';
        if (job->cnt[q]) job->nonempty++;
      }
      us_fmt += now_us() - t_fmt;
      t_fmt_last_end.store(now_us() - t_tick);
    };

    // -a match: lines [part * kFmtLines, ...) of a block -> "score <TAB> id" pairs joined by tabs, an empty line when nothing matched
    // (cli:219-248); written through a raw pointer (a line is at most cap x (15 + 1 + id + 1) bytes + the line feed)
    auto format_part = [&](Job* job, uint32_t part) {
      const uint64_t t_fmt = now_us();
      const size_t lo = (size_t)part * kFmtLines, hi = std::min(job->n_lines, lo + kFmtLines);
      const uint32_t cap = job->cap_used;
      std::string& buf = job->parts[part];
      // upper bound of the part's text: a match is a tab, a score of at most 16 characters, a tab and its id (34 covers every id
      // held in place; the rare longer ones are added)
      size_t need = (hi - lo) + 16, n_match = 0;
      for (size_t q = lo; q < hi; q++) n_match += job->cnt[q];
      need += n_match * 34;
      if (info.n_sentences && !sid_tab.empty())
        for (size_t q = lo; q < hi; q++)
          for (uint32_t k = 0; k < job->cnt[q]; k++) {
            const uint32_t sid = job->out[q * cap + k].s_id;
            if (sid < sid_tab.size() && sid_tab[sid].len == 255) need += sid_len(sid);
          }
      buf.resize(need);
      char* const b0 = &buf[0];
      char* w = b0;
      size_t nonempty = 0;
      constexpr size_t kAhead = 6;   // id entries of the lines ahead are on their way while this line is written
      for (size_t q = lo; q < hi; q++) {
        if (q + kAhead < hi)
          for (uint32_t k = 0; k < job->cnt[q + kAhead]; k++) {
            const uint32_t sid = job->out[(q + kAhead) * cap + k].s_id;
            if (sid < sid_tab.size()) __builtin_prefetch(&sid_tab[sid]);
          }
        const uint32_t c = job->cnt[q];
        const fmx_match* mt = job->out + q * cap;
        for (uint32_t k = 0; k < c; k++) {
          if (k) *w++ = '\t';
          const float v = mt[k].score;
          const int ks = (int)(v * 10000.f + 0.5f);
          if (ks >= 0 && ks <= 10000 && score_f[ks] == v) { memcpy(w, score_s[ks].data(), 16); w += score_n[ks]; }
          else w += snprintf(w, 17, "%.9g", v);
          *w++ = '\t';
          const uint32_t sid = mt[k].s_id;
          if (sid < sid_tab.size()) {
            const SidEntry& e = sid_tab[sid];
            if (e.len != 255) { memcpy(w, e.s, 15); w += e.len; }
            else { const char* c2; memcpy(&c2, e.s, sizeof c2); const uint32_t n32 = sid_len(sid); memcpy(w, c2, n32); w += n32; }
          }
        }
        *w++ = '\n';
        nonempty += c ? 1u : 0u;
      }
      buf.resize((size_t)(w - b0));
      job->part_nonempty[part] = nonempty;
      us_fmt += now_us() - t_fmt;
      t_fmt_last_end.store(now_us() - t_tick);
    };
    // blocks in flight; the task queue closes when the reader is through and the last block has been handed to the writer
    std::atomic<uint64_t> jobs_open{0};
    std::atomic<bool> reader_done{false};
    auto job_done = [&](Job* job) {
      const bool ok = to_write.push(job);
      if ((jobs_open.fetch_sub(1) == 1 && reader_done.load()) || !ok) to_work.close();
    };
    std::vector<std::thread> work_threads;
    for (unsigned t = 0; t < workers; t++)
      work_threads.emplace_back([&] {
        Task task;
        while (to_work.pop(task)) {
          Job* job = task.job;
          if (task.part >= 0) {
            if (!failed) format_part(job, (uint32_t)task.part);
            if (job->parts_left.fetch_sub(1) == 1) job_done(job);
            continue;
          }
          if (!failed) work(job);
          if (subseq || failed) { job->parts.clear(); job_done(job); continue; }
          const uint32_t n_parts = (uint32_t)((job->n_lines + kFmtLines - 1) / kFmtLines);
          job->parts.assign(n_parts, std::string());
          job->part_nonempty.assign(n_parts, 0);
          job->parts_left.store(n_parts);
          for (uint32_t k = 1; k < n_parts; k++) to_work.push(Task{job, (int)k});
          format_part(job, 0);
          if (job->parts_left.fetch_sub(1) == 1) job_done(job);
        }
      });

    size_t n_total = 0, n_nonempty = 0;
    std::atomic<uint64_t> jobs_made{~0ull};   // set by the reader at end of input
    std::thread writer([&] {
      std::vector<Job*> waiting;
      uint64_t next = 0;
      Job* job;
      for (;;) {
        if (next == jobs_made.load()) break;
        if (!to_write.pop(job)) break;
        waiting.push_back(job);
        for (bool again = true; again;) {
          again = false;
          for (size_t k = 0; k < waiting.size(); k++)
            if (waiting[k]->seq == next) {
              Job* j = waiting[k];
              waiting.erase(waiting.begin() + (long)k);
              const uint64_t a = now_us();
              if (!failed) {
                if (j->parts.empty()) fwrite(j->printed.data(), 1, j->printed.size(), stdout);
                else for (size_t k = 0; k < j->parts.size(); k++) { fwrite(j->parts[k].data(), 1, j->parts[k].size(), stdout); j->nonempty += j->part_nonempty[k]; }
              }
              us_write += now_us() - a;
              n_total += j->n_lines; n_nonempty += j->nonempty;
              j->nonempty = 0;
              j->n_bytes = j->n_lines = 0;
              free_jobs.push(j);
              next++;
              again = true;
              break;
            }
        }
      }
      fflush(stdout);
    });

    const size_t chunk = 1u << 20;
    std::vector<char> stage(8 * chunk);   // (allocated and touched before the clock starts, like the blocks' buffers)
    tick(subseq ? "Subsequencing" : "Matching");
    t_tick = now_us();
    {  // reader: this thread.  stdin in 1 MB reads into one staging buffer; blocks of whole lines are copied out of it into the
       // pinned buffer of a free job.  The first blocks are small (8192 lines, then twice as many, up to `batch`): a short input
       // is through its first GPU call while the rest is still being read, a long one runs on full blocks after four.
      size_t head = 0, tail = 0, scanned = 0;
      bool eof = false;
      uint64_t seq = 0;
      while (!failed) {
        const size_t want = std::min<size_t>(batch, (size_t)8192 << std::min<uint64_t>(seq, 16));
        const uint64_t t_r = now_us();
        size_t lines = 0, cut = head;
        for (;;) {
          while (scanned < tail && lines < want) {
            const char* nl = (const char*)memchr(stage.data() + scanned, '\n', tail - scanned);
            if (!nl) { scanned = tail; break; }
            scanned = (size_t)(nl - stage.data()) + 1;
            cut = scanned;
            lines++;
          }
          if (lines >= want || eof) break;
          if (tail - head >= (64u << 20) && lines > 0) break;   // (a block is at most ~64 MB)
          if (stage.size() - tail < chunk) {                    // make room: move what is left to the front, grow if that is not enough
            if (head) { memmove(stage.data(), stage.data() + head, tail - head); scanned -= head; cut -= head; tail -= head; head = 0; }
            if (stage.size() - tail < chunk) stage.resize(stage.size() * 2);
          }
          const size_t got = fread(stage.data() + tail, 1, chunk, stdin);
          if (got == 0) eof = true;
          tail += got;
        }
        if (eof && cut < tail) {  // a last line without a line feed
          if (tail == stage.size()) stage.resize(stage.size() + 1);
          stage[tail++] = '\n';
          cut = scanned = tail;
          lines++;
        }
        us_read += now_us() - t_r;
        if (lines == 0) break;  // end of input
        Job* job;
        const uint64_t t_w = now_us();
        if (!free_jobs.pop(job)) break;
        const uint64_t t_c = now_us();
        us_wait_job += t_c - t_w;
        job->n_bytes = 0;
        job->need_text(cut - head);
        memcpy(job->text, stage.data() + head, cut - head);
        job->n_bytes = cut - head;
        job->n_lines = lines;
        job->seq = seq++;
        head = cut;
        scanned = std::max(scanned, head);
        us_read += now_us() - t_c;
        jobs_open.fetch_add(1);
        if (!to_work.push(Task{job, -1})) break;
        if (eof && head == tail) break;
      }
      jobs_made.store(seq);
      t_read_end.store(now_us() - t_tick);
      reader_done.store(true);
      if (jobs_open.load() == 0) to_work.close();
    }
    writer.join();      // the writer leaves once it has printed `jobs_made` blocks: the output is complete
    const uint64_t t_end = now_us() - t_tick;
    auto wind_down = [&] {  // after the "Done" line: sixteen thread exits are a millisecond
      to_work.close();
      for (auto& t : work_threads) t.join();
      to_write.close();
      free_jobs.close();
      after_done = std::make_shared<std::vector<std::unique_ptr<Job>>>(std::move(pool));
    };
    if (timing)
      std::cerr << "TIMING\tread_ms\t" << us_read / 1e3 << "\twait_free_block_ms\t" << us_wait_job / 1e3 << "\tgpu_calls_ms(sum)\t" << us_gpu / 1e3
                << "\tformat_ms(sum)\t" << us_fmt / 1e3 << "\twrite_ms\t" << us_write / 1e3 << "\tat_ms:first_gpu_done/read_done/last_gpu_done/last_format_done/end\t" << t_gpu_first_end / 1e3 << "/" << t_read_end / 1e3 << "/" << t_gpu_last_end / 1e3 << "/"
                << t_fmt_last_end / 1e3 << "/" << t_end / 1e3 << "\tgpu_call_max_ms\t" << us_gpu_max / 1e3 << "\tretries\t" << n_retries << "\tworkers\t" << workers << "\tblocks\t" << n_jobs << std::endl;
    if (failed) {
      wind_down();
      std::cerr << "ERROR: " << fail_msg << std::endl;
      return 1;
    }
    std::cerr << "NMATCH\t" << n_nonempty << "\t/\t" << n_total << std::endl;
    tick("Done");
    done_ticked = true;
    wind_down();
  } else if (action != "index") {
    std::cerr << "ERROR: unknown action '" << action << "' (index|match|subseq)" << std::endl;
    return 2;
  }
  if (!done_ticked) tick("Done");
  fmx_index_free(ix);
  return 0;
}
