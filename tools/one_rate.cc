// one_rate.cc — design-time measurement: N threads, each calling fmx_match_one (the reference's calling pattern: one
// pattern per call, cli/src/FuzzyMatch-cli.cc:139-147) for a few seconds; prints calls per second.
//   one_rate <index.fmi> <queries.txt: one pattern per line, vocabulary ids> <threads> <seconds> [coalesce_us]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "fmx.h"

int main(int argc, char** argv) {
  if (argc < 5) return 2;
  char err[256];
  fmx_index* ix = fmx_index_load(argv[1], err, sizeof err);
  if (!ix) { std::fprintf(stderr, "load: %s\n", err); return 1; }
  fmx_index_finalize(ix);
  if (fmx_index_to_device(ix, 0) != 0) { std::fprintf(stderr, "device: %s\n", fmx_last_error()); return 3; }
  if (argc > 5) fmx_set_option(ix, "coalesce_us", atoll(argv[5]));
  std::vector<std::vector<uint32_t>> qs;
  std::ifstream f(argv[2]);
  std::string line;
  while (std::getline(f, line)) {
    std::istringstream is(line);
    std::vector<uint32_t> q; uint32_t w;
    while (is >> w) q.push_back(w);
    if (!q.empty()) qs.push_back(q);
  }
  const int nt = atoi(argv[3]);
  const double secs = atof(argv[4]);
  fmx_params prm{};
  prm.fuzzy = 0.7f; prm.nmatch = 5; prm.min_subseq_length = 3; prm.min_subseq_ratio = 0.3f;
  prm.insert_cost = prm.delete_cost = prm.replace_cost = 1.f; prm.contrast_buffer = -1;
  std::atomic<uint64_t> calls{0}, matches{0};
  std::atomic<bool> stop{false};
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&, t]() {
      fmx_match out[5]; uint32_t cnt = 0, tot = 0;
      for (size_t i = t; !stop.load(std::memory_order_relaxed); i += nt) {
        const auto& q = qs[i % qs.size()];
        if (fmx_match_one(ix, &prm, q.data(), (uint32_t)q.size(), nullptr, nullptr, out, 5, &cnt, &tot) != 0) { std::fprintf(stderr, "%s\n", fmx_last_error()); break; }
        calls++; matches += cnt;
      }
    });
  const auto t0 = std::chrono::steady_clock::now();
  std::this_thread::sleep_for(std::chrono::duration<double>(0.5));  // warm-up
  const uint64_t c0 = calls.load();
  const auto t1 = std::chrono::steady_clock::now();
  std::this_thread::sleep_for(std::chrono::duration<double>(secs));
  const uint64_t c1 = calls.load();
  const auto t2 = std::chrono::steady_clock::now();
  stop = true;
  for (auto& x : th) x.join();
  (void)t0;
  std::printf("threads %d: %.0f calls/s (%.1f matches per call)\n", nt, (c1 - c0) / std::chrono::duration<double>(t2 - t1).count(),
              calls.load() ? (double)matches.load() / calls.load() : 0.0);
  fmx_index_free(ix);
  return 0;
}
