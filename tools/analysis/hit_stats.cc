// hit_stats.cc — design-time analysis (not product, not a test): distribution of the suffix-array hits the
// expand stage walks, by range size; how many are dominated, how many pass the phase-A length + signature
// test, and how many whole blocks of consecutive suffixes a per-block summary would reject.
//   g++ -O2 -std=c++17 -I ../../fuzzy-match_amd/csrc hit_stats.cc ../../fuzzy-match_amd/csrc/fmx_host.o \
//       -L .. -lfm_synth -Wl,-rpath,$PWD/.. -pthread -o hit_stats
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <vector>

#include "fmx_host.hh"

extern "C" {
struct fms_set;
fms_set* fms_corpus_new(uint64_t, double, uint32_t, uint64_t);
fms_set* fms_queries_new(const fms_set*, uint64_t, double, uint32_t, uint64_t);
uint64_t fms_num_sentences(const fms_set*);
uint32_t fms_vocab_size(const fms_set*);
const uint32_t* fms_offsets(const fms_set*);
const uint32_t* fms_tokens(const fms_set*);
}

using namespace fmx;

int main(int argc, char** argv) {
  const uint64_t n_sent = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000;
  const uint64_t nq = argc > 2 ? strtoull(argv[2], 0, 10) : 20000;
  const double mean = argc > 3 ? atof(argv[3]) : 25.0;
  Params prm{};
  prm.fuzzy = argc > 4 ? (float)atof(argv[4]) : 0.7f;
  prm.nmatch = 5; prm.min_subseq_length = 3; prm.min_subseq_ratio = 0.3f; prm.ins = prm.del = prm.rep = 1.f;
  prm.cbuf = 5; prm.idf_max = 0.01f; prm.max_tokens = 300;
  fms_set* c = fms_corpus_new(n_sent, mean, 50000, 12345);
  fms_set* qs = fms_queries_new(c, nq, mean, 50000, 54321);
  HostIndex h;
  h.add_ids_bulk(fms_offsets(c), fms_tokens(c), fms_num_sentences(c), fms_vocab_size(c));
  h.sort(8);
  h.derive(); h.derive_heavy();
  fprintf(stderr, "index: %zu suffixes\n", h.suf_rec.size());
  const uint32_t* q_off = fms_offsets(qs);
  const uint32_t* q_tok = fms_tokens(qs);
  GramTables gt{h.gram.data(), h.gram5.data(), h.gram.empty() ? 0u : (uint32_t)(h.gram.size() - 1),
                h.gram5.empty() ? 0u : (uint32_t)(h.gram5.size() - 1)};
  // classes by range size: [1,8) [8,64) [64,512) [512,4096) [4096,32768) [32768,inf)
  auto cls = [](uint32_t n) { return n < 8 ? 0 : n < 64 ? 1 : n < 512 ? 2 : n < 4096 ? 3 : n < 32768 ? 4 : 5; };
  struct Acc { uint64_t pl_tmin = 0, pl_len = 0, pl_2cls = 0, pl_4cls = 0; uint64_t ranges = 0, hits = 0, dom = 0, passA = 0, passA_nd = 0, lenpass = 0; uint64_t blk[4] = {0, 0, 0, 0}, blk_rej[4] = {0, 0, 0, 0}; uint64_t whole_dom = 0, whole_dom_hits = 0; };
  Acc acc[6];
  std::vector<uint64_t> per_query;
  const int B[4] = {8, 16, 32, 64};
  uint64_t nd_sent_dupes = 0;
  for (uint64_t q = 0; q < nq; q++) {
    const uint32_t p = q_off[q + 1] - q_off[q];
    const uint32_t* pat = q_tok + q_off[q];
    const uint32_t msl = min_seq_len(p, prm);
    if (msl == kNoMatchLen) { per_query.push_back(0); continue; }
    const uint32_t m = seed_len(p, msl);
    std::vector<uint16_t> cmin(65536, 0xffff);
    std::vector<uint8_t> hbit(p);
    for (uint32_t j = 0; j < p; j++) hbit[j] = (uint8_t)sig_bit(pat[j]);
    uint64_t qh = 0;
    uint32_t prev_lo = 0, prev_hi = 0;
    for (uint32_t it = 0; it + m <= p; it++) {
      uint32_t lo, hi;
      mgram_range_hashed(gt, h.suf_key.data(), h.suf_rec.data(), h.buf.data(), h.qva.data(), (uint32_t)h.qva.size(), pat + it, m, lo, hi);
      const uint32_t n = hi - lo;
      if (!n) { prev_lo = prev_hi = 0; continue; }
      Acc& a = acc[cls(n)];
      a.ranges++; a.hits += n; qh += n;
      uint64_t dom = 0;
      std::vector<uint8_t> pass(n, 0);
      for (uint32_t i = lo; i < hi; i++) {
        const SufRecB& rb = h.suf_b[i];
        const bool d = dominated_hit_b(rb, pat + it, it);
        dom += d;
        const uint32_t sl = rb.slen;
        if (cmin[sl] == 0xffff && !reject_len(p, sl, prm)) cmin[sl] = (uint16_t)cover_threshold(p, sl, prm);
        const bool lp = !reject_len(p, sl, prm);
        a.lenpass += lp;
        const bool ok = lp && cmin[sl] != 0xffff && sig_cover_upper(sig_bits(h.suf_sig[i]), hbit.data(), p) >= cmin[sl];
        pass[i - lo] = ok;
        a.passA += ok;
        a.passA_nd += ok && !d;
      }
      {
        // plane-test variants: uniform threshold(s) instead of the per-length table
        uint32_t tmin = 0xffff, smin = 0xffff, smax = 0, t_lo = 0xffff, t_hi = 0xffff, t4[4] = {0xffff, 0xffff, 0xffff, 0xffff};
        for (uint32_t sl = 1; sl < 255; sl++) {
          if (reject_len(p, sl, prm)) continue;
          const uint32_t cm = cover_threshold(p, sl, prm);
          if (cm == 0xffff) continue;
          tmin = std::min(tmin, cm); smin = std::min(smin, sl); smax = std::max(smax, sl);
        }
        for (uint32_t sl = smin; sl <= smax && smin != 0xffff; sl++) {
          const uint32_t cm = cover_threshold(p, sl, prm);
          if (sl <= p) t_lo = std::min(t_lo, cm); else t_hi = std::min(t_hi, cm);
          const uint32_t c4 = sl <= p ? 0 : std::min<uint32_t>(3, 1 + (sl - p - 1) * 3 / std::max<uint32_t>(1, smax - p));
          t4[c4] = std::min(t4[c4], cm);
        }
        for (uint32_t i = lo; i < hi; i++) {
          const uint32_t sl = h.suf_b[i].slen;
          const uint32_t ub = sig_cover_upper(sig_bits(h.suf_sig[i]), hbit.data(), p);
          a.pl_tmin += ub >= tmin;
          const bool in = sl >= smin && sl <= smax;
          a.pl_len += in && ub >= tmin;
          a.pl_2cls += in && ub >= (sl <= p ? t_lo : t_hi);
          const uint32_t c4 = sl <= p ? 0 : std::min<uint32_t>(3, 1 + (sl - p - 1) * 3 / std::max<uint32_t>(1, smax - p));
          a.pl_4cls += in && ub >= t4[c4];
        }
      }
      a.dom += dom;
      if (dom == n) { a.whole_dom++; a.whole_dom_hits += n; }
      for (int b = 0; b < 4; b++) {
        const uint32_t bs = B[b];
        const uint32_t first = (lo + bs - 1) / bs * bs;
        for (uint32_t s = first; s + bs <= hi; s += bs) {
          a.blk[b]++;
          // summary: OR of signatures, min of cmin over the block's lengths
          uint64_t orsig = 0; uint32_t mn = 0xffff;
          for (uint32_t i = s; i < s + bs; i++) {
            orsig |= sig_bits(h.suf_sig[i]);
            const uint32_t sl = h.suf_b[i].slen;
            const uint32_t cm = reject_len(p, sl, prm) ? 0xffffu : cmin[sl];
            mn = std::min(mn, cm);
          }
          if (mn == 0xffff || sig_cover_upper(orsig, hbit.data(), p) < mn) a.blk_rej[b]++;
        }
      }
      (void)prev_lo; (void)prev_hi; (void)nd_sent_dupes;
    }
    per_query.push_back(qh);
  }
  const char* names[6] = {"1-7", "8-63", "64-511", "512-4095", "4096-32767", "32768+"};
  uint64_t th = 0;
  for (auto& a : acc) th += a.hits;
  printf("class        ranges       hits  share  dominated  lenpass  passA  passA&!dom  whole-dom-hits   blkrej8 blkrej16 blkrej32 blkrej64 (fraction of hits in rejected whole blocks)\n");
  for (int k = 0; k < 6; k++) {
    Acc& a = acc[k];
    if (!a.hits) continue;
    printf("%-10s %8lu %10lu  %5.3f  %6.3f   %6.3f   %6.4f  %6.4f   %6.3f   ", names[k], a.ranges, a.hits, (double)a.hits / th,
           (double)a.dom / a.hits, (double)a.lenpass / a.hits, (double)a.passA / a.hits, (double)a.passA_nd / a.hits, (double)a.whole_dom_hits / a.hits);
    for (int b = 0; b < 4; b++) printf("  %6.3f", (double)a.blk_rej[b] * B[b] / a.hits);
    printf("  | plane pass: tmin %.4f +len %.4f 2cls %.4f 4cls %.4f", (double)a.pl_tmin / a.hits, (double)a.pl_len / a.hits, (double)a.pl_2cls / a.hits, (double)a.pl_4cls / a.hits);
    printf("\n");
  }
  std::sort(per_query.begin(), per_query.end());
  printf("hits/query: mean %.1f median %lu p90 %lu p99 %lu p99.9 %lu max %lu\n", (double)th / nq, per_query[nq / 2], per_query[nq * 9 / 10],
         per_query[nq * 99 / 100], per_query[nq * 999 / 1000], per_query.back());
  uint64_t cum = 0; size_t k = per_query.size();
  while (k > 0 && cum < th / 2) cum += per_query[--k];
  printf("half of all hits come from the top %zu queries (%.3f%%)\n", per_query.size() - k, 100.0 * (per_query.size() - k) / per_query.size());
  return 0;
}
