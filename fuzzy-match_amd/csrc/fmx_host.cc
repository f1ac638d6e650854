// fmx_host.cc — host-side index: vocabulary, suffix-array construction, derived SoA arrays and
// the .fmi (Boost binary archive) reader/writer, all without Boost.
#include "fmx_host.hh"

#include <math.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <fstream>
#include <functional>
#include <stdexcept>
#include <thread>

namespace fmx {

// ------------------------------------------------------------------------------------------
// Sentence / vocabulary
// ------------------------------------------------------------------------------------------

// Sentence::operator Tokens — src/sentence.cc:7-22,36-38 (a trailing empty part is dropped)
std::vector<std::string> RealTok::split() const {
  std::vector<std::string> parts;
  std::string accu;
  for (const char c : tokstring) {
    if (c == '\t') { parts.emplace_back(std::move(accu)); accu.clear(); }
    else accu += c;
  }
  if (!accu.empty()) parts.emplace_back(std::move(accu));
  return parts;
}

// VocabIndexer::VocabIndexer — src/vocab_indexer.cc:10-20: id 0 = "" (sentence separator), id 1 = ｟unk｠
HostIndex::HostIndex() {
  add_word(std::string());
  add_word("\xef\xbd\x9funk\xef\xbd\xa0");
}

uint32_t HostIndex::add_word(const std::string& w) {
  auto it = form2index.find(w);
  if (it != form2index.end()) return it->second;
  const uint32_t id = (uint32_t)forms.size();
  form2index.emplace(w, id);
  forms.push_back(w);
  sfreq.push_back(0);
  vocab_size = (uint32_t)forms.size();
  vocab_gen_.fetch_add(1, std::memory_order_release);
  return id;
}

uint32_t HostIndex::get_index(const std::string& w) const {
  auto it = form2index.find(w);
  return it != form2index.end() ? it->second : 1u;  // VOCAB_UNK
}

static inline uint64_t spell_hash(const char* s, size_t& len, uint64_t& head) {  // FNV-1a, 64 bit; head = the first eight bytes
  uint64_t h = 1469598103934665603ull, hd = 0;
  size_t n = 0;
  for (; s[n]; n++) {
    h ^= (unsigned char)s[n]; h *= 1099511628211ull;
    if (n < 8) hd |= (uint64_t)(unsigned char)s[n] << (8 * n);
  }
  len = n;
  head = hd;
  return h ^ (h >> 29);
}

const HostIndex::FastVocab& HostIndex::fast_vocab() const {
  const uint64_t gen = vocab_gen_.load(std::memory_order_acquire);
  if (fast_.built_for.load(std::memory_order_acquire) == gen) return fast_;  // the common case: no lock
  std::lock_guard<std::mutex> lk(fast_mu_);
  if (fast_.built_for.load(std::memory_order_relaxed) == gen && !fast_.slot.empty()) return fast_;
  FastVocab& f = fast_;
  uint64_t cap = 64;
  while (cap < forms.size() * 2) cap <<= 1;
  f.slot.assign(cap, ~0ull);
  f.head.assign(cap, 0ull);
  f.mask = cap - 1;
  f.arena.clear();
  for (size_t id = 0; id < forms.size(); id++) {
    if (id >= 2 && forms[id].empty()) continue;  // ids without a spelling (partially attached vocabularies)
    if (form2index.find(forms[id]) == form2index.end() || form2index.at(forms[id]) != id) continue;  // not the id getIndex returns
    const uint64_t off = f.arena.size();
    f.arena.insert(f.arena.end(), forms[id].begin(), forms[id].end());
    f.arena.push_back('\0');
    size_t len;
    uint64_t hd;
    uint64_t h = spell_hash(f.arena.data() + off, len, hd) & f.mask;
    while (f.slot[h] != ~0ull) h = (h + 1) & f.mask;
    f.slot[h] = (off << 32) | (uint64_t)id;
    f.head[h] = hd;
  }
  f.built_for.store(gen, std::memory_order_release);
  return f;
}

HostIndex::VocabTable HostIndex::vocab_table() const {
  const FastVocab& f = fast_vocab();
  return VocabTable{f.slot.data(), f.head.data(), f.arena.data(), (uint64_t)f.slot.size(), (uint64_t)f.arena.size(), f.built_for.load()};
}

// The interned real tokens of the index (derive(): real_intern) as the same kind of table: spelling -> interned id, for the
// device-side lookup of a text batch's real tokens.  Built once per derive(); the empty spelling has no entry (a packed text
// cannot hold it).
HostIndex::VocabTable HostIndex::real_table() const {
  std::lock_guard<std::mutex> lk(fast_mu_);
  FastVocab& f = real_fast_;
  if (f.built_for.load() != real_gen_ || f.slot.empty()) {
    uint64_t cap = 64;
    while (cap < real_intern.size() * 2) cap <<= 1;
    f.slot.assign(cap, ~0ull);
    f.head.assign(cap, 0ull);
    f.mask = cap - 1;
    f.arena.clear();
    for (const auto& kv : real_intern) {
      if (kv.first.empty()) continue;
      const uint64_t off = f.arena.size();
      f.arena.insert(f.arena.end(), kv.first.begin(), kv.first.end());
      f.arena.push_back('\0');
      size_t len;
      uint64_t hd;
      uint64_t h = spell_hash(f.arena.data() + off, len, hd) & f.mask;
      while (f.slot[h] != ~0ull) h = (h + 1) & f.mask;
      f.slot[h] = (off << 32) | (uint64_t)kv.second;
      f.head[h] = hd;
    }
    f.built_for.store(real_gen_);
  }
  return VocabTable{f.slot.data(), f.head.data(), f.arena.data(), (uint64_t)f.slot.size(), (uint64_t)f.arena.size(), f.built_for.load()};
}

void HostIndex::get_indices(const char* const* toks, uint64_t n, uint32_t* ids, int nthreads) const {
  const FastVocab& f = fast_vocab();
  auto work = [&](uint64_t lo, uint64_t hi) {
    for (uint64_t i = lo; i < hi; i++) {
      const char* t = toks[i];
      size_t len;
      uint64_t hd;
      uint64_t h = spell_hash(t, len, hd) & f.mask;
      uint32_t id = 1u;  // VOCAB_UNK
      for (;;) {
        const uint64_t e = f.slot[h];
        if (e == ~0ull) break;
        if (f.head[h] == hd) {
          // up to seven bytes: equal heads are equal spellings (the zero padding carries the length); longer ones: the rest
          // (and the terminator) from the arena
          if (len < 8 || memcmp(f.arena.data() + (e >> 32) + 8, t + 8, len - 8 + 1) == 0) { id = (uint32_t)e; break; }
        }
        h = (h + 1) & f.mask;
      }
      ids[i] = id;
    }
  };
  const uint64_t nt = (uint64_t)std::max(1, std::min<int>(nthreads, (int)std::min<uint64_t>(n / 4096 + 1, 256)));
  if (nt == 1) { work(0, n); return; }
  std::vector<std::thread> th;
  for (uint64_t k = 1; k < nt; k++) th.emplace_back(work, n * k / nt, n * (k + 1) / nt);
  work(0, n / nt);
  for (auto& t : th) t.join();
}

void HostIndex::set_forms(const char* const* spellings, uint64_t n) {
  if (!ids.empty()) throw std::runtime_error("set_forms: the index already holds string-level sentences");
  if (n > forms.size()) { forms.resize(n); if (sfreq.size() < n) sfreq.resize(n, 0); }
  for (uint64_t id = 2; id < n; id++) {
    if (!spellings[id]) continue;
    if (forms[id] == spellings[id]) continue;
    if (!forms[id].empty()) {  // a spelling is being replaced: the old one no longer names this id
      auto old = form2index.find(forms[id]);
      if (old != form2index.end() && old->second == (uint32_t)id) form2index.erase(old);
    }
    forms[id] = spellings[id];
    form2index.emplace(forms[id], (uint32_t)id);  // the first id of a spelling wins, like addWord
  }
  if (forms.size() > vocab_size) vocab_size = (uint32_t)forms.size();
  vocab_gen_.fetch_add(1, std::memory_order_release);
}

static void push_sentence(HostIndex& ix, const uint32_t* toks, uint32_t n) {
  // SuffixArray::add_sentence — src/suffix_array.cc:9-27
  const uint32_t sidx = (uint32_t)ix.sent_pos.size();
  ix.sent_pos.push_back((uint32_t)ix.buf.size());
  ix.buf.push_back(n);
  for (uint32_t i = 0; i < n; i++) {
    ix.buf.push_back(toks[i]);
    ix.suf_sid.push_back(sidx);
    ix.suf_pos.push_back((uint16_t)(i + 1));
  }
  ix.buf.push_back(0);
  ix.sorted = false;
}

int64_t HostIndex::add_tm(const std::string& id, const RealTok& real_tokens, const std::vector<std::string>& norm) {
  // SuffixArrayIndex::add_tm — src/suffix_array_index.cc:16
  if (real_tokens.tokstring.empty() || norm.size() > max_tokens || norm.size() > 65535) return -1;
  // VocabIndexer::addWords — src/vocab_indexer.cc:73-90: sfreq bumped once per distinct word
  std::vector<uint32_t> wids;
  wids.reserve(norm.size());
  for (const auto& w : norm) wids.push_back(add_word(w));
  std::vector<uint32_t> distinct(wids);
  std::sort(distinct.begin(), distinct.end());
  distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
  for (uint32_t w : distinct) sfreq[w]++;
  push_sentence(*this, wids.data(), (uint32_t)wids.size());
  ids.push_back(id);
  real.push_back(real_tokens);
  return (int64_t)sent_pos.size() - 1;
}

// CPUs this process may really use: min(affinity mask, cgroup v2 quota).  (The GPU boxes show 256 hardware threads and grant
// 16: sizing thread pools by hardware_concurrency() there oversubscribes sixteen-fold.)
unsigned usable_cpus() {
  static const unsigned n = [] {
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) hw = std::min<unsigned>(hw, (unsigned)c); }
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      char q[64] = {0};
      double period = 0;
      if (fscanf(f, "%63s %lf", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
        const double quota = atof(q);
        if (quota > 0) hw = std::min<unsigned>(hw, std::max<unsigned>(1u, (unsigned)(quota / period + 0.5)));
      }
      fclose(f);
    }
    return std::max(1u, hw);
  }();
  return n;
}

// Ids of many sentences at once (SuffixArrayIndex::add_tm per sentence, src/suffix_array_index.cc:10-28, without the strings):
// where every sentence goes is a prefix sum, so the buffer, the suffix enumeration (SuffixArray::add_sentence,
// src/suffix_array.cc:9-27) and the sentence frequencies (VocabIndexer::addWords, src/vocab_indexer.cc:73-90: once per distinct
// word of a sentence) are filled by all usable CPUs side by side — round 4 did this in one loop with a sort + unique per sentence:
// 1.9 of the 2.5 s a 5 M-sentence build took.
void HostIndex::add_ids_bulk(const uint32_t* off, const uint32_t* toks, uint64_t n_sent, uint32_t vocab_size_) {
  if (!ids.empty() || !real.empty())
    throw std::runtime_error("add_ids_bulk: index already holds string-level sentences");
  if (vocab_size_ > vocab_size) vocab_size = vocab_size_;
  // (every thread keeps two words per vocabulary entry: fewer threads for a vocabulary of many millions)
  const uint64_t nt = std::max<uint64_t>(1, std::min<uint64_t>(std::min<uint64_t>(usable_cpus(), n_sent / 16384 + 1),
                                                                (1ull << 27) / std::max<uint64_t>(std::max(vocab_size, vocab_size_), 1) + 1));
  auto run = [&](const std::function<void(uint64_t, uint64_t, uint64_t)>& fn) {  // fn(thread, first sentence, end)
    if (nt == 1) { fn(0, 0, n_sent); return; }
    std::vector<std::thread> th;
    for (uint64_t k = 1; k < nt; k++) th.emplace_back(fn, k, n_sent * k / nt, n_sent * (k + 1) / nt);
    fn(0, 0, n_sent / nt);
    for (auto& t : th) t.join();
  };
  // 1. per thread: sentences taken, their tokens, the largest word id
  std::vector<uint64_t> t_sent(nt + 1, 0), t_tok(nt + 1, 0);
  std::vector<uint32_t> t_maxw(nt, 0);
  run([&](uint64_t k, uint64_t s0, uint64_t s1) {
    uint64_t ns = 0, ntok = 0;
    uint32_t mw = 0;
    for (uint64_t s = s0; s < s1; s++) {
      const uint32_t n = off[s + 1] - off[s];
      if (n > max_tokens || n > 65535) continue;
      ns++; ntok += n;
      const uint32_t* t = toks + off[s];
      for (uint32_t i = 0; i < n; i++) mw = t[i] > mw ? t[i] : mw;
    }
    t_sent[k + 1] = ns; t_tok[k + 1] = ntok; t_maxw[k] = mw;
  });
  uint32_t max_w = 0;
  for (uint64_t k = 0; k < nt; k++) { t_sent[k + 1] += t_sent[k]; t_tok[k + 1] += t_tok[k]; max_w = std::max(max_w, t_maxw[k]); }
  if (t_sent[nt] && max_w >= vocab_size) vocab_size = max_w + 1;
  sfreq.resize(vocab_size, 0);
  const uint64_t sent0 = sent_pos.size(), buf0 = buf.size(), suf0 = suf_sid.size();
  if (sent0 + t_sent[nt] > 0xffffffffull || buf0 + t_tok[nt] + 2 * t_sent[nt] > 0xffffffffull)
    throw std::runtime_error("add_ids_bulk: more than 2^32 - 1 sentences or buffer words");
  sent_pos.resize(sent0 + t_sent[nt]);
  buf.resize(buf0 + t_tok[nt] + 2 * t_sent[nt]);
  suf_sid.resize(suf0 + t_tok[nt]);
  suf_pos.resize(suf0 + t_tok[nt]);
  // 2. fill; sentence frequencies per thread (a word is counted when the sentence at hand has not stamped it yet)
  std::vector<std::vector<uint32_t>> t_freq(nt);
  run([&](uint64_t k, uint64_t s0, uint64_t s1) {
    std::vector<uint32_t>& freq = t_freq[k];
    freq.assign(vocab_size, 0);
    std::vector<uint32_t> stamp(vocab_size, 0);
    uint64_t sidx = sent0 + t_sent[k], bp = buf0 + t_tok[k] + 2 * t_sent[k], sp = suf0 + t_tok[k];
    uint32_t mark = 0;
    for (uint64_t s = s0; s < s1; s++) {
      const uint32_t n = off[s + 1] - off[s];
      if (n > max_tokens || n > 65535) continue;
      const uint32_t* t = toks + off[s];
      if (++mark == 0) { std::fill(stamp.begin(), stamp.end(), 0u); mark = 1; }
      sent_pos[sidx] = (uint32_t)bp;
      buf[bp++] = n;
      for (uint32_t i = 0; i < n; i++) {
        const uint32_t w = t[i];
        buf[bp++] = w;
        suf_sid[sp] = (uint32_t)sidx;
        suf_pos[sp] = (uint16_t)(i + 1);
        sp++;
        if (stamp[w] != mark) { stamp[w] = mark; freq[w]++; }
      }
      buf[bp++] = 0;  // VocabIndexer::SENTENCE_SEPARATOR
      sidx++;
    }
  });
  for (uint64_t k = 0; k < nt; k++)
    for (uint32_t w = 0; w < vocab_size; w++) sfreq[w] += t_freq[k][w];
  if (t_sent[nt]) sorted = false;
}

// ------------------------------------------------------------------------------------------
// suffix sort (CPU)
// ------------------------------------------------------------------------------------------

namespace {

struct SortKey {
  uint64_t prefix;  // tok0 << 32 | tok1 (0 when the suffix has one token): decides most compares
  uint64_t key;     // sid << 16 | pos
};

// SuffixArray::comp — src/suffix_array.cc:235-251 (tokens, then shorter first, then sentence id)
struct SuffixLess {
  const uint32_t* buf;
  const uint32_t* sent_pos;
  bool operator()(const SortKey& a, const SortKey& b) const {
    if (a.prefix != b.prefix) return a.prefix < b.prefix;
    const uint32_t sa = (uint32_t)(a.key >> 16), sb = (uint32_t)(b.key >> 16);
    const uint32_t pa = (uint32_t)(a.key & 0xffff), pb = (uint32_t)(b.key & 0xffff);
    const uint32_t* ba = buf + sent_pos[sa];
    const uint32_t* bb = buf + sent_pos[sb];
    const uint32_t la = ba[0] - (pa - 1), lb = bb[0] - (pb - 1);
    const uint32_t* ta = ba + pa;
    const uint32_t* tb = bb + pb;
    const uint32_t mn = la < lb ? la : lb;
    for (uint32_t i = 0; i < mn; i++) {
      if (ta[i] != tb[i]) return ta[i] < tb[i];
    }
    if (la != lb) return la < lb;
    return sa < sb;
  }
};

template <class T, class Less>
void parallel_sort(T* a, uint64_t n, Less less, int nthreads) {
  if (nthreads <= 1 || n < (1u << 15)) { std::sort(a, a + n, less); return; }
  int parts = 1;
  while (parts < nthreads) parts <<= 1;
  std::vector<uint64_t> bound(parts + 1);
  for (int i = 0; i <= parts; i++) bound[i] = n * (uint64_t)i / (uint64_t)parts;
  {
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    auto work = [&]() {
      for (;;) {
        int k = next.fetch_add(1);
        if (k >= parts) break;
        std::sort(a + bound[k], a + bound[k + 1], less);
      }
    };
    for (int t = 1; t < nthreads; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
  for (int width = 1; width < parts; width <<= 1) {
    std::atomic<int> next{0};
    const int merges = parts / (2 * width);
    std::vector<std::thread> th;
    auto work = [&]() {
      for (;;) {
        int k = next.fetch_add(1);
        if (k >= merges) break;
        const int lo = k * 2 * width;
        std::inplace_merge(a + bound[lo], a + bound[lo + width], a + bound[lo + 2 * width], less);
      }
    };
    const int nt = std::min(nthreads, merges);
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (auto& t : th) t.join();
  }
}

}  // namespace

void HostIndex::sort(int nthreads) {
  if (sorted) return;
  const uint64_t n = suf_sid.size();
  std::vector<SortKey> keys(n);
  for (uint64_t i = 0; i < n; i++) {
    const uint32_t  sid = suf_sid[i], pos = suf_pos[i];
    const uint32_t* b = buf.data() + sent_pos[sid];
    const uint32_t  rem = b[0] - (pos - 1);
    keys[i].prefix = ((uint64_t)b[pos] << 32) | (rem > 1 ? b[pos + 1] : 0u);
    keys[i].key = ((uint64_t)sid << 16) | pos;
  }
  parallel_sort(keys.data(), n, SuffixLess{buf.data(), sent_pos.data()}, nthreads);
  std::vector<uint64_t> flat(n);
  for (uint64_t i = 0; i < n; i++) flat[i] = keys[i].key;
  keys.clear();
  keys.shrink_to_fit();
  adopt_sorted(flat.data(), n);
}

void HostIndex::adopt_sorted(const uint64_t* keys, uint64_t n) {
  if (n != suf_sid.size()) throw std::runtime_error("adopt_sorted: size mismatch");
  // _quickVocabAccess — src/suffix_array.cc:76-98: first suffix per first-word id, [vocab_size] = total
  for (uint64_t i = 0; i < n; i++) {
    const uint32_t w = buf[sent_pos[(uint32_t)(keys[i] >> 16)] + (uint32_t)(keys[i] & 0xffff)];
    if (w >= vocab_size) { vocab_size = w + 1; sfreq.resize(vocab_size, 0); }
  }
  qva.assign((size_t)vocab_size + 1, 0);
  for (uint64_t i = 0; i < n; i++) {
    const uint32_t sid = (uint32_t)(keys[i] >> 16), pos = (uint32_t)(keys[i] & 0xffff);
    suf_sid[i] = sid;
    suf_pos[i] = (uint16_t)pos;
    qva[buf[sent_pos[sid] + pos] + 1]++;
  }
  for (uint32_t w = 0; w < vocab_size; w++) qva[w + 1] += qva[w];
  sorted = true;
}

// true when suffix a sorts before suffix b under SuffixArray::comp (src/suffix_array.cc:235-251)
static bool suffix_before(const std::vector<uint32_t>& buf, const std::vector<uint32_t>& sent_pos, uint32_t sa, uint32_t pa,
                          uint32_t sb, uint32_t pb) {
  const uint32_t* ba = buf.data() + sent_pos[sa];
  const uint32_t* bb = buf.data() + sent_pos[sb];
  const uint32_t la = ba[0] - (pa - 1), lb = bb[0] - (pb - 1);
  const uint32_t mn = la < lb ? la : lb;
  for (uint32_t i = 0; i < mn; i++) {
    const uint32_t x = ba[pa + i], y = bb[pb + i];
    if (x != y) return x < y;
  }
  if (la != lb) return la < lb;
  return sa < sb;
}

void HostIndex::adopt_arrays(const uint32_t* sid, const uint16_t* pos, uint64_t n_suf, const uint32_t* buffer, uint64_t n_buf,
                             const uint32_t* spos, uint64_t n_sent, const uint32_t* qva_in, uint64_t n_qva,
                             const uint32_t* sfreq_in, uint32_t vocab_size_) {
  if (!sent_pos.empty() || !buf.empty()) throw std::runtime_error("adopt_arrays: index is not empty");
  if (n_sent > 0xffffffffull || n_buf > 0xffffffffull) throw std::runtime_error("adopt_arrays: more than 2^32 - 1 sentences / buffer words");
  // the sentence buffer: [len, tok x len, 0] per sentence, sentence_pos = offsets of the len words, in order
  uint64_t expect = 0, tokens = 0;
  uint32_t vs = vocab_size_;
  for (uint64_t s = 0; s < n_sent; s++) {
    if (spos[s] != expect || expect >= n_buf) throw std::runtime_error("adopt_arrays: sentence_pos does not walk the sentence buffer");
    const uint64_t len = buffer[expect];
    if (len == 0 || len > 65535 || expect + len + 1 >= n_buf || buffer[expect + len + 1] != 0)
      throw std::runtime_error("adopt_arrays: malformed sentence in the buffer");
    for (uint64_t k = 1; k <= len; k++) { const uint32_t w = buffer[expect + k]; if (w >= vs) vs = w + 1; }
    tokens += len;
    expect += len + 2;
  }
  if (expect != n_buf) throw std::runtime_error("adopt_arrays: trailing words in the sentence buffer");
  if (n_suf != tokens) throw std::runtime_error("adopt_arrays: the suffix array must hold one suffix per token");
  buf.assign(buffer, buffer + n_buf);
  sent_pos.assign(spos, spos + n_sent);
  suf_sid.assign(sid, sid + n_suf);
  suf_pos.assign(pos, pos + n_suf);
  for (uint64_t i = 0; i < n_suf; i++)
    if (suf_sid[i] >= n_sent || suf_pos[i] == 0 || suf_pos[i] > buf[sent_pos[suf_sid[i]]])
      throw std::runtime_error("adopt_arrays: suffix outside its sentence");
  // order: every adjacent pair up to a million suffixes, an evenly spread sample of a million pairs beyond
  const uint64_t step = n_suf > (1ull << 20) ? n_suf >> 20 : 1;
  for (uint64_t i = 0; i + 1 < n_suf; i += step)
    if (!suffix_before(buf, sent_pos, suf_sid[i], suf_pos[i], suf_sid[i + 1], suf_pos[i + 1]))
      throw std::runtime_error("adopt_arrays: the suffix array is not in SuffixArray::comp order");
  vocab_size = vs;
  // _quickVocabAccess — src/suffix_array.cc:76-98 (recomputed; a caller-supplied copy must agree)
  qva.assign((size_t)vocab_size + 1, 0);
  for (uint64_t i = 0; i < n_suf; i++) qva[buf[sent_pos[suf_sid[i]] + suf_pos[i]] + 1]++;
  for (uint32_t w = 0; w < vocab_size; w++) qva[w + 1] += qva[w];
  if (qva_in) {
    // the reference's vector has one entry per word id it has seen plus the total; compare the common prefix
    const uint64_t n_cmp = std::min<uint64_t>(n_qva, qva.size());
    for (uint64_t w = 0; w < n_cmp; w++)
      if (qva_in[w] != qva[w]) throw std::runtime_error("adopt_arrays: _quickVocabAccess does not match the suffix array");
  }
  // VocabIndexer::sfreq — sentences containing the word (src/vocab_indexer.cc:37-50)
  sfreq.assign(vocab_size, 0);
  if (sfreq_in) {
    for (uint32_t w = 0; w < vocab_size_ && w < vocab_size; w++) sfreq[w] = sfreq_in[w];
  } else {
    std::vector<uint32_t> distinct;
    for (uint64_t s = 0; s < n_sent; s++) {
      const uint32_t* t = buf.data() + sent_pos[s] + 1;
      distinct.assign(t, t + buf[sent_pos[s]]);
      std::sort(distinct.begin(), distinct.end());
      distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
      for (uint32_t w : distinct) sfreq[w]++;
    }
  }
  sorted = true;
}

void HostIndex::adopt_sorted_arrays(std::vector<uint32_t>&& sid, std::vector<uint16_t>&& pos, std::vector<uint32_t>&& qva_in) {
  if (sid.size() != suf_sid.size() || pos.size() != suf_pos.size()) throw std::runtime_error("adopt_sorted_arrays: size mismatch");
  if (qva_in.size() < 2) throw std::runtime_error("adopt_sorted_arrays: empty _quickVocabAccess");
  const uint32_t vs = (uint32_t)qva_in.size() - 1;
  if (vs > vocab_size) { vocab_size = vs; sfreq.resize(vocab_size, 0); }
  if (vs < vocab_size) qva_in.resize((size_t)vocab_size + 1, qva_in.back());  // words beyond the largest first token: empty buckets at the end
  suf_sid = std::move(sid);
  suf_pos = std::move(pos);
  qva = std::move(qva_in);
  sorted = true;
}

void HostIndex::bucket_keys(std::vector<uint64_t>& keys, std::vector<uint64_t>& bucket_off) const {
  const uint64_t n = suf_sid.size();
  keys.resize(n);
  for (uint64_t i = 0; i < n; i++) keys[i] = ((uint64_t)suf_sid[i] << 16) | suf_pos[i];
  bucket_off.clear();
}

// ------------------------------------------------------------------------------------------
// derived arrays
// ------------------------------------------------------------------------------------------

int edit_distance_char(const char* s1, int n1, const char* s2, int n2) {
  if (n1 == 0) return n2;
  if (n2 == 0) return n1;
  std::vector<int> prev(n2 + 1), curr(n2 + 1);
  for (int j = 0; j <= n2; j++) prev[j] = j;
  for (int i = 1; i <= n1; i++) {
    curr[0] = i;
    for (int j = 1; j <= n2; j++)
      curr[j] = std::min({prev[j] + 1, curr[j - 1] + 1, prev[j - 1] + (s1[i - 1] == s2[j - 1] ? 0 : 1)});
    std::swap(prev, curr);
  }
  return prev[n2];
}

// Trigram table (fmx_core.hh: GramEntry): one entry per distinct (w0, w1, w2) that starts at least one
// suffix with three or more tokens.  Groups are contiguous in the suffix array; built in parallel with
// CAS inserts, load factor <= 0.5.
void HostIndex::build_gram_table(int nthreads) {
  gram.clear();
  const uint64_t n = suf_rec.size();
  if (n == 0) return;
  if (nthreads < 1) nthreads = 1;
  auto w0_of = [&](uint64_t i) { return buf[suf_rec[i].off]; };
  auto is_start = [&](uint64_t i) {
    if (suf_rec[i].rem < 3) return false;
    if (i == 0 || suf_rec[i - 1].rem < 3) return true;
    return w0_of(i) != w0_of(i - 1) || suf_key[i].t[0] != suf_key[i - 1].t[0] || suf_key[i].t[1] != suf_key[i - 1].t[1];
  };
  std::vector<uint64_t> part_count(nthreads, 0);
  auto run = [&](auto fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < nthreads; t++) th.emplace_back(fn, t);
    fn(0);
    for (auto& x : th) x.join();
  };
  run([&](int t) {
    const uint64_t a = n * (uint64_t)t / (uint64_t)nthreads, b = n * (uint64_t)(t + 1) / (uint64_t)nthreads;
    uint64_t c = 0;
    for (uint64_t i = a; i < b; i++) c += is_start(i) ? 1 : 0;
    part_count[t] = c;
  });
  uint64_t groups = 0;
  for (auto c : part_count) groups += c;
  if (groups == 0) return;
  uint64_t cap = 16;
  while (cap < groups * 2) cap <<= 1;
  if (cap > (1ull << 31)) return;  // would not fit 32-bit slot arithmetic: the keyed search is used instead
  GramEntry empty;
  empty.w1 = empty.w2 = 0; empty.lo = kGramEmpty; empty.cnt = 0;
  gram.assign(cap, empty);
  const uint32_t mask = (uint32_t)(cap - 1);
  run([&](int t) {
    const uint64_t a = n * (uint64_t)t / (uint64_t)nthreads, b = n * (uint64_t)(t + 1) / (uint64_t)nthreads;
    for (uint64_t i = a; i < b; i++) {
      if (!is_start(i)) continue;
      uint64_t e = i + 1;  // the group may extend into the next thread's slice
      const uint32_t w0 = w0_of(i), w1 = suf_key[i].t[0], w2 = suf_key[i].t[1];
      while (e < n && suf_rec[e].rem >= 3 && w0_of(e) == w0 && suf_key[e].t[0] == w1 && suf_key[e].t[1] == w2) e++;
      uint32_t slot = gram_hash(w0, w1, w2) & mask;
      for (;;) {
        uint32_t expect = kGramEmpty;
        if (__atomic_compare_exchange_n(&gram[slot].lo, &expect, (uint32_t)i, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
          gram[slot].w1 = w1; gram[slot].w2 = w2; gram[slot].cnt = (uint32_t)(e - i);
          break;
        }
        slot = (slot + 1) & mask;
      }
    }
  });

  // ---- 5-gram buckets: every distinct 5-gram (fmx_core.hh: G5Bucket); 7-gram entries under the heavy ones ----
  gram5.clear();
  gram7.clear();
  // a 5-gram group: a maximal run of suffixes with at least five tokens and the same first five
  auto same5 = [&](uint64_t a, uint64_t b2) {
    return w0_of(a) == w0_of(b2) && suf_key[a].t[0] == suf_key[b2].t[0] && suf_key[a].t[1] == suf_key[b2].t[1] &&
           suf_key[a].t[2] == suf_key[b2].t[2] && suf_key[a].t[3] == suf_key[b2].t[3];
  };
  auto is_start5 = [&](uint64_t i) {
    if (suf_rec[i].rem < 5) return false;
    if (i == 0 || suf_rec[i - 1].rem < 5) return true;
    return !same5(i, i - 1);
  };
  auto for_groups5 = [&](int t, auto fn) {  // fn(lo, cnt)
    const uint64_t a = n * (uint64_t)t / (uint64_t)nthreads, b = n * (uint64_t)(t + 1) / (uint64_t)nthreads;
    for (uint64_t i = a; i < b; i++) {
      if (!is_start5(i)) continue;
      uint64_t e = i + 1;  // the group may extend into the next thread's slice
      while (e < n && suf_rec[e].rem >= 5 && same5(e, i)) e++;
      fn(i, e - i);
    }
  };
  std::fill(part_count.begin(), part_count.end(), 0);
  run([&](int t) { uint64_t c = 0; for_groups5(t, [&](uint64_t, uint64_t) { c++; }); part_count[t] = c; });
  uint64_t groups5 = 0;
  for (auto c : part_count) groups5 += c;
  if (groups5 == 0) return;
  const uint64_t nb5 = std::max<uint64_t>(16, (groups5 * 16 + kG5FillPer16 - 1) / kG5FillPer16);
  if (nb5 > 0xffffffffull) return;  // (the trigram table + keys are used instead)
  {
    G5Bucket zero;
    memset(&zero, 0, sizeof zero);
    gram5.assign(nb5, zero);
  }
  run([&](int t) {
    for_groups5(t, [&](uint64_t lo, uint64_t cnt) {
      const SufKey& first = suf_key[lo];
      const SufKey& last = suf_key[lo + cnt - 1];
      uint32_t lcp = 0;
      while (lcp < 4 && first.t[4 + lcp] == last.t[4 + lcp]) lcp++;
      uint32_t b = g5_bucket_of(gram5_hash(w0_of(lo), first.t[0], first.t[1], first.t[2], first.t[3]), (uint32_t)nb5);
      const uint32_t tag = gram5_tag(w0_of(lo), first.t[0], first.t[1], first.t[2], first.t[3]);
      for (;;) {
        // claim a slot: entries in use = bits 16..17 of dir_b (compare-and-swap: the count never passes three)
        uint32_t old = __atomic_load_n(&gram5[b].dir_b, __ATOMIC_ACQUIRE), slot = kG5PerBucket;
        while (g5_used(old) < kG5PerBucket) {
          if (__atomic_compare_exchange_n(&gram5[b].dir_b, &old, old + (1u << 16), false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) { slot = g5_used(old); break; }
        }
        if (slot < kG5PerBucket) {
          G5Entry& e = gram5[b].e[slot];
          e.w1 = first.t[0]; e.w2 = first.t[1]; e.w3 = first.t[2]; e.w4 = first.t[3];
          e.lo = (uint32_t)lo; e.cnt = (uint32_t)cnt;
          for (int k = 0; k < 4; k++) e.t[k] = first.t[4 + k];
          if (slot < 2) __atomic_fetch_or(&gram5[b].dir_a, tag << (16 * slot), __ATOMIC_ACQ_REL);
          __atomic_fetch_or(&gram5[b].dir_b, (slot == 2 ? tag : 0u) | (lcp << (20 + 3 * slot)), __ATOMIC_ACQ_REL);
          break;
        }
        __atomic_fetch_or(&gram5[b].dir_b, 1u << 18, __ATOMIC_ACQ_REL);
        b = b + 1 == nb5 ? 0u : b + 1;
      }
    });
  });
  // second level: the distinct 7-grams of the 5-grams with more than kDeepGramMin suffixes
  auto for_deep7 = [&](int t, auto fn) {  // fn(lo5, w5, w6, lo7, cnt7)
    for_groups5(t, [&](uint64_t lo, uint64_t cnt) {
      if (cnt <= kDeepGramMin) return;
      for (uint64_t j = lo; j < lo + cnt;) {
        if (suf_rec[j].rem < 7) { j++; continue; }
        uint64_t g = j + 1;
        while (g < lo + cnt && suf_rec[g].rem >= 7 && suf_key[g].t[4] == suf_key[j].t[4] && suf_key[g].t[5] == suf_key[j].t[5]) g++;
        fn((uint32_t)lo, suf_key[j].t[4], suf_key[j].t[5], (uint32_t)j, (uint32_t)(g - j));
        j = g;
      }
    });
  };
  std::fill(part_count.begin(), part_count.end(), 0);
  run([&](int t) { uint64_t c = 0; for_deep7(t, [&](uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) { c++; }); part_count[t] = c; });
  uint64_t groups7 = 0;
  for (auto c : part_count) groups7 += c;
  if (groups7 == 0) return;
  uint64_t cap7 = 16;
  while (cap7 < groups7 * 2) cap7 <<= 1;
  if (cap7 > (1ull << 31)) return;
  gram7.assign(cap7, empty);
  const uint32_t mask7 = (uint32_t)(cap7 - 1);
  run([&](int t) {
    for_deep7(t, [&](uint32_t lo5, uint32_t w5, uint32_t w6, uint32_t lo7, uint32_t cnt7) {
      uint32_t slot = gram_hash(lo5, w5, w6) & mask7;
      for (;;) {
        uint32_t expect = kGramEmpty;
        if (__atomic_compare_exchange_n(&gram7[slot].lo, &expect, lo7, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
          gram7[slot].w1 = w5; gram7[slot].w2 = w6; gram7[slot].cnt = cnt7;
          break;
        }
        slot = (slot + 1) & mask7;
      }
    });
  });
}

constexpr uint32_t kMaxItokStrings = 32768;  // the table is quadratic: 2 GB at the limit

uint32_t HostIndex::intern_itok(const std::string& s) {
  std::lock_guard<std::mutex> lk(itok_sync_mu);
  if (itok_strings.empty()) {
    itok_strings.emplace_back();
    itok_intern.emplace(std::string(), 0u);
    itok_cap = 16;
    itok_tab.assign((size_t)itok_cap * itok_cap, 0);
  }
  auto it = itok_intern.find(s);
  if (it != itok_intern.end()) return it->second;
  const uint32_t id = (uint32_t)itok_strings.size();
  if (id >= kMaxItokStrings) throw std::runtime_error("too many distinct penalty-token strings (limit 32768)");
  itok_strings.push_back(s);
  itok_intern.emplace(s, id);
  if (id >= itok_cap) {  // stride doubles: re-lay the rows out
    const uint32_t cap2 = itok_cap * 2;
    std::vector<uint16_t> t((size_t)cap2 * cap2, 0);
    for (uint32_t a = 0; a < id; a++) memcpy(&t[(size_t)a * cap2], &itok_tab[(size_t)a * itok_cap], (size_t)id * sizeof(uint16_t));
    itok_tab.swap(t);
    itok_cap = cap2;
  }
  for (uint32_t a = 0; a <= id; a++) {  // the new row and column
    const int v = edit_distance_char(itok_strings[a].data(), (int)itok_strings[a].size(), itok_strings[id].data(), (int)itok_strings[id].size());
    const int w = a == id ? v : edit_distance_char(itok_strings[id].data(), (int)itok_strings[id].size(), itok_strings[a].data(), (int)itok_strings[a].size());
    itok_tab[(size_t)a * itok_cap + id] = (uint16_t)std::min(v, 65535);
    itok_tab[(size_t)id * itok_cap + a] = (uint16_t)std::min(w, 65535);
  }
  itok_version++;
  return id;
}

static void compact_itok(const std::vector<uint16_t>& tab, uint32_t cap, uint32_t n, std::vector<uint16_t>& out) {
  out.assign((size_t)n * n, 0);
  for (uint32_t a = 0; a < n; a++) memcpy(&out[(size_t)a * n], &tab[(size_t)a * cap], (size_t)n * sizeof(uint16_t));
}

void HostIndex::sync_itok_dist() const {
  std::lock_guard<std::mutex> lk(itok_sync_mu);
  const uint64_t v = itok_version.load();
  if (itok_dist_version == v) return;
  compact_itok(itok_tab, itok_cap, (uint32_t)itok_strings.size(), itok_dist);
  itok_dist_version = v;
}

void HostIndex::snapshot_itok_dist(std::vector<uint16_t>& out, uint32_t& n, uint64_t& version) const {
  std::lock_guard<std::mutex> lk(itok_sync_mu);
  version = itok_version.load();
  n = (uint32_t)itok_strings.size();
  compact_itok(itok_tab, itok_cap, n, out);
}

uint32_t HostIndex::lookup_real(const std::string& s) const {
  auto it = real_intern.find(s);
  return it != real_intern.end() ? it->second : 0x7ffffffeu;
}

// The per-suffix records, signatures, keys and the n-gram tables.  The product builds them on the device
// (fmx_derive.hip.hh); this host version serves the CPU emulation of the kernels (tests/emul) and is the oracle the
// device build is compared with (tests/test_gpu_derive.py).
void HostIndex::derive_heavy() {
  if (heavy_ready) return;
  if (!sorted) throw std::runtime_error("derive: index is not sorted");
  const uint64_t n = suf_sid.size();
  suf_rec.resize(n);
  for (uint64_t i = 0; i < n; i++) {
    const uint32_t sp = sent_pos[suf_sid[i]];
    const uint32_t slen = buf[sp];
    SufRec r;
    r.off = sp + suf_pos[i];
    r.slen = (uint16_t)slen;
    r.rem = (uint16_t)(slen - (suf_pos[i] - 1u));
    suf_rec[i] = r;
  }
  {
    std::vector<uint64_t> sent_sig(sent_pos.size(), 0);
    for (size_t sdx = 0; sdx < sent_pos.size(); sdx++) {
      const uint32_t sp = sent_pos[sdx], len = buf[sp];
      uint64_t sg = 0;
      for (uint32_t k = 0; k < len; k++) sg |= 1ull << sig_bit(buf[sp + 1 + k]);
      sent_sig[sdx] = sg;
    }
    suf_sig.resize(n);
    suf_key.resize(n);
    suf_b.resize(n);
    for (uint64_t i = 0; i < n; i++) {
      SufRecB rb;
      rb.off = suf_rec[i].off; rb.rem = suf_rec[i].rem; rb.slen = suf_rec[i].slen;
      rb.prev = suf_rec[i].rem < suf_rec[i].slen ? buf[suf_rec[i].off - 1] : 0u;
      rb.sid = suf_sid[i];
      suf_b[i] = rb;
      suf_sig[i] = sig_pack(suf_rec[i].slen, sent_sig[suf_sid[i]]);
      SufKey k;
      for (uint32_t t = 0; t < kKeyTokens; t++) k.t[t] = t + 1 < suf_rec[i].rem ? buf[suf_rec[i].off + 1 + t] : 0u;
      suf_key[i] = k;
    }
  }
  {
    unsigned hw = std::thread::hardware_concurrency();
    build_gram_table((int)std::min<unsigned>(hw ? hw : 1, 32));
  }
  heavy_ready = true;
}

void HostIndex::derive() {
  if (!sorted) throw std::runtime_error("derive: index is not sorted");
  heavy_ready = false;
  suf_rec.clear(); suf_rec.shrink_to_fit(); suf_sig.clear(); suf_sig.shrink_to_fit(); suf_key.clear(); suf_key.shrink_to_fit();
  suf_b.clear(); suf_b.shrink_to_fit(); gram.clear(); gram.shrink_to_fit(); gram5.clear(); gram5.shrink_to_fit(); gram7.clear(); gram7.shrink_to_fit();
  // compute_idf_penalty / compute_max_idf_penalty — src/fuzzy_match.cc:367-390 (glibc logf / log)
  const unsigned N = num_sentences();
  idf.assign(vocab_size, 0.f);
  for (uint32_t w = 0; w < vocab_size; w++) {
    const uint32_t f = w < sfreq.size() ? sfreq[w] : 0;
    idf[w] = logf((float)N / (float)f);
  }
  if (vocab_size > 1) idf[1] = 0.f;  // VOCAB_UNK -> unknown_vocab_word_penalty = 0
  idf_max = (float)log((double)N);

  // real tokens / penalty tokens
  real_differs = false;
  real_ids.clear();
  itok_ids.clear();
  real_intern.clear();
  real_gen_++;
  if (!real.empty()) {
    for (uint32_t s = 0; s < real.size() && !real_differs; s++) {
      if (!real[s].itoks.empty()) { real_differs = true; break; }
      const auto parts = real[s].split();
      const uint32_t sp = sent_pos[s], len = buf[sp];
      if (parts.size() != len) { real_differs = true; break; }
      for (uint32_t i = 0; i < len; i++) {
        const uint32_t w = buf[sp + 1 + i];
        if (w >= forms.size() || forms[w] != parts[i]) { real_differs = true; break; }
      }
    }
  }
  if (real_differs) {
    real_ids.assign(buf.size(), 0);
    itok_ids.assign(buf.size(), 0);
    intern_itok(std::string());
    for (uint32_t s = 0; s < real.size(); s++) {
      const auto parts = real[s].split();
      const uint32_t sp = sent_pos[s], len = buf[sp];
      for (uint32_t i = 0; i < len; i++) {
        // out-of-range real tokens (trailing empty token dropped by the split) read as ""
        const std::string tok = i < parts.size() ? parts[i] : std::string();
        auto it = real_intern.find(tok);
        uint32_t id;
        if (it == real_intern.end()) { id = (uint32_t)real_intern.size(); real_intern.emplace(tok, id); }
        else id = it->second;
        // strchr("LUMC", tok[0]) — src/edit_distance.cc:57 ('\0' of an empty token matches too)
        const char c0 = tok.empty() ? '\0' : tok[0];
        const bool case_only = c0 == '\0' || c0 == 'L' || c0 == 'U' || c0 == 'M' || c0 == 'C';
        real_ids[sp + 1 + i] = id | (case_only ? 0x80000000u : 0u);
      }
      for (const auto& kv : real[s].itoks) {
        if (kv.first > len) continue;
        itok_ids[sp + kv.first] = intern_itok(kv.second);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// .fmi reader / writer — layout decoded in SURVEY §8(b); class-info records {u8 tracking; u32 version}
// appear once per C++ class at first use.
// ------------------------------------------------------------------------------------------

namespace {

class Reader {
 public:
  explicit Reader(std::vector<char>&& d) : d_(std::move(d)) {}
  template <class T> T get() {
    T v;
    need(sizeof(T));
    memcpy(&v, d_.data() + pos_, sizeof(T));
    pos_ += sizeof(T);
    return v;
  }
  void raw(void* dst, size_t n) { need(n); memcpy(dst, d_.data() + pos_, n); pos_ += n; }
  const char* span(uint64_t n) { need(n); const char* p = d_.data() + pos_; pos_ += n; return p; }  // n bytes, consumed
  std::string str() {
    const uint64_t n = get<uint64_t>();
    need(n);
    std::string s(d_.data() + pos_, (size_t)n);
    pos_ += n;
    return s;
  }
  uint32_t class_info() { get<uint8_t>(); return get<uint32_t>(); }  // tracking, version
  // a collection size: every element takes at least `min_bytes` of what is left of the file (a corrupt count must not size a vector)
  uint64_t count(uint64_t min_bytes) {
    const uint64_t n = get<uint64_t>();
    if (n > (d_.size() - pos_) / (min_bytes ? min_bytes : 1)) throw std::runtime_error("FMI file truncated or corrupt (collection size)");
    return n;
  }
  size_t pos() const { return pos_; }
  size_t size() const { return d_.size(); }
 private:
  void need(uint64_t n) const {
    if (n > d_.size() - pos_) throw std::runtime_error("FMI file truncated or corrupt");
  }
  std::vector<char> d_;
  size_t pos_ = 0;
};

class Writer {
 public:
  template <class T> void put(T v) { const char* p = (const char*)&v; d_.insert(d_.end(), p, p + sizeof(T)); }
  void raw(const void* p, size_t n) { d_.insert(d_.end(), (const char*)p, (const char*)p + n); }
  void str(const std::string& s) { put<uint64_t>(s.size()); raw(s.data(), s.size()); }
  void class_info(uint32_t version) { put<uint8_t>(0); put<uint32_t>(version); }
  char* grow(size_t n) { const size_t at = d_.size(); d_.resize(at + n); return d_.data() + at; }  // n more bytes, to be filled by the caller
  std::vector<char> d_;
};

template <class T> void read_pod_vector(Reader& r, std::vector<T>& v) {
  const uint64_t n = r.count(sizeof(T));
  v.resize(n);
  if (n) r.raw(v.data(), n * sizeof(T));
}

template <class T> void write_pod_vector(Writer& w, const std::vector<T>& v) {
  w.put<uint64_t>(v.size());
  if (!v.empty()) w.raw(v.data(), v.size() * sizeof(T));
}

}  // namespace

void HostIndex::load_fmi(const std::string& path) {
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs) throw std::runtime_error("cannot open FMI file: " + path);
  ifs.seekg(0, std::ios::end);
  const std::streamoff fsz = ifs.tellg();
  ifs.seekg(0);
  std::vector<char> data((size_t)fsz);
  if (fsz) ifs.read(data.data(), fsz);
  if (!ifs) throw std::runtime_error("cannot read FMI file: " + path);
  Reader r(std::move(data));

  char magic[4];
  r.raw(magic, 4);
  // the reference's magic check (fuzzy_matcher_binarization.cc:41) never fires; a wrong magic here is
  // reported like any other unreadable file
  if (memcmp(magic, "FMI", 3) != 0) throw std::runtime_error("invalid FMI file format: " + path);
  if (r.str() != "serialization::archive") throw std::runtime_error("FMI: not a Boost binary archive");
  const uint16_t lvt = r.get<uint16_t>();
  if (lvt < 10) throw std::runtime_error("FMI: unsupported Boost archive library version");
  fmi = FmiMeta();
  fmi.loaded = true;
  fmi.lvt = lvt;
  uint8_t sizes[4];
  r.raw(sizes, 4);
  if (sizes[0] != 4 || sizes[1] != 8 || sizes[2] != 4 || sizes[3] != 8 || r.get<uint32_t>() != 1)
    throw std::runtime_error("FMI: archive written on an incompatible platform");

  fmi.fm_version = r.class_info();     // FuzzyMatch
  pt = r.get<int32_t>();
  r.get<int16_t>();                    // class_id of SuffixArrayIndex (pointer)
  const uint32_t sai_version = r.class_info();
  fmi.sai_version = sai_version;
  r.get<uint32_t>();                   // object_id

  // VocabIndexer: forms, sfreq, form2index — vocab_indexer.hh:45-52
  r.class_info();                      // VocabIndexer
  r.class_info();                      // std::vector<std::string>
  forms.clear(); sfreq.clear(); form2index.clear();
  vocab_gen_.fetch_add(1, std::memory_order_release);
  {
    const uint64_t n = r.count(8);     // (a string is at least its length word)
    r.get<uint32_t>();                 // item_version
    forms.reserve(n);
    for (uint64_t i = 0; i < n; i++) forms.push_back(r.str());
  }
  read_pod_vector(r, sfreq);
  r.class_info();                      // std::unordered_map<std::string, unsigned>
  bool pair_su_seen = false;
  {
    const uint64_t n = r.count(12);
    fmi.map_buckets = r.get<uint64_t>();  // bucket_count
    r.get<uint32_t>();                 // item_version
    form2index.reserve(n);
    fmi.map_order.reserve(n);
    for (uint64_t i = 0; i < n; i++) {
      if (!pair_su_seen) { r.class_info(); pair_su_seen = true; }
      std::string k = r.str();
      const uint32_t v = r.get<uint32_t>();
      if (v >= forms.size()) throw std::runtime_error("FMI file corrupt (vocabulary map names an id beyond the vocabulary)");
      fmi.map_order.push_back(v);
      form2index.emplace(std::move(k), v);
    }
  }
  vocab_size = (uint32_t)forms.size();
  sfreq.resize(vocab_size, 0);
  vocab_gen_.fetch_add(1, std::memory_order_release);

  // SuffixArray — suffix_array.hxx:55-91
  const uint32_t sa_version = r.class_info();
  sorted = r.get<uint8_t>() != 0;
  suf_sid.clear(); suf_pos.clear();
  if (sa_version == 1) {
    r.class_info();                    // std::vector<SuffixView>
    const uint64_t n = r.count(6);
    r.get<uint32_t>();                 // item_version
    suf_sid.resize(n); suf_pos.resize(n);
    if (n) r.class_info();             // SuffixView
    if (n) {                           // packed {u32 sid; u16 pos} records: one bounds check, then a plain loop over the bytes
      const char* p = r.span(n * 6);
      for (uint64_t i = 0; i < n; i++, p += 6) { memcpy(&suf_sid[i], p, 4); memcpy(&suf_pos[i], p + 4, 2); }
    }
  } else if (sa_version == 0) {
    r.class_info();                    // std::vector<std::pair<unsigned, unsigned short>>
    const uint64_t n = r.count(8);
    suf_sid.resize(n); suf_pos.resize(n);
    for (uint64_t i = 0; i < n; i++) {
      suf_sid[i] = r.get<uint32_t>();
      suf_pos[i] = r.get<uint16_t>();
      r.get<uint16_t>();               // struct padding
    }
  } else
    throw std::invalid_argument("Unsupported FMI format");
  read_pod_vector(r, buf);
  read_pod_vector(r, sent_pos);
  read_pod_vector(r, qva);

  // _ids — std::vector<std::string> (class info already emitted)
  ids.clear();
  {
    const uint64_t n = r.count(8);
    r.get<uint32_t>();
    ids.reserve(n);
    for (uint64_t i = 0; i < n; i++) ids.push_back(r.str());
  }
  // _real_tokens — std::vector<Sentence>, sentence.hxx:23-30
  real.clear();
  {
    r.class_info();                    // std::vector<Sentence>
    const uint64_t n = r.count(8 + 8 + 8 + 4);  // tokstring length + map count + bucket count + item version
    r.get<uint32_t>();
    real.resize(n);
    fmi.itok_buckets.reserve(n);
    bool map_seen = false, pair_seen = false;
    for (uint64_t i = 0; i < n; i++) {
      if (i == 0) r.class_info();      // Sentence
      real[i].tokstring = r.str();
      if (!map_seen) { r.class_info(); map_seen = true; }  // std::unordered_map<size_t, std::string>
      const uint64_t m = r.count(16);
      fmi.itok_buckets.push_back(r.get<uint64_t>());  // bucket_count
      r.get<uint32_t>();               // item_version
      for (uint64_t k = 0; k < m; k++) {
        if (!pair_seen) { r.class_info(); pair_seen = true; }
        const uint64_t idx = r.get<uint64_t>();
        real[i].itoks.emplace_back(idx, r.str());
      }
    }
  }
  // suffix_array_index.hxx:55-56 — older files keep the default-constructed value
  max_tokens = 300;
  if (sai_version >= 1) max_tokens = r.get<uint64_t>();
  if (r.pos() != r.size()) throw std::runtime_error("FMI: trailing bytes after archive");

  // sanity: everything the kernels index must be in range
  const uint64_t nb = buf.size();
  for (uint32_t sp : sent_pos)
    if (sp >= nb || (uint64_t)sp + buf[sp] + 2 > nb) throw std::runtime_error("FMI file corrupt (sentence_pos)");
  for (uint64_t i = 0; i < suf_sid.size(); i++) {
    if (suf_sid[i] >= sent_pos.size()) throw std::runtime_error("FMI file corrupt (suffix sentence id)");
    const uint32_t len = buf[sent_pos[suf_sid[i]]];
    if (suf_pos[i] == 0 || suf_pos[i] > len) throw std::runtime_error("FMI file corrupt (suffix position)");
  }
  if (!qva.empty() && qva.back() != suf_sid.size() && !suf_sid.empty() && sorted)
    throw std::runtime_error("FMI file corrupt (quick vocab access)");
  // _quickVocabAccess names suffix ranges (src/suffix_array.cc:76-98): the probe indexes the suffix array with it
  for (size_t w = 0; w < qva.size(); w++)
    if (qva[w] > suf_sid.size() || (w && qva[w] < qva[w - 1])) throw std::runtime_error("FMI file corrupt (quick vocab access)");
  if (ids.size() != sent_pos.size() || real.size() != sent_pos.size())
    throw std::runtime_error("FMI file corrupt (ids / real tokens count)");
}

void HostIndex::save_fmi(const std::string& path) const {
  Writer w;
  // (one allocation for the bulk of a large index: suffixes, buffers, ~ (8 + 6 bytes per token) per Sentence record)
  w.d_.reserve(4096 + suf_sid.size() * 6 + (buf.size() + sent_pos.size() + qva.size()) * 4 + sent_pos.size() * 64 + buf.size() * 8);
  w.raw("FMI", 3);
  w.put<char>('1');                    // FuzzyMatch::version — src/fuzzy_match.cc:23
  w.str("serialization::archive");
  // a loaded file keeps its versions and map layout (FmiMeta) as long as nothing was added since
  const bool keep = fmi.loaded && fmi.map_order.size() == forms.size() && forms.size() >= vocab_size &&
                    fmi.itok_buckets.size() == sent_pos.size();
  const uint32_t sai_version = keep ? fmi.sai_version : 1u;
  w.put<uint16_t>(keep ? fmi.lvt : (uint16_t)14);  // 14 = the library version of the reference's fixtures
  const uint8_t sizes[4] = {4, 8, 4, 8};
  w.raw(sizes, 4);
  w.put<uint32_t>(1);
  w.class_info(keep ? fmi.fm_version : 1u);  // FuzzyMatch class version
  w.put<int32_t>(pt);
  w.put<int16_t>(1);                   // class_id
  w.put<uint8_t>(1);                   // tracking
  w.put<uint32_t>(sai_version);        // SuffixArrayIndex class version
  w.put<uint32_t>(0);                  // object_id
  w.class_info(0);                     // VocabIndexer
  w.class_info(0);                     // vector<string>
  // id-only corpora have no word forms: synthesise stable placeholder spellings so the file is loadable
  std::vector<std::string> f = forms;
  if (f.size() < vocab_size) {
    for (uint32_t k = (uint32_t)f.size(); k < vocab_size; k++) f.push_back("w" + std::to_string(k));
  }
  w.put<uint64_t>(f.size());
  w.put<uint32_t>(0);
  for (const auto& s : f) w.str(s);
  std::vector<uint32_t> sf = sfreq;
  sf.resize(f.size(), 0);
  write_pod_vector(w, sf);
  w.class_info(0);                     // unordered_map<string, unsigned>
  w.put<uint64_t>(f.size());
  w.put<uint64_t>(keep ? fmi.map_buckets : f.size() * 2 + 1);
  w.put<uint32_t>(0);
  for (uint32_t k = 0; k < f.size(); k++) {
    if (k == 0) w.class_info(0);       // pair<const string, unsigned>
    const uint32_t v = keep ? fmi.map_order[k] : k;  // any order loads: the reader rebuilds the map
    if (v >= f.size()) throw std::runtime_error("FMI: vocabulary map out of range");
    w.str(f[v]);
    w.put<uint32_t>(v);
  }
  w.class_info(1);                     // SuffixArray v1
  w.put<uint8_t>(sorted ? 1 : 0);
  w.class_info(0);                     // vector<SuffixView>
  w.put<uint64_t>(suf_sid.size());
  w.put<uint32_t>(0);
  if (!suf_sid.empty()) {
    w.class_info(0);                   // SuffixView
    char* p = w.grow(suf_sid.size() * 6);  // packed {u32 sid; u16 pos} records (suffix_array.hxx:55-91)
    for (uint64_t i = 0; i < suf_sid.size(); i++, p += 6) { memcpy(p, &suf_sid[i], 4); memcpy(p + 4, &suf_pos[i], 2); }
  }
  write_pod_vector(w, buf);
  write_pod_vector(w, sent_pos);
  write_pod_vector(w, qva);
  const uint64_t ns = sent_pos.size();
  w.put<uint64_t>(ns);
  w.put<uint32_t>(0);
  for (uint64_t i = 0; i < ns; i++) w.str(i < ids.size() ? ids[i] : std::to_string(i));
  w.class_info(0);                     // vector<Sentence>
  w.put<uint64_t>(ns);
  w.put<uint32_t>(0);
  bool pair_seen = false;
  for (uint64_t i = 0; i < ns; i++) {
    if (i == 0) w.class_info(0);       // Sentence
    if (i < real.size()) w.str(real[i].tokstring);
    else {
      // real tokens = normalised forms (Sentence(const Tokens&), src/sentence.cc:24): '\t'-joined, written in place
      const uint32_t sp = sent_pos[i], len = buf[sp];
      uint64_t bytes = len ? len - 1 : 0;
      for (uint32_t k = 0; k < len; k++) bytes += f[buf[sp + 1 + k]].size();
      w.put<uint64_t>(bytes);
      char* p = w.grow(bytes);
      for (uint32_t k = 0; k < len; k++) {
        if (k) *p++ = '\t';
        const std::string& t = f[buf[sp + 1 + k]];
        memcpy(p, t.data(), t.size());
        p += t.size();
      }
    }
    if (i == 0) w.class_info(0);       // unordered_map<size_t, string>
    const uint64_t m = i < real.size() ? real[i].itoks.size() : 0;
    w.put<uint64_t>(m);
    w.put<uint64_t>(keep ? fmi.itok_buckets[i] : (m ? m * 2 + 1 : 1));
    w.put<uint32_t>(0);
    for (uint64_t k = 0; k < m; k++) {
      if (!pair_seen) { w.class_info(0); pair_seen = true; }
      w.put<uint64_t>(real[i].itoks[k].first);
      w.str(real[i].itoks[k].second);
    }
  }
  if (sai_version >= 1) w.put<uint64_t>(max_tokens);  // suffix_array_index.hxx:33-43
  std::ofstream ofs(path, std::ios::binary);
  if (!ofs) throw std::runtime_error("cannot open FMI file for writing: " + path);
  ofs.write(w.d_.data(), (std::streamsize)w.d_.size());
  if (!ofs) throw std::runtime_error("cannot write FMI file: " + path);
}

}  // namespace fmx
