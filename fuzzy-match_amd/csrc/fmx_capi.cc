// fmx_capi.cc — the C ABI declared in include/fmx.h.  Thin: argument checks, exception -> status
// translation, and the host-side preparation the reference does in match() before its loops
// (src/fuzzy_match.cc:450-477).
#include "../../include/fmx.h"

#include <cstdlib>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <algorithm>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "fmx_gpu.hh"
#include "fmx_host.hh"
#include "fmx_text.hh"

using namespace fmx;

struct Coalescer;

struct fmx_index {
  HostIndex                                 host;
  std::vector<std::unique_ptr<DeviceIndex>> devs;  // replicas of the frozen index, one per device
  bool                                      finalized = false;
  std::mutex                                mu;    // interning of penalty strings, lazy construction of the coalescer
  std::unique_ptr<Coalescer>                co;
  std::once_flag                            co_once;   // first fmx_match_one from any thread constructs the coalescer
  std::unique_ptr<SortResident, void (*)(SortResident*)> sort_resident{nullptr, gpu_sort_resident_free};  // what the GPU sort left in HBM
  int64_t                                   coalesce_us = 100;
  int64_t                                   host_lanes = 1;   // host threads (contexts) per device inside one fmx_match_batch call
  int64_t                                   min_block_queries = 32768;  // a replica / host lane gets a block only if it is at least this large
  bool                                      dup_devices = false;  // tests: the same ordinal may carry several replicas (option "duplicate_devices")
  std::atomic<uint32_t>                     rr{0};            // round-robin replica for batches too small to shard
  std::mutex                                pend_mu;          // the begun-and-not-ended calls of this index (fmx_pending handles):
  std::vector<fmx_pending*>                 pend;             // drained by fmx_index_free / fmx_index_reopen, so no call outlives its replica
  void drain_pending();
  ~fmx_index();
};

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

template <class F> static int guarded(F&& f) {
  try {
    return f();
  } catch (const UnsupportedError& e) {
    return fail(FMX_E_UNSUPPORTED, e.what());
  } catch (const GpuError& e) {
    return fail(FMX_E_NO_DEVICE, e.what());
  } catch (const std::invalid_argument& e) {
    return fail(FMX_E_INVALID, e.what());
  } catch (const std::bad_alloc&) {
    return fail(FMX_E_NOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return fail(FMX_E_INVALID, e.what());
  }
}

// The HIP runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues per device (four by default), and
// streams that share a queue run in order.  Several matching calls in flight bring two compute streams each: with eight
// queues six calls overlap (a 125 k-query pass 1.15 -> 0.97 ms, 100 k queries on the 1 M-sentence index 106 -> 129 M
// patterns/s; more queues than that lost again).  Only a default: a value in the environment wins, and it only takes
// effect when this library is loaded before the process's first HIP call (bench.py and the Python module set it before
// importing torch for the same reason).
__attribute__((constructor)) static void fmx_runtime_defaults() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" {

const char* fmx_last_error(void) { return g_err.c_str(); }
const char* fmx_version(void) { return "fmx 0.1 (gfx950)"; }
const char* fmx_stage_name(int stage) { return stage_name(stage); }

int fmx_device_count(void) { return gpu_device_count(); }
void* fmx_host_alloc(size_t bytes) { return gpu_host_alloc(bytes); }
void  fmx_host_free(void* p) { gpu_host_free(p); }

fmx_index* fmx_index_create(int32_t pt, uint64_t max_tokens_in_pattern) {
  try {
    auto* ix = new fmx_index;
    ix->host.pt = pt;
    ix->host.max_tokens = max_tokens_in_pattern;
    return ix;
  } catch (...) {
    g_err = "out of host memory";
    return nullptr;
  }
}

void fmx_index_free(fmx_index* ix) {
  if (ix) ix->drain_pending();  // calls begun and never ended: waited for and released before their replica goes
  delete ix;
}

fmx_index* fmx_index_load(const char* fmi_path, char* err, size_t errlen) {
  std::unique_ptr<fmx_index> ix;
  try {
    if (!fmi_path) throw std::runtime_error("null path");
    ix.reset(new fmx_index);
    ix->host.load_fmi(fmi_path);
    return ix.release();
  } catch (const std::exception& e) {
    g_err = e.what();
    if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = '\0'; }
    return nullptr;
  }
}

fmx_index* fmx_index_from_arrays(int32_t pt, uint64_t max_tokens_in_pattern, const uint32_t* suf_sid, const uint16_t* suf_pos,
                                 uint64_t n_suffixes, const uint32_t* sentence_buffer, uint64_t buffer_size,
                                 const uint32_t* sentence_pos, uint64_t n_sentences, const uint32_t* quick_vocab_access,
                                 uint64_t n_qva, const uint32_t* sfreq, uint32_t vocab_size, char* err, size_t errlen) {
  try {
    if ((n_suffixes && (!suf_sid || !suf_pos)) || (buffer_size && !sentence_buffer) || (n_sentences && !sentence_pos))
      throw std::runtime_error("null array");
    std::unique_ptr<fmx_index> ix(new fmx_index);
    ix->host.pt = pt;
    ix->host.max_tokens = max_tokens_in_pattern;
    ix->host.adopt_arrays(suf_sid, suf_pos, n_suffixes, sentence_buffer, buffer_size, sentence_pos, n_sentences,
                          quick_vocab_access, n_qva, sfreq, vocab_size);
    return ix.release();
  } catch (const std::exception& e) {
    g_err = e.what();
    if (err && errlen) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = '\0'; }
    return nullptr;
  }
}

int fmx_index_save(const fmx_index* ix, const char* fmi_path) {
  if (!ix || !fmi_path) return fail(FMX_E_INVALID, "null argument");
  try {
    ix->host.save_fmi(fmi_path);
    return FMX_OK;
  } catch (const std::exception& e) {
    return fail(FMX_E_IO, e.what());
  }
}

int64_t fmx_index_add_tm(fmx_index* ix, const char* id, const char* const* norm, uint32_t n_norm,
                         const char* real_tokstring, uint32_t n_itoks, const uint64_t* itok_idx,
                         const char* const* itok_str) {
  if (!ix || (n_norm && !norm)) return fail(FMX_E_INVALID, "null argument");
  if (ix->finalized) return fail(FMX_E_STATE, "index is finalized");
  int64_t r = FMX_TM_SKIPPED;
  int rc = guarded([&]() {
    std::vector<std::string> toks(norm, norm + n_norm);
    RealTok real;
    if (real_tokstring) real.tokstring = real_tokstring;
    else {
      // Sentence(const Tokens&) — src/sentence.cc:24: real tokens = the normalised tokens
      for (uint32_t i = 0; i < n_norm; i++) { if (i) real.tokstring += '\t'; real.tokstring += toks[i]; }
    }
    for (uint32_t k = 0; k < n_itoks; k++) real.itoks.emplace_back(itok_idx[k], std::string(itok_str[k]));
    if (!ix->host.ids.empty() || ix->host.sent_pos.empty()) {
      r = ix->host.add_tm(id ? id : "", real, toks);
      if (r < 0) r = FMX_TM_SKIPPED;  // ignored like the reference ignores it: not an error
    } else
      throw std::runtime_error("cannot mix id-only and string-level sentences in one index");
    return FMX_OK;
  });
  return rc == FMX_OK ? r : rc;
}

int fmx_index_add_ids(fmx_index* ix, const uint32_t* off, const uint32_t* toks, uint64_t n_sent, uint32_t vocab_size) {
  if (!ix || !off || (n_sent && !toks)) return fail(FMX_E_INVALID, "null argument");
  if (ix->finalized) return fail(FMX_E_STATE, "index is finalized");
  return guarded([&]() { ix->host.add_ids_bulk(off, toks, n_sent, vocab_size); return FMX_OK; });
}

int fmx_index_sort(fmx_index* ix, int device, int nthreads) {
  if (!ix) return fail(FMX_E_INVALID, "null argument");
  if (ix->finalized) return fail(FMX_E_STATE, "index is finalized");
  return guarded([&]() {
    if (ix->host.sorted) return (int)FMX_OK;
    if (device >= 0) ix->sort_resident.reset(gpu_suffix_sort(ix->host, device));
    else ix->host.sort(nthreads < 1 ? 1 : nthreads);
    return (int)FMX_OK;
  });
}

int fmx_index_finalize(fmx_index* ix) {
  if (!ix) return fail(FMX_E_INVALID, "null argument");
  if (ix->finalized) return FMX_OK;
  if (!ix->host.sorted) return fail(FMX_E_STATE, "index is not sorted");
  return guarded([&]() { ix->host.derive(); ix->finalized = true; return (int)FMX_OK; });
}

static int add_device(fmx_index* ix, int device) {
  if (!ix->dup_devices)
    for (auto& d : ix->devs) if (d->device() == device) return FMX_OK;
  return guarded([&]() {
    std::unique_ptr<DeviceIndex> d(new DeviceIndex);
    // the first replica on the device the suffix sort ran on adopts what the sort left in HBM
    std::unique_ptr<SortResident, void (*)(SortResident*)> res(ix->sort_resident.release(), gpu_sort_resident_free);
    d->upload(ix->host, device, res.get());
    d->set_contexts((uint32_t)std::max<int64_t>(ix->host_lanes + 1, 3));  // host_lanes may have been set before the upload
    ix->devs.push_back(std::move(d));
    return (int)FMX_OK;
  });
}

int fmx_index_to_device(fmx_index* ix, int device) {
  if (!ix) return fail(FMX_E_INVALID, "null argument");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (!ix->devs.empty())
    return ix->devs[0]->device() == device ? (int)FMX_OK : fail(FMX_E_STATE, "index already on another device (fmx_index_to_devices adds replicas)");
  return add_device(ix, device);
}

int fmx_index_to_devices(fmx_index* ix, const int* devices, int n_devices) {
  if (!ix || !devices || n_devices <= 0) return fail(FMX_E_INVALID, "null argument");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  for (int k = 0; k < n_devices; k++) {
    const int rc = add_device(ix, devices[k]);
    if (rc != FMX_OK) return rc;
  }
  return FMX_OK;
}

int fmx_set_devices(fmx_index* ix, const int* devices, int n_devices) { return fmx_index_to_devices(ix, devices, n_devices); }

int fmx_vocab_lookup(const fmx_index* ix, const char* const* toks, uint64_t n, uint32_t* ids) {
  if (!ix || (n && (!toks || !ids))) return fail(FMX_E_INVALID, "null argument");
  return guarded([&]() {
    for (uint64_t i = 0; i < n; i++) ids[i] = ix->host.get_index(toks[i]);
    return (int)FMX_OK;
  });
}

int fmx_vocab_lookup_mt(const fmx_index* ix, const char* const* toks, uint64_t n, uint32_t* ids, int nthreads) {
  if (!ix || (n && (!toks || !ids))) return fail(FMX_E_INVALID, "null argument");
  return guarded([&]() { ix->host.get_indices(toks, n, ids, nthreads < 1 ? 1 : nthreads); return (int)FMX_OK; });
}

int fmx_index_set_forms(fmx_index* ix, const char* const* spellings, uint64_t n) {
  if (!ix || (n && !spellings)) return fail(FMX_E_INVALID, "null argument");
  return guarded([&]() { ix->host.set_forms(spellings, n); return (int)FMX_OK; });
}

uint32_t fmx_index_lookup_real(const fmx_index* ix, const char* s, size_t len) {
  if (!ix || !s) return 0x7ffffffeu;
  return ix->host.lookup_real(std::string(s, len));
}

uint32_t fmx_index_intern_itok(fmx_index* ix, const char* s, size_t len) {
  if (!ix || !s) return 0;
  try {
    std::lock_guard<std::mutex> lk(ix->mu);
    return ix->host.intern_itok(std::string(s, len));
  } catch (const std::exception& e) {
    g_err = e.what();
    return 0xffffffffu;
  }
}

const char* fmx_index_sentence_id(const fmx_index* ix, uint32_t s_id) {
  if (!ix || s_id >= ix->host.ids.size()) return "";
  return ix->host.ids[s_id].c_str();
}

int fmx_index_device_array(const fmx_index* ix, int replica, int which, void* dst, uint64_t capacity_bytes, uint64_t* count) {
  if (!ix || !count) return fail(FMX_E_INVALID, "null argument");
  if (replica < 0 || (size_t)replica >= ix->devs.size()) return fail(FMX_E_NO_DEVICE, "no such replica");
  int k;
  switch (which) {
    case FMX_A_SUF_REC: k = 0; break;
    case FMX_A_SUF_SIG: k = 1; break;
    case FMX_A_SUF_KEY: k = 2; break;
    case FMX_A_GRAM: k = 3; break;
    case FMX_A_GRAM5: k = 4; break;
    case FMX_A_GRAM7: k = 8; break;
    case FMX_A_SUF_B: k = 5; break;
    case FMX_A_SENT_POS: k = 6; break;  // the device layout of the sentence buffer (DeviceIndex::upload): positions ...
    case FMX_A_BUFFER: k = 7; break;    // ... and words
    default: return fail(FMX_E_INVALID, "not a device-built array");
  }
  return guarded([&]() { *count = ix->devs[replica]->download(k, dst, capacity_bytes); return (int)FMX_OK; });
}

int fmx_index_array(const fmx_index* ix, int which, const void** data, uint64_t* count) {
  if (!ix || !data || !count) return fail(FMX_E_INVALID, "null argument");
  const HostIndex& h = ix->host;
  switch (which) {  // the accelerator arrays live in HBM only; their host twins (test oracle, CPU emulation) are built on request
    case FMX_A_SUF_REC: case FMX_A_SUF_SIG: case FMX_A_SUF_KEY: case FMX_A_GRAM: case FMX_A_GRAM5: case FMX_A_GRAM7: case FMX_A_SUF_B: {
      if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
      fmx_index* mix = const_cast<fmx_index*>(ix);
      std::lock_guard<std::mutex> lk(mix->mu);
      const int rc = guarded([&]() { mix->host.derive_heavy(); return (int)FMX_OK; });
      if (rc != FMX_OK) return rc;
      break;
    }
    default: break;
  }
  switch (which) {
    case FMX_A_SUF_SID: *data = h.suf_sid.data(); *count = h.suf_sid.size(); break;
    case FMX_A_SUF_POS: *data = h.suf_pos.data(); *count = h.suf_pos.size(); break;
    case FMX_A_BUFFER: *data = h.buf.data(); *count = h.buf.size(); break;
    case FMX_A_SENT_POS: *data = h.sent_pos.data(); *count = h.sent_pos.size(); break;
    case FMX_A_QVA: *data = h.qva.data(); *count = h.qva.size(); break;
    case FMX_A_SFREQ: *data = h.sfreq.data(); *count = h.sfreq.size(); break;
    case FMX_A_SUF_REC: *data = h.suf_rec.data(); *count = h.suf_rec.size(); break;
    case FMX_A_IDF: *data = h.idf.data(); *count = h.idf.size(); break;
    case FMX_A_REAL_IDS: *data = h.real_ids.data(); *count = h.real_ids.size(); break;
    case FMX_A_ITOK_IDS: *data = h.itok_ids.data(); *count = h.itok_ids.size(); break;
    case FMX_A_ITOK_DIST: h.sync_itok_dist(); *data = h.itok_dist.data(); *count = h.itok_dist.size(); break;
    case FMX_A_SUF_SIG: *data = h.suf_sig.data(); *count = h.suf_sig.size(); break;
    case FMX_A_SUF_KEY: *data = h.suf_key.data(); *count = h.suf_key.size(); break;
    case FMX_A_GRAM: *data = h.gram.data(); *count = h.gram.size(); break;
    case FMX_A_GRAM5: *data = h.gram5.data(); *count = h.gram5.size(); break;
    case FMX_A_GRAM7: *data = h.gram7.data(); *count = h.gram7.size(); break;
    case FMX_A_SUF_B: *data = h.suf_b.data(); *count = h.suf_b.size(); break;
    default: return fail(FMX_E_INVALID, "unknown array id");
  }
  return FMX_OK;
}

int fmx_index_get_info(const fmx_index* ix, fmx_index_info* info) {
  if (!ix || !info) return fail(FMX_E_INVALID, "null argument");
  const HostIndex& h = ix->host;
  memset(info, 0, sizeof(*info));
  info->n_sentences = h.sent_pos.size();
  info->n_suffixes = h.suf_sid.size();
  info->buffer_size = h.buf.size();
  info->vocab_size = h.vocab_size;
  info->max_tokens_in_pattern = h.max_tokens;
  info->pt = h.pt;
  info->sorted = h.sorted;
  info->finalized = ix->finalized;
  info->on_device = (int32_t)ix->devs.size();
  info->device = !ix->devs.empty() ? ix->devs[0]->device() : -1;
  info->real_differs = h.real_differs;
  info->n_itok = (uint32_t)h.itok_strings.size();
  info->idf_max = h.idf_max;
  info->device_bytes = !ix->devs.empty() ? ix->devs[0]->bytes() : 0;
  return FMX_OK;
}

// src/fuzzy_match.cc:451-452,472-477: contrast_buffer default and the idf normaliser
static Params make_params(const fmx_index* ix, const fmx_params* p) {
  Params prm;
  prm.fuzzy = p->fuzzy;
  prm.nmatch = p->nmatch;
  prm.no_perfect = p->no_perfect;
  prm.min_subseq_length = p->min_subseq_length;
  prm.min_subseq_ratio = p->min_subseq_ratio;
  prm.idf_penalty = p->idf_penalty;
  prm.ins = p->insert_cost; prm.del = p->delete_cost; prm.rep = p->replace_cost;
  prm.contrast = p->contrast;
  prm.reduce = p->reduce;
  prm.cbuf = p->contrast_buffer == -1 ? (int)p->nmatch : p->contrast_buffer;
  prm.idf_max = p->idf_penalty != 0.f ? ix->host.idf_max : 0.01f;
  prm.max_tokens = (uint32_t)std::min<uint64_t>(ix->host.max_tokens, 0xfffffffeu);
  return prm;
}

static void copy_stats(const Stats& s, fmx_stats* o) {
  if (!o) return;
  o->ms_total = s.ms_total; o->ms_h2d = s.ms_h2d; o->ms_d2h = s.ms_d2h;
  for (int i = 0; i < 16; i++) { o->ms_kernel[i] = s.ms_kernel[i]; o->launches[i] = s.launches[i]; }
  o->n_hits = s.n_hits; o->n_kept = s.n_kept; o->n_candidates = s.n_candidates; o->n_dp = s.n_dp;
  o->n_retries = s.n_retries; o->n_big = s.n_big; o->n_phase_b = s.n_phase_b; o->max_kept = s.max_kept;
  o->n_heavy_hits = s.n_heavy_hits; o->n_heavy_pass = s.n_heavy_pass; o->pinned = s.pinned;
  o->plane_words = s.plane_words; o->probe_bytes = s.probe_bytes; o->probe_lines = s.probe_lines;
  o->n_light_hits = s.n_light_hits; o->n_small = s.n_small; o->n_rerank = s.n_rerank;
}

// match(const Tokens&) builds `real` from the pattern itself (src/fuzzy_match.cc:428): the real token of position
// j is the word form; unknown words never equal an indexed word id anyway.
static void real_from_forms(const fmx_index* ix, const uint32_t* wids, uint64_t n, uint32_t* out) {
  for (uint64_t t = 0; t < n; t++) {
    const uint32_t w = wids[t];
    out[t] = (w < ix->host.forms.size() && w != 1u) ? ix->host.lookup_real(ix->host.forms[w]) : 0x7ffffffeu;
  }
}

// One batch over every replica of the index: contiguous query blocks, one per device, run concurrently; each
// block's records land directly in the caller's buffers (index replicated, no data-path exchange).
// A device's block is cut once more into `host_lanes` contiguous parts, each driven by its own host thread through
// its own context (workspace + streams): while one part sits in a kernel's tail or waits for a copy, the other's
// kernels fill the GPU, so a single call keeps the device as busy as several concurrent callers would.
static void match_on_devices(fmx_index* ix, const Params& prm, const HostBatchView& v, fmx_match* out, uint32_t* out_count,
                             uint32_t* out_total, uint32_t cap, Stats* st, fmx_match* d_keep = nullptr,
                             uint32_t* d_keep_count = nullptr) {
  size_t nd = d_keep ? 1 : ix->devs.size();  // a device-resident copy lives on one device: no sharding over replicas
  const uint64_t min_block = (uint64_t)std::max<int64_t>(ix->min_block_queries, 1);
  // a block must be worth a pass of its own (and a host thread): small batches — the coalescer's micro-batches above all — run
  // whole on ONE replica, chosen round-robin, instead of paying nd launches and synchronisations for a few queries each
  while (nd > 1 && v.nq / nd < min_block) nd--;
  size_t lanes = (size_t)std::max<int64_t>(ix->host_lanes, 1);
  while (lanes > 1 && v.nq / (nd * lanes) < min_block) lanes--;
  const size_t nb = nd * lanes;
  if (nb == 1 || v.nq < 2 * nb) {
    const size_t k = (d_keep || ix->devs.size() == 1) ? 0 : ix->rr.fetch_add(1) % ix->devs.size();
    ix->devs[k]->match_host(ix->host, prm, v, (Rec*)out, out_count, out_total, cap, st, (Rec*)d_keep, d_keep_count);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<Stats> sts(nb);
  std::vector<std::exception_ptr> errs(nb);
  auto run = [&](size_t k) {
    try {
      const uint64_t lo = v.nq * k / nb, hi = v.nq * (k + 1) / nb;
      HostBatchView s = v;
      s.nq = hi - lo;
      s.q_off = v.q_off + lo;                            // absolute offsets: the block's tokens are q_tok[q_off[lo] .. q_off[hi])
      if (v.q_itok) s.q_itok = v.q_itok + lo;            // slot of (query, j) = q_off[q] + q + j: the block's view starts lo slots in
      ix->devs[k / lanes]->match_host(ix->host, prm, s, (Rec*)(out + lo * cap), out_count + lo, out_total ? out_total + lo : nullptr, cap, &sts[k],
                                      d_keep ? (Rec*)(d_keep + lo * cap) : nullptr, d_keep_count ? d_keep_count + lo : nullptr);
    } catch (...) {
      errs[k] = std::current_exception();
    }
  };
  std::vector<std::thread> th;
  for (size_t k = 1; k < nb; k++) th.emplace_back(run, k);
  run(0);
  for (auto& t : th) t.join();
  for (auto& e : errs) if (e) std::rethrow_exception(e);
  if (st) {
    for (size_t k = 0; k < nb; k++) {
      const Stats& s = sts[k];
      for (int i = 0; i < 16; i++) { st->ms_kernel[i] = std::max(st->ms_kernel[i], s.ms_kernel[i]); st->launches[i] += s.launches[i]; }
      st->ms_h2d = std::max(st->ms_h2d, s.ms_h2d); st->ms_d2h = std::max(st->ms_d2h, s.ms_d2h);
      st->n_hits += s.n_hits; st->n_kept += s.n_kept; st->n_candidates += s.n_candidates; st->n_dp += s.n_dp;
      st->n_retries += s.n_retries; st->n_big += s.n_big; st->n_phase_b += s.n_phase_b; st->max_kept = std::max(st->max_kept, s.max_kept);
      st->n_heavy_hits += s.n_heavy_hits; st->n_heavy_pass += s.n_heavy_pass; st->pinned = s.pinned;
      st->plane_words += s.plane_words; st->probe_bytes += s.probe_bytes; st->probe_lines += s.probe_lines;
      st->n_light_hits += s.n_light_hits; st->n_small += s.n_small; st->n_rerank += s.n_rerank;
    }
    st->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
}

int fmx_match_batch(fmx_index* ix, const fmx_params* params, const fmx_batch* hb, fmx_match* out,
                    uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_stats* stats) {
  return fmx_match_batch_keep(ix, params, hb, out, out_count, out_total, cap, nullptr, nullptr, stats);
}

int fmx_match_batch_keep(fmx_index* ix, const fmx_params* params, const fmx_batch* hb, fmx_match* out,
                         uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_match* d_keep,
                         uint32_t* d_keep_count, fmx_stats* stats) {
  if (!ix || !params || !hb || !out || !out_count) return fail(FMX_E_INVALID, "null argument");
  if (hb->nq && (!hb->q_off || (hb->q_off[hb->nq] && !hb->q_tok))) return fail(FMX_E_INVALID, "null batch arrays");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  // re-entrant: no index-wide lock; every call works in its own context (fmx_gpu.hip)
  return guarded([&]() {
    HostBatchView v;
    v.nq = hb->nq; v.q_off = hb->q_off; v.q_tok = hb->q_tok; v.q_real = hb->q_real; v.q_itok = hb->q_itok;
    std::vector<uint32_t> forms_real;
    if (ix->host.real_differs && !v.q_real && hb->nq) {
      const uint64_t nt = hb->q_off[hb->nq];
      forms_real.resize(nt);
      real_from_forms(ix, hb->q_tok, nt, forms_real.data());
      v.q_real = forms_real.data();
    }
    Stats st;
    match_on_devices(ix, make_params(ix, params), v, out, out_count, out_total, cap, &st, d_keep, d_keep_count);
    copy_stats(st, stats);
    return (int)FMX_OK;
  });
}

// the (slot, id) pairs of a text batch index the penalty-string distance table on the device: ids the index never interned and slots
// beyond the batch's tokens (+ one trailing slot per pattern) are refused here, on the host (they are few)
static const char* text_batch_itoks_invalid(fmx_index* ix, const fmx_text_batch* tb) {
  if (!tb->n_itoks || !ix->host.real_differs) return nullptr;
  if (tb->nq == 0 || !tb->q_off) return "text batch: penalty tokens without patterns";  // (nq == 0 may come with q_off == NULL)
  uint64_t n_known;
  { std::lock_guard<std::mutex> g(ix->host.itok_sync_mu); n_known = ix->host.itok_strings.size(); }
  const uint64_t n_slots = (uint64_t)tb->q_off[tb->nq] + tb->nq;
  for (uint64_t k = 0; k < tb->n_itoks; k++) {
    if (tb->itok_id[k] >= n_known) return "text batch: a penalty-token id the index has not interned (fmx_intern_itok)";
    if (tb->itok_slot[k] >= n_slots) return "text batch: a penalty-token slot beyond the batch (slots = tokens + one trailing slot per pattern)";
  }
  return nullptr;
}

int fmx_match_batch_text(fmx_index* ix, const fmx_params* params, const fmx_text_batch* tb, fmx_match* out,
                         uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_stats* stats) {
  return fmx_match_batch_text_keep(ix, params, tb, out, out_count, out_total, cap, nullptr, nullptr, stats);
}

int fmx_match_batch_text_keep(fmx_index* ix, const fmx_params* params, const fmx_text_batch* tb, fmx_match* out,
                              uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_match* d_keep, uint32_t* d_keep_count,
                              fmx_stats* stats) {
  if (!ix || !params || !tb || !out || !out_count) return fail(FMX_E_INVALID, "null argument");
  if (tb->nq && (!tb->q_off || (tb->q_off[tb->nq] && !tb->text))) return fail(FMX_E_INVALID, "null batch arrays");
  if (tb->nq && tb->q_off[0] != 0) return fail(FMX_E_INVALID, "text batch: offsets must start at 0");
  // the text must hold exactly q_off[nq] spellings; every spelling takes at least its terminator
  if (tb->nq && tb->q_off[tb->nq] > tb->n_bytes) return fail(FMX_E_INVALID, "text batch: fewer bytes than spellings counted by the offsets");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  if (tb->nq && tb->real_text && tb->q_off[tb->nq] > tb->n_real_bytes) return fail(FMX_E_INVALID, "text batch: fewer real-token bytes than spellings counted by the offsets");
  if (tb->n_itoks && (!tb->itok_slot || !tb->itok_id)) return fail(FMX_E_INVALID, "text batch: null penalty-token arrays");
  if (const char* why = text_batch_itoks_invalid(ix, tb)) return fail(FMX_E_INVALID, why);
  return guarded([&]() {
    HostBatchView v;
    v.nq = tb->nq; v.q_off = tb->q_off; v.text = tb->text; v.n_text = tb->n_bytes;
    if (ix->host.real_differs) {  // (an index whose real tokens are its normalised tokens has no use for them)
      v.real_text = tb->real_text; v.n_real_text = tb->real_text ? tb->n_real_bytes : 0;
      v.n_itoks = tb->n_itoks; v.itok_slot = tb->itok_slot; v.itok_id = tb->itok_id;
    }
    Stats st;
    // (one replica per call, round-robin: a block of a text batch would need the byte offset of its first spelling)
    // (a kept copy lives on the first replica's device, like fmx_match_batch_keep's)
    const size_t k = (ix->devs.size() == 1 || d_keep) ? 0 : ix->rr.fetch_add(1) % ix->devs.size();
    ix->devs[k]->match_host(ix->host, make_params(ix, params), v, (Rec*)out, out_count, out_total, cap, &st, (Rec*)d_keep, d_keep_count);
    copy_stats(st, stats);
    return (int)FMX_OK;
  });
}

// One handle type serves both asynchronous entry points; whichever _end receives it, the call is finished through the path it was
// begun on (a text handle handed to fmx_match_batch_device_end used to leak its context lease: round-4 advice).
struct fmx_pending {
  DeviceIndex*           dev = nullptr;
  DeviceIndex::Pending*  p = nullptr;   // a device-resident call (fmx_match_batch_device_begin)
  DeviceIndex::HostCall* hc = nullptr;  // a host-buffer call (fmx_match_batch_text_begin)
};

static void pending_register(fmx_index* ix, fmx_pending* h) {
  std::lock_guard<std::mutex> g(ix->pend_mu);
  ix->pend.push_back(h);
}
// true if the handle belongs to this index and is still pending (it is then removed from the list: the caller finishes it)
static bool pending_claim(fmx_index* ix, fmx_pending* h) {
  std::lock_guard<std::mutex> g(ix->pend_mu);
  for (size_t k = 0; k < ix->pend.size(); k++)
    if (ix->pend[k] == h) { ix->pend[k] = ix->pend.back(); ix->pend.pop_back(); return true; }
  return false;
}
// waits for the call and brings its records home; throws what the call throws
static void pending_finish(fmx_pending* h) {
  if (DeviceIndex::HostCall* hc = h->hc) { h->hc = nullptr; h->dev->match_host_end(hc); }
  if (DeviceIndex::Pending* p = h->p) { h->p = nullptr; h->dev->match_device_end(p); }
}
static int pending_end(fmx_index* ix, fmx_pending* pending) {
  if (!ix || !pending) return fail(FMX_E_INVALID, "null argument");
  if (!pending_claim(ix, pending)) return fail(FMX_E_INVALID, "not a pending call of this index (ended twice, or begun on another index)");
  std::unique_ptr<fmx_pending> h(pending);
  return guarded([&]() { pending_finish(h.get()); return (int)FMX_OK; });
}
void fmx_index::drain_pending() {
  std::vector<fmx_pending*> left;
  { std::lock_guard<std::mutex> g(pend_mu); left.swap(pend); }
  for (fmx_pending* h : left) {
    try { pending_finish(h); } catch (...) {}
    delete h;
  }
}

int fmx_match_batch_text_begin(fmx_index* ix, const fmx_params* params, const fmx_text_batch* tb, fmx_match* out, uint32_t* out_count,
                               uint32_t* out_total, uint32_t cap, fmx_match* d_keep, uint32_t* d_keep_count, fmx_pending** pending) {
  if (!ix || !params || !tb || !out || !out_count || !pending) return fail(FMX_E_INVALID, "null argument");
  *pending = nullptr;
  if (tb->nq && (!tb->q_off || (tb->q_off[tb->nq] && !tb->text))) return fail(FMX_E_INVALID, "null batch arrays");
  if (tb->nq && tb->q_off[0] != 0) return fail(FMX_E_INVALID, "text batch: offsets must start at 0");
  if (tb->nq && tb->q_off[tb->nq] > tb->n_bytes) return fail(FMX_E_INVALID, "text batch: fewer bytes than spellings counted by the offsets");
  if (tb->nq && tb->real_text && tb->q_off[tb->nq] > tb->n_real_bytes) return fail(FMX_E_INVALID, "text batch: fewer real-token bytes than spellings counted by the offsets");
  if (tb->n_itoks && (!tb->itok_slot || !tb->itok_id)) return fail(FMX_E_INVALID, "text batch: null penalty-token arrays");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  if (const char* why = text_batch_itoks_invalid(ix, tb)) return fail(FMX_E_INVALID, why);
  return guarded([&]() {
    std::unique_ptr<fmx_pending> hp(new fmx_pending);
    HostBatchView v;
    v.nq = tb->nq; v.q_off = tb->q_off; v.text = tb->text; v.n_text = tb->n_bytes;
    if (ix->host.real_differs) {
      v.real_text = tb->real_text; v.n_real_text = tb->real_text ? tb->n_real_bytes : 0;
      v.n_itoks = tb->n_itoks; v.itok_slot = tb->itok_slot; v.itok_id = tb->itok_id;
    }
    hp->dev = ix->devs[0].get();  // (pending calls of one thread share the first replica: their contexts overlap on one GPU)
    hp->hc = hp->dev->match_host_begin(ix->host, make_params(ix, params), v, (Rec*)out, out_count, out_total, cap, (Rec*)d_keep, d_keep_count);
    pending_register(ix, hp.get());
    *pending = hp.release();
    return (int)FMX_OK;
  });
}

int fmx_match_batch_text_end(fmx_index* ix, fmx_pending* pending) { return pending_end(ix, pending); }

int fmx_match_batch_lines(fmx_index* ix, const fmx_params* params, const char* text, uint64_t n_bytes, uint64_t n_lines,
                          fmx_match* out, uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_stats* stats) {
  if (!ix || !params || !out || !out_count || (n_bytes && !text)) return fail(FMX_E_INVALID, "null argument");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (n_lines > n_bytes) return fail(FMX_E_INVALID, "line batch: more lines than bytes (every line ends with a line feed)");
  if (n_bytes && text[n_bytes - 1] != '\n') return fail(FMX_E_INVALID, "line batch: the text must end with a line feed");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  return guarded([&]() {
    Stats st;
    const size_t k = ix->devs.size() == 1 ? 0 : ix->rr.fetch_add(1) % ix->devs.size();
    ix->devs[k]->match_lines(ix->host, make_params(ix, params), text, n_bytes, n_lines, (Rec*)out, out_count, out_total, cap, &st);
    copy_stats(st, stats);
    return (int)FMX_OK;
  });
}

// Raw text lines: FuzzyMatch::match(const std::string&, ...) (src/fuzzy_match.cc:393-412) for a block of '\n'-terminated lines, the
// reference CLI's only input (cli/src/FuzzyMatch-cli.cc:219-233).  Lines of ASCII letters, digits, punctuation and blanks are
// tokenised ON THE DEVICE (fmx_vocab.hip.hh: k_raw_emit — the aggressive tokenizer is a byte-class machine on that alphabet) when the
// index's penalty tokens are ones that machine implements (tag, nbr, cas; pct / sep / jnr turn punctuation and joiners into penalty
// tokens: every line then takes the host path); every other line — UTF-8, placeholders, control characters — is tokenised by the
// host front-end (fmx_text.cc: NFC, scripts, placeholders, case mapping) and answered through an id batch with its real tokens and
// penalty tokens.  The rows of those lines overwrite whatever the device made of them.
int fmx_match_batch_raw_lines(fmx_index* ix, const fmx_params* params, const char* text, uint64_t n_bytes, uint64_t n_lines,
                              fmx_match* out, uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_stats* stats) {
  if (!ix || !params || !out || !out_count || (n_bytes && !text)) return fail(FMX_E_INVALID, "null argument");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (n_lines > n_bytes) return fail(FMX_E_INVALID, "line batch: more lines than bytes (every line ends with a line feed)");
  if (n_bytes && text[n_bytes - 1] != '\n') return fail(FMX_E_INVALID, "line batch: the text must end with a line feed");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  if (n_lines == 0) return FMX_OK;
  return guarded([&]() {
    const int pt = ix->host.pt;
    const bool device_tok = (pt & (kPtPct | kPtSep | kPtJnr)) == 0;
    Stats st;
    const size_t k = ix->devs.size() == 1 ? 0 : ix->rr.fetch_add(1) % ix->devs.size();
    DeviceIndex* dev = ix->devs[k].get();
    const Params prm = make_params(ix, params);
    std::vector<uint64_t> host_lines;
    if (device_tok) {
      RawLineTok tok;
      tok.cas = (pt & kPtCas) ? 1u : 0u;
      tok.nbr = (pt & kPtNbr) ? 1u : 0u;
      tok.num_id = ix->host.get_index(std::string("\xEF\xBD\x9F" "num" "\xEF\xBD\xA0"));  // ｟num｠ (src/fuzzy_match.cc:168)
      static const char* const kFeat[4] = {"L", "C", "U", "M"};
      for (int f = 0; f < 4; f++) tok.feat_real[f] = ix->host.real_differs ? ix->host.lookup_real(kFeat[f]) : 0x7ffffffeu;
      dev->match_raw_lines(ix->host, prm, text, n_bytes, n_lines, (Rec*)out, out_count, out_total, cap, &st, tok, host_lines);
    } else {
      host_lines.resize(n_lines);
      for (uint64_t l = 0; l < n_lines; l++) host_lines[l] = l;
    }
    if (!host_lines.empty()) {
      // where the lines start (only the flagged ones are looked at)
      std::vector<uint64_t> at;
      if (host_lines.size() == n_lines) {
        at.reserve(n_lines + 1);
        at.push_back(0);
        for (const char* p = text; p < text + n_bytes;) { const char* nl = (const char*)memchr(p, '\n', (size_t)(text + n_bytes - p)); p = nl + 1; at.push_back((uint64_t)(p - text)); }
        // (the device path refuses a wrong line count itself; here every line is the host's: same answer, before a worker indexes `at`)
        if (at.size() != n_lines + 1) throw std::invalid_argument("line batch: the text does not hold exactly n_lines line feeds");
      } else {
        at.resize(host_lines.size() * 2);
        uint64_t line = 0, want = 0;
        for (const char* p = text; p < text + n_bytes && want < host_lines.size();) {
          const char* nl = (const char*)memchr(p, '\n', (size_t)(text + n_bytes - p));
          if (line == host_lines[want]) { at[2 * want] = (uint64_t)(p - text); at[2 * want + 1] = (uint64_t)(nl - text); want++; }
          p = nl + 1; line++;
        }
      }
      const uint64_t nh = host_lines.size();
      std::vector<uint32_t> q_off(nh + 1, 0), q_tok, q_real, q_itok;
      std::vector<TextTokens> toks(nh);
      // the front-end on all usable CPUs (a line is independent of its neighbours)
      {
        const uint64_t nt = std::max<uint64_t>(1, std::min<uint64_t>(usable_cpus(), nh / 256 + 1));
        auto work = [&](uint64_t a, uint64_t b) {
          for (uint64_t i = a; i < b; i++) {
            const uint64_t b0 = host_lines.size() == n_lines ? at[i] : at[2 * i];
            const uint64_t b1 = host_lines.size() == n_lines ? at[i + 1] - 1 : at[2 * i + 1];
            tokenize_and_normalize(pt, std::string(text + b0, (size_t)(b1 - b0)), toks[i]);
          }
        };
        // (a worker's exception — bad_alloc, an ICU error in the general tokenizer — is carried to this thread, not to std::terminate)
        std::vector<std::exception_ptr> err(nt);
        auto guarded_work = [&](uint64_t t, uint64_t a, uint64_t b) { try { work(a, b); } catch (...) { err[t] = std::current_exception(); } };
        std::vector<std::thread> th;
        for (uint64_t t = 1; t < nt; t++) th.emplace_back(guarded_work, t, nh * t / nt, nh * (t + 1) / nt);
        guarded_work(0, 0, nh / nt);
        for (auto& t : th) t.join();
        for (auto& e : err) if (e) std::rethrow_exception(e);
      }
      const bool real = ix->host.real_differs;
      for (uint64_t i = 0; i < nh; i++) q_off[i + 1] = q_off[i] + (uint32_t)toks[i].norm.size();
      q_tok.resize(q_off[nh]);
      if (real) { q_real.resize(q_off[nh]); q_itok.assign((size_t)q_off[nh] + nh, 0u); }
      for (uint64_t i = 0; i < nh; i++) {
        const TextTokens& t = toks[i];
        for (size_t j = 0; j < t.norm.size(); j++) {
          q_tok[q_off[i] + j] = ix->host.get_index(t.norm[j]);
          if (real) q_real[q_off[i] + j] = ix->host.lookup_real(t.real[j]);
        }
        if (real)
          for (const auto& kv : t.itoks)
            if (kv.first <= t.norm.size()) q_itok[(size_t)q_off[i] + i + kv.first] = ix->host.intern_itok(kv.second);
      }
      std::vector<fmx_match> h_out(nh * cap);
      std::vector<uint32_t> h_cnt(nh, 0), h_tot(nh, 0);
      HostBatchView v;
      v.nq = nh; v.q_off = q_off.data(); v.q_tok = q_tok.empty() ? q_off.data() : q_tok.data();
      if (real) { v.q_real = q_real.empty() ? q_off.data() : q_real.data(); v.q_itok = q_itok.data(); }
      dev->match_host(ix->host, prm, v, (Rec*)h_out.data(), h_cnt.data(), h_tot.data(), cap, nullptr);
      for (uint64_t i = 0; i < nh; i++) {
        const uint64_t l = host_lines[i];
        out_count[l] = h_cnt[i];
        if (out_total) out_total[l] = h_tot[i];
        memcpy(out + l * cap, h_out.data() + i * cap, (size_t)h_cnt[i] * sizeof(fmx_match));
      }
    }
    copy_stats(st, stats);
    return (int)FMX_OK;
  });
}

// ---------------------------------------------------------------------------------------------
// Request coalescer behind fmx_match_one.  The reference's match() is called one pattern at a time from N
// threads; a GPU pass costs ~45 launches whatever its size, so concurrent requests with equal parameters are
// gathered into one micro-batch.  Dispatcher threads (one per context) take everything that is pending the moment
// they become free; an idle dispatcher waits at most coalesce_us after the first arrival for company.
// ---------------------------------------------------------------------------------------------
struct Request {
  fmx_params      prm;
  const uint32_t* wids; uint32_t p;
  const uint32_t* real; const uint32_t* itok;
  fmx_match*      out; uint32_t cap;
  uint32_t        count = 0, total = 0;
  int             status = FMX_OK;
  std::string     err;
  // the caller sleeps on its own condition variable (one shared by all callers made every wake-up queue for one mutex)
  std::mutex              m;
  std::condition_variable cv;
  bool                    done = false, lead = false;  // under m
  bool                    taken = false;               // under Coalescer::mu: a leader moved it from `pending` into its batch
};

// Flat combining: there are no dispatcher threads.  A caller that finds fewer than kLeaders batches in flight becomes a
// leader: it waits a moment for company (at most coalesce_us, less once as many requests as last time are waiting), takes
// every pending request with its own parameters, runs the batch on its own thread and wakes the others.  A request thus
// costs one sleep / wake-up for its followers (each on its own condition variable) and none for its leader — with dispatcher threads every batch paid two
// hand-overs (caller -> dispatcher -> caller), which, once a micro-batch is answered in ~60 us, was most of a call.
struct Coalescer {
  static constexpr int kLeaders = 2;  // batches in flight at once (their kernels overlap on the GPU)
  fmx_index*               ix;
  std::mutex               mu;
  std::condition_variable  cv_more;   // a gathering leader: another request arrived
  std::deque<Request*>     pending;
  int                      leaders = 0, gathering = 0;  // a leader that is still waiting for company takes every arrival
  uint32_t                 last_batch = 1;

  explicit Coalescer(fmx_index* ix_, int) : ix(ix_) {}

  void submit_and_wait(Request& r) {
    std::unique_lock<std::mutex> lk(mu);
    pending.push_back(&r);
    cv_more.notify_one();
    std::vector<Request*> batch;
    for (;;) {
      // A caller only leads while its own request is still pending.  Once another leader has taken it (possible while this
      // thread led a batch of other parameters, or between an offered seat and re-locking mu) it just waits for the answer:
      // leading on would keep it busy with other callers' batches long after its own result is complete.
      if (r.taken || leaders >= kLeaders || gathering != 0 || pending.empty()) {
        // sleep until a leader has answered this request, or hands this caller a free seat
        const bool taken = r.taken;
        lk.unlock();
        {
          std::unique_lock<std::mutex> rl(r.m);
          r.cv.wait(rl, [&]() { return r.done || (!taken && r.lead); });
          r.lead = false;
          if (r.done) return;
        }
        lk.lock();
        continue;
      }
      // ---- this caller leads a batch ----
      leaders++; gathering++;
      const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(ix->coalesce_us);
      while (pending.size() < last_batch && std::chrono::steady_clock::now() < deadline) cv_more.wait_until(lk, deadline);
      gathering--;
      batch.clear();
      if (!pending.empty()) {
        const fmx_params first = pending.front()->prm;
        const uint32_t cap = pending.front()->cap;
        for (auto it = pending.begin(); it != pending.end() && batch.size() < 65536;) {
          if (memcmp(&(*it)->prm, &first, sizeof(fmx_params)) == 0 && (*it)->cap == cap) { (*it)->taken = true; batch.push_back(*it); it = pending.erase(it); }
          else ++it;
        }
        last_batch = (uint32_t)batch.size();
        if (!pending.empty() && leaders < kLeaders) offer_seat();  // other parameter sets: someone else may lead them
      }
      bool mine = false;
      if (!batch.empty()) {
        lk.unlock();
        int status = FMX_OK;
        std::string err;
        run(batch, status, err);
        for (Request* q : batch) {
          if (q == &r) { mine = true; r.status = status; if (status != FMX_OK) r.err = err; continue; }
          std::lock_guard<std::mutex> ql(q->m);  // notified under its mutex: the request lives on its caller's stack
          q->status = status;
          if (status != FMX_OK) q->err = err;
          q->done = true;
          q->cv.notify_one();
        }
        lk.lock();
      }
      leaders--;
      if (!pending.empty()) offer_seat();
      if (mine) return;
    }
  }

  // under mu: wakes the owner of the oldest pending request so that it leads the next batch
  void offer_seat() {
    Request* q = pending.front();
    std::lock_guard<std::mutex> ql(q->m);
    q->lead = true;
    q->cv.notify_one();
  }

  // staging buffers of the calling thread (the one-launch path copies them into its pinned block; larger batches are rare)
  struct Staging { std::vector<uint32_t> off, tok, real, itok, cnt, tot; std::vector<fmx_match> out; };

  void run(const std::vector<Request*>& batch, int& status, std::string& err) {
    static thread_local Staging st;
    const uint64_t nq = batch.size();
    uint64_t nt = 0;
    const bool any_real = ix->host.real_differs;
    bool any_itok = false;
    for (auto* r : batch) { nt += r->p; any_itok = any_itok || r->itok; }
    const uint32_t cap = batch[0]->cap;
    try {
      st.off.resize(nq + 1); st.tok.resize(nt + 1); st.cnt.resize(nq); st.tot.resize(nq); st.out.resize(nq * cap);
      if (any_real) st.real.resize(nt + 1);
      if (any_itok) st.itok.resize(nt + nq + 1);
    } catch (const std::exception&) { status = FMX_E_NOMEM; err = "out of host memory"; return; }
    uint32_t* q_real = any_real ? st.real.data() : nullptr;
    uint32_t* q_itok = any_itok ? st.itok.data() : nullptr;
    uint32_t o = 0;
    for (uint64_t q = 0; q < nq; q++) {
      const Request& r = *batch[q];
      st.off[q] = o;
      memcpy(st.tok.data() + o, r.wids, (size_t)r.p * 4);
      if (q_real) { if (r.real) memcpy(q_real + o, r.real, (size_t)r.p * 4); else real_from_forms(ix, r.wids, r.p, q_real + o); }
      if (q_itok) { if (r.itok) memcpy(q_itok + o + q, r.itok, (size_t)(r.p + 1) * 4); else memset(q_itok + o + q, 0, (size_t)(r.p + 1) * 4); }
      o += r.p;
    }
    st.off[nq] = o;
    status = guarded([&]() {
      HostBatchView v;
      v.nq = nq; v.q_off = st.off.data(); v.q_tok = st.tok.data(); v.q_real = q_real; v.q_itok = q_itok;
      match_on_devices(ix, make_params(ix, &batch[0]->prm), v, st.out.data(), st.cnt.data(), st.tot.data(), cap, nullptr);
      return (int)FMX_OK;
    });
    if (status != FMX_OK) { err = g_err; return; }
    for (uint64_t q = 0; q < nq; q++) {  // results into the callers' buffers (they sleep until `done`)
      Request& r = *batch[q];
      r.count = st.cnt[q]; r.total = st.tot[q];
      memcpy(r.out, st.out.data() + q * cap, (size_t)st.cnt[q] * sizeof(fmx_match));
    }
  }

};

fmx_index::~fmx_index() { co.reset(); }

int fmx_match_one(fmx_index* ix, const fmx_params* params, const uint32_t* wids, uint32_t p, const uint32_t* q_real,
                  const uint32_t* q_itok, fmx_match* out, uint32_t cap, uint32_t* out_count, uint32_t* out_total) {
  if (!ix || !params || !out || !out_count || (p && !wids)) return fail(FMX_E_INVALID, "null argument");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  try {
    std::call_once(ix->co_once, [&]() { ix->co.reset(new Coalescer(ix, 2)); });
  } catch (const std::exception& e) { return fail(FMX_E_NOMEM, e.what()); }
  Request r;
  r.prm = *params; r.wids = wids; r.p = p; r.real = q_real; r.itok = q_itok; r.out = out; r.cap = cap;
  ix->co->submit_and_wait(r);
  if (r.status != FMX_OK) return fail(r.status, r.err);
  *out_count = r.count;
  if (out_total) *out_total = r.total;
  return FMX_OK;
}

// match() with entries already in the caller's vector and a contrastive factor (src/fuzzy_match.cc:622-650): not coalesced
// (every caller brings its own prior matches); one query through the pipeline with the priors as a side input.
int fmx_match_one_prior(fmx_index* ix, const fmx_params* params, const uint32_t* wids, uint32_t p, const uint32_t* q_real,
                        const uint32_t* q_itok, const uint32_t* prior_off, const uint32_t* prior_tok, uint32_t n_prior,
                        fmx_match* out, uint32_t cap, uint32_t* out_count, uint32_t* out_total) {
  return fmx_match_one_prior_ids(ix, params, wids, p, q_real, q_itok, prior_off, prior_tok, nullptr, n_prior, out, cap, out_count, out_total);
}

// ... with the prior entries' Match::s_id.  The reference memoises the cost of a (candidate, memory entry) pair under the two
// s_ids (src/fuzzy_match.cc:627, 637-650), so a memory entry whose id an earlier one carried is seen through the earlier one's
// cost.  Among the prior entries that is resolved here (the later entry's tokens are replaced by the first one's); a pick that
// carries a prior entry's id is resolved by the rerank kernel (select_body), which gets the ids.
int fmx_match_one_prior_ids(fmx_index* ix, const fmx_params* params, const uint32_t* wids, uint32_t p, const uint32_t* q_real,
                            const uint32_t* q_itok, const uint32_t* prior_off, const uint32_t* prior_tok, const uint32_t* prior_sid,
                            uint32_t n_prior, fmx_match* out, uint32_t cap, uint32_t* out_count, uint32_t* out_total) {
  if (!ix || !params || !out || !out_count || (p && !wids)) return fail(FMX_E_INVALID, "null argument");
  if (n_prior && (!prior_off || (prior_off[n_prior] && !prior_tok))) return fail(FMX_E_INVALID, "null prior arrays");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  return guarded([&]() {
    const uint32_t off[2] = {0, p};
    const uint32_t dummy = 0;
    HostBatchView v;
    v.nq = 1; v.q_off = off; v.q_tok = p ? wids : &dummy; v.q_real = q_real; v.q_itok = q_itok;
    std::vector<uint32_t> forms_real;
    if (ix->host.real_differs && !v.q_real && p) {
      forms_real.resize(p);
      real_from_forms(ix, wids, p, forms_real.data());
      v.q_real = forms_real.data();
    }
    std::vector<uint32_t> p_off(prior_off, prior_off + (n_prior ? n_prior + 1 : 0));
    std::vector<uint32_t> p_tok;
    if (n_prior) {  // rebase to 0: the staging copy starts at the first prior's tokens
      const uint32_t base = p_off[0];
      for (auto& o : p_off) o -= base;
      v.n_prior = n_prior; v.prior_off = p_off.data(); v.prior_tok = prior_tok + base;
      if (prior_sid) {
        v.prior_sid = prior_sid;
        bool shared = false;
        std::vector<uint32_t> first(n_prior);
        for (uint32_t r = 0; r < n_prior; r++) {
          first[r] = r;
          for (uint32_t k = 0; k < r; k++) if (prior_sid[k] == prior_sid[r]) { first[r] = k; shared = true; break; }
        }
        if (shared) {  // an entry whose id an earlier entry carries: the memo hands every candidate the earlier entry's cost
          std::vector<uint32_t> n_off(n_prior + 1, 0);
          for (uint32_t r = 0; r < n_prior; r++) {
            const uint32_t f = first[r];
            p_tok.insert(p_tok.end(), v.prior_tok + p_off[f], v.prior_tok + p_off[f + 1]);
            n_off[r + 1] = (uint32_t)p_tok.size();
          }
          if (p_tok.empty()) p_tok.push_back(0);
          p_off.swap(n_off);
          v.prior_off = p_off.data(); v.prior_tok = p_tok.data();
        }
      }
    }
    uint32_t total = 0;
    const size_t k = ix->devs.size() == 1 ? 0 : ix->rr.fetch_add(1) % ix->devs.size();
    ix->devs[k]->match_host(ix->host, make_params(ix, params), v, (Rec*)out, out_count, &total, cap, nullptr);
    if (out_total) *out_total = total;
    return (int)FMX_OK;
  });
}

int fmx_subsequence_batch(fmx_index* ix, const fmx_batch* hb, uint32_t number_of_matches, int32_t no_perfect,
                          int32_t min_subseq_length, float min_subseq_ratio, int32_t idf_weighting, fmx_match* out,
                          uint32_t* out_found, uint32_t* out_position) {
  if (!ix || !hb || !out || !out_found) return fail(FMX_E_INVALID, "null argument");
  if (hb->nq && (!hb->q_off || (hb->q_off[hb->nq] && !hb->q_tok))) return fail(FMX_E_INVALID, "null batch arrays");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  return guarded([&]() {
    HostBatchView v;
    v.nq = hb->nq; v.q_off = hb->q_off; v.q_tok = hb->q_tok; v.q_real = hb->q_real; v.q_itok = hb->q_itok;
    std::vector<uint32_t> forms_real;
    if (ix->host.real_differs && !v.q_real && hb->nq) {
      const uint64_t nt = hb->q_off[hb->nq];
      forms_real.resize(nt);
      real_from_forms(ix, hb->q_tok, nt, forms_real.data());
      v.q_real = forms_real.data();
    }
    ix->devs[0]->subsequence_host(ix->host, v, number_of_matches, no_perfect != 0, min_subseq_length, min_subseq_ratio,
                                  idf_weighting != 0, (Rec*)out, out_found, out_position);
    return (int)FMX_OK;
  });
}

struct fmx_text {
  int         pt = 0;
  TextTokens  t;
  std::string scratch;
};

fmx_text* fmx_text_tokenize(int32_t pt, const char* sentence, size_t len) {
  try {
    std::unique_ptr<fmx_text> h(new fmx_text);
    h->pt = pt;
    tokenize_and_normalize(pt, sentence ? std::string(sentence, len) : std::string(), h->t);
    return h.release();
  } catch (...) {
    g_err = "out of host memory";
    return nullptr;
  }
}
// the general tokenizer alone, without the fast path ASCII sentences take (tests compare the two)
fmx_text* fmx_text_tokenize_general(int32_t pt, const char* sentence, size_t len) {
  try {
    std::unique_ptr<fmx_text> h(new fmx_text);
    h->pt = pt;
    tokenize_and_normalize_general(pt, sentence ? std::string(sentence, len) : std::string(), h->t);
    return h.release();
  } catch (...) {
    g_err = "out of host memory";
    return nullptr;
  }
}
void        fmx_text_free(fmx_text* h) { delete h; }
uint32_t    fmx_text_num_tokens(const fmx_text* h) { return h ? (uint32_t)h->t.norm.size() : 0u; }
const char* fmx_text_norm(const fmx_text* h, uint32_t i) { return h && i < h->t.norm.size() ? h->t.norm[i].c_str() : ""; }
const char* fmx_text_real(const fmx_text* h, uint32_t i) { return h && i < h->t.real.size() ? h->t.real[i].c_str() : ""; }
uint32_t    fmx_text_num_itoks(const fmx_text* h) { return h ? (uint32_t)h->t.itoks.size() : 0u; }
uint64_t    fmx_text_itok_index(const fmx_text* h, uint32_t k) { return h && k < h->t.itoks.size() ? h->t.itoks[k].first : 0u; }
const char* fmx_text_itok(const fmx_text* h, uint32_t k) { return h && k < h->t.itoks.size() ? h->t.itoks[k].second.c_str() : ""; }
uint32_t    fmx_text_num_raw(const fmx_text* h) { return h ? (uint32_t)h->t.raw.size() : 0u; }
const char* fmx_text_raw(const fmx_text* h, uint32_t i) { return h && i < h->t.raw.size() ? h->t.raw[i].c_str() : ""; }
const char* fmx_text_feature(const fmx_text* h, uint32_t i) { return h && i < h->t.feat.size() ? h->t.feat[i].c_str() : ""; }
uint32_t    fmx_text_map(const fmx_text* h, uint32_t k) { return h && k < h->t.map_tokens.size() ? h->t.map_tokens[k] : 0u; }
const char* fmx_text_detokenize(fmx_text* h, uint32_t from, uint32_t to) {
  if (!h) return "";
  try {
    const uint32_t n = (uint32_t)h->t.map_tokens.size();
    const uint32_t a = from < n ? h->t.map_tokens[from] : (uint32_t)h->t.raw.size();
    const uint32_t b = to < n ? h->t.map_tokens[to] : (uint32_t)h->t.raw.size();
    std::vector<std::string> toks, feats;
    for (uint32_t i = a; i < b && i < h->t.raw.size(); i++) { toks.push_back(h->t.raw[i]); feats.push_back(h->t.feat[i]); }
    h->scratch = detokenize(h->pt, toks, feats);
    return h->scratch.c_str();
  } catch (...) {
    return "";
  }
}

int64_t fmx_index_add_text(fmx_index* ix, const char* id, const char* sentence, size_t len) {
  if (!ix || !sentence) return fail(FMX_E_INVALID, "null argument");
  if (ix->finalized) return fail(FMX_E_STATE, "index is finalized");
  int64_t r = FMX_TM_SKIPPED;
  int rc = guarded([&]() {
    TextTokens t;
    tokenize_and_normalize(ix->host.pt, std::string(sentence, len), t);
    if (t.norm.empty()) return (int)FMX_OK;  // "WARNING: cannot index empty segment" (src/fuzzy_match.cc:218-221)
    RealTok real;
    for (size_t i = 0; i < t.real.size(); i++) { if (i) real.tokstring += '\t'; real.tokstring += t.real[i]; }
    for (const auto& kv : t.itoks) real.itoks.emplace_back(kv.first, kv.second);
    if (!ix->host.ids.empty() || ix->host.sent_pos.empty()) {
      r = ix->host.add_tm(id ? id : "", real, t.norm);
      if (r < 0) r = FMX_TM_SKIPPED;
    } else
      throw std::runtime_error("cannot mix id-only and string-level sentences in one index");
    return (int)FMX_OK;
  });
  return rc == FMX_OK ? r : rc;
}

// FuzzyMatch::add_tm(id, sentence) (src/fuzzy_match.cc:213-224) for n sentences — the reference CLI's import loop is this call per
// line (cli/src/FuzzyMatch-cli.cc:32-79) — on every usable CPU, leaving the index byte for byte as n calls of fmx_index_add_text
// do.  Three phases per chunk of lines:
//   1. (parallel) _tokenize_and_normalize into packed storage per thread; every token is looked up in the vocabulary as it stands
//      (read-only here); a word it does not hold is noted with its first occurrence (line, position);
//   2. (one thread) the new words of all threads in the order of their first occurrence get their ids — the order
//      VocabIndexer::addWord (src/vocab_indexer.cc:37-50) would have met them in;
//   3. (parallel) the remaining tokens are resolved, and buffer, suffix enumeration, sentence frequencies (once per distinct
//      word of a sentence, src/vocab_indexer.cc:73-90), ids and real tokens are filled at offsets known from a prefix sum.
int64_t fmx_index_add_text_bulk(fmx_index* ix, const char* const* ids, const char* const* sentences, const size_t* lens, uint64_t n,
                                int64_t* sids_out) {
  if (!ix || (n && (!sentences || !lens))) return fail(FMX_E_INVALID, "null argument");
  if (ix->finalized) return fail(FMX_E_STATE, "index is finalized");
  int64_t added = 0;
  int rc = guarded([&]() {
    HostIndex& H = ix->host;
    if (!(!H.ids.empty() || H.sent_pos.empty())) throw std::runtime_error("cannot mix id-only and string-level sentences in one index");
    const int pt = H.pt;
    constexpr uint64_t kChunk = 1u << 18;  // lines per round of the three phases
    struct Line { uint64_t norm_at; uint32_t n_tok; uint64_t real_at, real_len; uint32_t itok_at, n_itok; bool take; };
    struct alignas(128) Part {  // what one thread made of its lines (a cache line of its own: the threads append side by side)
      std::string norm, real;
      std::vector<std::pair<uint64_t, std::string>> itoks;
      std::vector<uint32_t> wid;                       // per token: vocabulary id, or kUnset
      std::vector<Line> line;
      std::unordered_map<std::string, std::pair<uint64_t, uint32_t>> fresh;  // new word -> first occurrence (line, position)
      std::vector<uint32_t> freq;
    };
    constexpr uint32_t kUnset = 0xffffffffu;
    for (uint64_t c0 = 0; c0 < n; c0 += kChunk) {
      const uint64_t c1 = std::min(n, c0 + kChunk), m = c1 - c0;
      const uint64_t nt = std::max<uint64_t>(1, std::min<uint64_t>(usable_cpus(), m / 1024 + 1));
      std::vector<Part> parts(nt);
      auto range = [&](uint64_t t) { return std::make_pair(m * t / nt, m * (t + 1) / nt); };
      auto run = [&](const std::function<void(uint64_t)>& fn) {
        // (a worker's exception — bad_alloc on the packed strings or the per-thread frequency arrays, an ICU error in the general
        // tokenizer — is carried to this thread and on to guarded(), not to std::terminate)
        std::vector<std::exception_ptr> err(nt);
        auto safe = [&](uint64_t t) { try { fn(t); } catch (...) { err[t] = std::current_exception(); } };
        std::vector<std::thread> th;
        for (uint64_t t = 1; t < nt; t++) th.emplace_back(safe, t);
        safe(0);
        for (auto& t : th) t.join();
        for (auto& e : err) if (e) std::rethrow_exception(e);
      };
      static const bool timing = getenv("FMX_BULK_TIMING") != nullptr;  // design-time
      auto now = [] { return std::chrono::steady_clock::now(); };
      auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
      const auto tp0 = now();
      // ---- 1 ----
      run([&](uint64_t t) {
        Part& P = parts[t];
        const auto r = range(t);
        P.line.reserve(r.second - r.first);
        std::vector<std::pair<uint64_t, std::string>> it;
        std::string key;
        for (uint64_t i = r.first; i < r.second; i++) {
          const char* sp = sentences[c0 + i];
          Line L{};
          L.norm_at = P.norm.size(); L.real_at = P.real.size(); L.itok_at = (uint32_t)P.itoks.size();
          tokenize_and_normalize_packed(pt, sp ? sp : "", sp ? lens[c0 + i] : 0, P.norm, L.n_tok, P.real, it);
          L.real_len = P.real.size() - L.real_at;
          L.n_itok = (uint32_t)it.size();
          for (auto& kv : it) P.itoks.emplace_back(kv.first, std::move(kv.second));
          // FuzzyMatch::add_tm: an empty segment is not indexed (src/fuzzy_match.cc:218-221); SuffixArrayIndex::add_tm: nor is one
          // whose real sentence is empty or that is longer than the longest pattern (src/suffix_array_index.cc:16)
          L.take = L.n_tok != 0 && L.real_len != 0 && L.n_tok <= H.max_tokens && L.n_tok <= 65535;
          const char* w = P.norm.data() + L.norm_at;
          for (uint32_t k = 0; k < L.n_tok; k++) {
            const size_t wl = strlen(w);
            uint32_t id = kUnset;
            if (L.take) {
              key.assign(w, wl);
              auto f = H.form2index.find(key);
              if (f != H.form2index.end()) id = f->second;
              else if (P.fresh.find(key) == P.fresh.end()) P.fresh.emplace(key, std::make_pair(i, k));  // (the first occurrence inside this thread's lines; find first: emplace builds a node even when the key is there)
            }
            P.wid.push_back(id);
            w += wl + 1;
          }
          P.line.push_back(L);
        }
      });
      const auto tp1 = now();
      // where every thread's sentences and tokens go (the 2^32 limit is checked BEFORE the vocabulary changes: a refused block
      // leaves the index as it was)
      std::vector<uint64_t> s_base(nt + 1, 0), t_base(nt + 1, 0);
      for (uint64_t t = 0; t < nt; t++) {
        uint64_t ns = 0, ntk = 0;
        for (const Line& L : parts[t].line) if (L.take) { ns++; ntk += L.n_tok; }
        s_base[t + 1] = s_base[t] + ns; t_base[t + 1] = t_base[t] + ntk;
      }
      const uint64_t sent0 = H.sent_pos.size(), buf0 = H.buf.size(), suf0 = H.suf_sid.size();
      if (sent0 + s_base[nt] > 0xffffffffull || buf0 + t_base[nt] + 2 * s_base[nt] > 0xffffffffull)
        throw std::runtime_error("more than 2^32 - 1 sentences or buffer words");
      // ---- 2 ----
      {
        std::vector<std::tuple<uint64_t, uint32_t, const std::string*>> order;
        for (auto& P : parts) for (auto& kv : P.fresh) order.emplace_back(kv.second.first, kv.second.second, &kv.first);
        std::sort(order.begin(), order.end(), [](const auto& x, const auto& y) { return std::get<0>(x) != std::get<0>(y) ? std::get<0>(x) < std::get<0>(y) : std::get<1>(x) < std::get<1>(y); });
        for (auto& o : order) H.add_word(*std::get<2>(o));  // (a word several threads met: the earliest occurrence adds it, the others find it)
      }
      H.sent_pos.resize(sent0 + s_base[nt]);
      H.buf.resize(buf0 + t_base[nt] + 2 * s_base[nt]);
      H.suf_sid.resize(suf0 + t_base[nt]);
      H.suf_pos.resize(suf0 + t_base[nt]);
      H.ids.resize(sent0 + s_base[nt]);
      H.real.resize(sent0 + s_base[nt]);
      const uint32_t vsz = H.vocab_size;
      const auto tp2 = now();
      // ---- 3 ----
      run([&](uint64_t t) {
        Part& P = parts[t];
        const auto r = range(t);
        P.freq.assign(vsz, 0);
        std::vector<uint32_t> stamp(vsz, 0);
        uint32_t mark = 0;
        uint64_t sidx = sent0 + s_base[t], bp = buf0 + t_base[t] + 2 * s_base[t], sp = suf0 + t_base[t], tok_at = 0;
        std::string key;
        for (uint64_t i = r.first; i < r.second; i++) {
          const Line& L = P.line[i - r.first];
          if (!L.take) { if (sids_out) sids_out[c0 + i] = FMX_TM_SKIPPED; tok_at += L.n_tok; continue; }
          if (++mark == 0) { std::fill(stamp.begin(), stamp.end(), 0u); mark = 1; }
          H.sent_pos[sidx] = (uint32_t)bp;
          H.buf[bp++] = L.n_tok;
          const char* w = P.norm.data() + L.norm_at;
          for (uint32_t k = 0; k < L.n_tok; k++) {
            const size_t wl = strlen(w);
            uint32_t id = P.wid[tok_at + k];
            if (id == kUnset) { key.assign(w, wl); id = H.form2index.find(key)->second; }
            H.buf[bp++] = id;
            H.suf_sid[sp] = (uint32_t)sidx;
            H.suf_pos[sp] = (uint16_t)(k + 1);
            sp++;
            if (stamp[id] != mark) { stamp[id] = mark; P.freq[id]++; }
            w += wl + 1;
          }
          H.buf[bp++] = 0;  // VocabIndexer::SENTENCE_SEPARATOR
          H.ids[sidx] = ids && ids[c0 + i] ? ids[c0 + i] : "";
          RealTok& R = H.real[sidx];
          R.tokstring.assign(P.real.data() + L.real_at, L.real_len);
          R.itoks.clear();
          for (uint32_t k = 0; k < L.n_itok; k++) R.itoks.emplace_back(P.itoks[L.itok_at + k].first, P.itoks[L.itok_at + k].second);
          if (sids_out) sids_out[c0 + i] = (int64_t)sidx;
          sidx++;
          tok_at += L.n_tok;
        }
      });
      for (uint64_t t = 0; t < nt; t++)
        for (uint32_t w = 0; w < vsz; w++) H.sfreq[w] += parts[t].freq[w];
      if (s_base[nt]) H.sorted = false;
      added += (int64_t)s_base[nt];
      if (timing) fprintf(stderr, "[fmx] add_text_bulk: %llu lines on %llu threads: tokenise + look up %.1f ms, new words + resize %.1f ms, fill %.1f ms\n",
                          (unsigned long long)m, (unsigned long long)nt, ms(tp0, tp1), ms(tp1, tp2), ms(tp2, now()));
    }
    return (int)FMX_OK;
  });
  return rc == FMX_OK ? added : rc;
}

int fmx_index_reopen(fmx_index* ix) {
  if (!ix) return fail(FMX_E_INVALID, "null argument");
  // (the coalescer stays: it holds no device state, and it is constructed once per index — std::call_once)
  ix->drain_pending();
  ix->sort_resident.reset();
  ix->devs.clear();
  ix->finalized = false;
  return FMX_OK;
}

int fmx_match_batch_device(fmx_index* ix, const fmx_params* params, const fmx_batch* db, uint64_t total_tokens,
                           fmx_match* d_out, uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap,
                           void* stream, fmx_stats* stats) {
  if (!ix || !params || !db || !d_out || !d_out_count) return fail(FMX_E_INVALID, "null argument");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  if (ix->host.real_differs && !db->q_real)
    return fail(FMX_E_INVALID, "this index has real tokens: device batches must carry q_real");
  if (db->nq == 0) return FMX_OK;
  return guarded([&]() {
    DeviceIndex* dev = ix->devs[0].get();  // device buffers live on one device: the first replica's
    DeviceBatchView v;
    v.nq = db->nq; v.q_off = db->q_off; v.q_tok = db->q_tok; v.q_real = db->q_real; v.q_itok = db->q_itok;
    v.p_max = db->max_len;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> bounds;
    if (total_tokens != 0 && total_tokens <= 0xffffffffull && db->max_len != 0 && db->nq <= dev->sub_queries())
      bounds = {0u, (uint32_t)total_tokens};  // one pass, everything known: no device read, no synchronisation up front
    else
      bounds = dev->sub_bounds_from_device(db->q_off, db->nq, &v.p_max, stream);
    if (v.p_max == 0) v.p_max = 1;
    Stats st;
    // stats == NULL: the passes run back to back without per-stage events, and large calls issued from several host threads
    // (each on its own stream) are chained on the GPU like fmx_match_batch chains them
    dev->match_device(ix->host, make_params(ix, params), v, (Rec*)d_out, d_out_count, d_out_total, cap, stream,
                          bounds, stats ? &st : nullptr);
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    copy_stats(st, stats);
    return (int)FMX_OK;
  });
}


int fmx_match_batch_device_begin(fmx_index* ix, const fmx_params* params, const fmx_batch* db, uint64_t total_tokens,
                                 fmx_match* d_out, uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap, void* stream,
                                 fmx_pending** pending) {
  if (!ix || !params || !db || !d_out || !d_out_count || !pending) return fail(FMX_E_INVALID, "null argument");
  *pending = nullptr;
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->devs.empty()) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  if (ix->host.real_differs && !db->q_real)
    return fail(FMX_E_INVALID, "this index has real tokens: device batches must carry q_real");
  return guarded([&]() {
    std::unique_ptr<fmx_pending> h(new fmx_pending);
    if (db->nq == 0) { pending_register(ix, h.get()); *pending = h.release(); return (int)FMX_OK; }
    DeviceIndex* dev = ix->devs[0].get();
    DeviceBatchView v;
    v.nq = db->nq; v.q_off = db->q_off; v.q_tok = db->q_tok; v.q_real = db->q_real; v.q_itok = db->q_itok;
    v.p_max = db->max_len;
    std::vector<uint32_t> bounds;
    if (total_tokens != 0 && total_tokens <= 0xffffffffull && db->max_len != 0 && db->nq <= dev->sub_queries())
      bounds = {0u, (uint32_t)total_tokens};
    else
      bounds = dev->sub_bounds_from_device(db->q_off, db->nq, &v.p_max, stream);
    if (v.p_max == 0) v.p_max = 1;
    h->dev = dev;
    h->p = dev->match_device_begin(ix->host, make_params(ix, params), v, (Rec*)d_out, d_out_count, d_out_total, cap, stream, bounds);
    pending_register(ix, h.get());
    *pending = h.release();
    return (int)FMX_OK;
  });
}

int fmx_match_batch_device_end(fmx_index* ix, fmx_pending* pending) { return pending_end(ix, pending); }

int fmx_set_option(fmx_index* ix, const char* name, int64_t value) {
  if (!ix || !name) return fail(FMX_E_INVALID, "null argument");
  const std::string n(name);
  if (n == "coalesce_us") { ix->coalesce_us = std::max<int64_t>(value, 0); return FMX_OK; }
  if (n == "duplicate_devices") { ix->dup_devices = value != 0; return FMX_OK; }   // tests on a one-GPU box: several replicas on one ordinal
  if (n == "min_block_queries") { ix->min_block_queries = std::max<int64_t>(value, 1); return FMX_OK; }
  if (n == "host_lanes") {
    ix->host_lanes = std::min<int64_t>(std::max<int64_t>(value, 1), 8);
    for (auto& d : ix->devs) d->set_contexts((uint32_t)ix->host_lanes + 1);  // + one for a concurrent caller / the coalescer
    return FMX_OK;
  }
  if (ix->devs.empty()) return fail(FMX_E_STATE, "index is not on a device");
  for (auto& d : ix->devs) {
    if (n == "sub_queries") d->set_sub_queries((uint32_t)std::max<int64_t>(value, 1));
    else if (n == "host_sub_queries") d->set_host_sub_queries((uint32_t)std::max<int64_t>(value, 1));
    else if (n == "initial_segment") d->set_initial_segment((uint64_t)std::max<int64_t>(value, 1024));
    else if (n == "heavy_min") d->set_heavy_min((uint32_t)std::max<int64_t>(value, 0));
    else if (n == "contexts") d->set_contexts((uint32_t)std::max<int64_t>(value, 1));
    else if (n == "max_host_active") d->set_max_host_active((uint32_t)std::max<int64_t>(value, 1));
    else if (n == "probe_count") d->set_probe_count(value != 0);
    else if (n == "serial") d->set_serial(value != 0);
    else if (n == "pair_finish") d->set_pair_finish(value != 0);
    else if (n == "small_path") d->set_small_path(value != 0);
    else if (n == "real_hybrid") d->set_real_hybrid(value != 0);
    else if (n == "turns") d->set_turns((uint32_t)std::max<int64_t>(value, 0));
    else if (n == "device_turns") d->set_device_turns((uint32_t)std::max<int64_t>(value, 0));
    else if (n == "expand_order") d->set_expand_order((uint32_t)std::max<int64_t>(value, 0));
    else if (n == "b_grid") d->set_b_grid((uint32_t)std::max<int64_t>(value, 1));
    else if (n == "heavy_grid") d->set_heavy_grid((uint32_t)std::max<int64_t>(value, 1));
    else if (n == "light_max") d->set_light_max((uint32_t)std::max<int64_t>(value, 0));
    else if (n == "probe_split") d->set_probe_split(value != 0);
    else if (n == "light_flat") d->set_light_flat((uint32_t)std::max<int64_t>(value, 0));
    else return fail(FMX_E_INVALID, "unknown option " + n);
  }
  return FMX_OK;
}

}  // extern "C"
