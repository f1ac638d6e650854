// fmx_capi.cc — the C ABI declared in include/fmx.h.  Thin: argument checks, exception -> status
// translation, and the host-side preparation the reference does in match() before its loops
// (src/fuzzy_match.cc:450-477).
#include "../../include/fmx.h"

#include <string.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "fmx_gpu.hh"
#include "fmx_host.hh"

using namespace fmx;

struct fmx_index {
  HostIndex                    host;
  std::unique_ptr<DeviceIndex> dev;
  bool                         finalized = false;
  std::mutex                   mu;  // serialises matching calls (one workspace per index) and interning
};

static thread_local std::string g_err;

static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

template <class F> static int guarded(F&& f) {
  try {
    return f();
  } catch (const GpuError& e) {
    return fail(FMX_E_NO_DEVICE, e.what());
  } catch (const std::bad_alloc&) {
    return fail(FMX_E_NOMEM, "out of host memory");
  } catch (const std::exception& e) {
    return fail(FMX_E_INVALID, e.what());
  }
}

extern "C" {

const char* fmx_last_error(void) { return g_err.c_str(); }
const char* fmx_version(void) { return "fmx 0.1 (gfx950)"; }
const char* fmx_stage_name(int stage) { return stage_name(stage); }

int fmx_device_count(void) { return gpu_device_count(); }

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

void fmx_index_free(fmx_index* ix) { delete ix; }

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
  int64_t r = -1;
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
    if (device >= 0) gpu_suffix_sort(ix->host, device);
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

int fmx_index_to_device(fmx_index* ix, int device) {
  if (!ix) return fail(FMX_E_INVALID, "null argument");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (ix->dev) return ix->dev->device() == device ? (int)FMX_OK : fail(FMX_E_STATE, "index already on another device");
  return guarded([&]() {
    std::unique_ptr<DeviceIndex> d(new DeviceIndex);
    d->upload(ix->host, device);
    ix->dev = std::move(d);
    return (int)FMX_OK;
  });
}

int fmx_vocab_lookup(const fmx_index* ix, const char* const* toks, uint64_t n, uint32_t* ids) {
  if (!ix || (n && (!toks || !ids))) return fail(FMX_E_INVALID, "null argument");
  return guarded([&]() {
    for (uint64_t i = 0; i < n; i++) ids[i] = ix->host.get_index(toks[i]);
    return (int)FMX_OK;
  });
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

int fmx_index_array(const fmx_index* ix, int which, const void** data, uint64_t* count) {
  if (!ix || !data || !count) return fail(FMX_E_INVALID, "null argument");
  const HostIndex& h = ix->host;
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
    case FMX_A_ITOK_DIST: *data = h.itok_dist.data(); *count = h.itok_dist.size(); break;
    case FMX_A_SUF_SIG: *data = h.suf_sig.data(); *count = h.suf_sig.size(); break;
    case FMX_A_SUF_KEY: *data = h.suf_key.data(); *count = h.suf_key.size(); break;
    case FMX_A_GRAM: *data = h.gram.data(); *count = h.gram.size(); break;
    case FMX_A_GRAM5: *data = h.gram5.data(); *count = h.gram5.size(); break;
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
  info->on_device = ix->dev != nullptr;
  info->device = ix->dev ? ix->dev->device() : -1;
  info->real_differs = h.real_differs;
  info->n_itok = (uint32_t)h.itok_strings.size();
  info->idf_max = h.idf_max;
  info->device_bytes = ix->dev ? ix->dev->bytes() : 0;
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
}

int fmx_match_batch(fmx_index* ix, const fmx_params* params, const fmx_batch* hb, fmx_match* out,
                    uint32_t* out_count, uint32_t* out_total, uint32_t cap, fmx_stats* stats) {
  if (!ix || !params || !hb || !out || !out_count) return fail(FMX_E_INVALID, "null argument");
  if (hb->nq && (!hb->q_off || (hb->q_off[hb->nq] && !hb->q_tok))) return fail(FMX_E_INVALID, "null batch arrays");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (!ix->dev) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  return guarded([&]() {
    std::lock_guard<std::mutex> lk(ix->mu);
    HostBatchView v;
    v.nq = hb->nq; v.q_off = hb->q_off; v.q_tok = hb->q_tok; v.q_real = hb->q_real; v.q_itok = hb->q_itok;
    std::vector<uint32_t> real_from_forms;
    if (ix->host.real_differs && !v.q_real && hb->nq) {
      // match(const Tokens&) builds `real` from the pattern itself (src/fuzzy_match.cc:428): the real
      // token of position j is the word form; unknown words never equal an indexed word id anyway.
      const uint64_t nt = hb->q_off[hb->nq];
      real_from_forms.resize(nt);
      for (uint64_t t = 0; t < nt; t++) {
        const uint32_t w = hb->q_tok[t];
        real_from_forms[t] = (w < ix->host.forms.size() && w != 1u) ? ix->host.lookup_real(ix->host.forms[w]) : 0x7ffffffeu;
      }
      v.q_real = real_from_forms.data();
    }
    Stats st;
    ix->dev->match_host(ix->host, make_params(ix, params), v, (Rec*)out, out_count, out_total, cap, &st);
    copy_stats(st, stats);
    return (int)FMX_OK;
  });
}

int fmx_match_batch_device(fmx_index* ix, const fmx_params* params, const fmx_batch* db, uint64_t total_tokens,
                           fmx_match* d_out, uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap,
                           void* stream, fmx_stats* stats) {
  (void)total_tokens;
  if (!ix || !params || !db || !d_out || !d_out_count) return fail(FMX_E_INVALID, "null argument");
  if (cap == 0) return fail(FMX_E_INVALID, "cap must be > 0");
  if (!ix->finalized) return fail(FMX_E_STATE, "index is not finalized");
  if (!ix->dev) return fail(FMX_E_NO_DEVICE, "index is not on a device (fmx_index_to_device); there is no CPU path");
  if (ix->host.real_differs && !db->q_real)
    return fail(FMX_E_INVALID, "this index has real tokens: device batches must carry q_real");
  if (db->nq == 0) return FMX_OK;
  return guarded([&]() {
    std::lock_guard<std::mutex> lk(ix->mu);
    DeviceBatchView v;
    v.nq = db->nq; v.q_off = db->q_off; v.q_tok = db->q_tok; v.q_real = db->q_real; v.q_itok = db->q_itok;
    v.p_max = db->max_len;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> bounds = ix->dev->sub_bounds_from_device(db->q_off, db->nq, &v.p_max, stream);
    if (v.p_max == 0) v.p_max = 1;
    Stats st;
    ix->dev->match_device(ix->host, make_params(ix, params), v, (Rec*)d_out, d_out_count, d_out_total, cap, stream,
                          bounds, &st);
    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    copy_stats(st, stats);
    return (int)FMX_OK;
  });
}

int fmx_set_option(fmx_index* ix, const char* name, int64_t value) {
  if (!ix || !name) return fail(FMX_E_INVALID, "null argument");
  if (!ix->dev) return fail(FMX_E_STATE, "index is not on a device");
  const std::string n(name);
  if (n == "sub_queries") ix->dev->set_sub_queries((uint32_t)std::max<int64_t>(value, 1));
  else if (n == "host_sub_queries") ix->dev->set_host_sub_queries((uint32_t)std::max<int64_t>(value, 1));
  else if (n == "initial_segment") ix->dev->set_initial_segment((uint64_t)std::max<int64_t>(value, 1024));
  else return fail(FMX_E_INVALID, "unknown option " + n);
  return FMX_OK;
}

}  // extern "C"
