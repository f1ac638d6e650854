// fmx_gpu.hh — host-visible interface of the device side (implemented in fmx_gpu.hip).
#pragma once
#include <stdint.h>
#include <stdexcept>
#include <string>
#include <vector>

#include "fmx_host.hh"

namespace fmx {

// spare zero words behind the device token buffer (and its per-slot twins): the edit-distance kernels read a sentence four tokens at a
// time past its end, the group rerank its first 32 token slots unconditionally (round 6: one dependent load level fewer)
constexpr uint64_t kTokSpare = 40;


struct GpuError : std::runtime_error {  // a HIP call failed / no usable device
  using std::runtime_error::runtime_error;
};
struct UnsupportedError : std::runtime_error {  // a documented limit of this build (not a device failure)
  using std::runtime_error::runtime_error;
};

constexpr int kNumStages = 13;  // eleven consecutive stages of the main stream + two of the side stream beside stages 3-5
constexpr int kStageExpandH = 11, kStageExpandB = 12;
const char* stage_name(int s);

struct Stats {
  double   ms_total = 0, ms_h2d = 0, ms_d2h = 0;
  double   ms_kernel[16] = {0};
  uint64_t launches[16] = {0};
  uint64_t n_hits = 0, n_kept = 0, n_candidates = 0, n_dp = 0, n_retries = 0, n_big = 0, n_phase_b = 0, max_kept = 0;
  uint64_t n_heavy_hits = 0, n_heavy_pass = 0, pinned = 0;
  uint64_t plane_words = 0, probe_bytes = 0, probe_lines = 0, n_light_hits = 0, n_small = 0, n_rerank = 0;
};

struct Rec;  // == fmx_match, defined in fmx_kernels.hip.hh

struct HostBatchView {
  uint64_t        nq = 0;
  const uint32_t* q_off = nullptr;
  const uint32_t* q_tok = nullptr;
  const uint32_t* q_real = nullptr;
  const uint32_t* q_itok = nullptr;
  // matches already in the caller's vector (contrastive rerank penalises against them, src/fuzzy_match.cc:622-650): CSR
  // token sequences, shared by every query of the batch
  uint32_t        n_prior = 0;
  const uint32_t* prior_off = nullptr;
  const uint32_t* prior_tok = nullptr;
  const uint32_t* prior_sid = nullptr;  // their Match::s_id, or null (the rerank's memo is keyed on ids: src/fuzzy_match.cc:637)
  // the patterns as token SPELLINGS instead of ids (q_tok == nullptr): q_off[nq] NUL-terminated spellings back to back; the
  // vocabulary ids are looked up on the device (fmx_vocab.hip.hh)
  const char*     text = nullptr;
  uint64_t        n_text = 0;
  // ... with the real tokens as a second stream of q_off[nq] spellings (nullptr: the pattern tokens' own spellings) and the
  // penalty tokens as (slot, interned id) pairs, slot = q_off[q] + q + j like q_itok — indexes with real tokens only
  const char*     real_text = nullptr;
  uint64_t        n_real_text = 0;
  uint64_t        n_itoks = 0;
  const uint64_t* itok_slot = nullptr;
  const uint32_t* itok_id = nullptr;
};

struct DeviceBatchView {
  uint64_t        nq = 0;
  const uint32_t* q_off = nullptr;
  const uint32_t* q_tok = nullptr;
  const uint32_t* q_real = nullptr;
  const uint32_t* q_itok = nullptr;
  uint32_t        p_max = 0;  // longest pattern of the batch (0 = unknown)
  uint32_t        n_prior = 0;        // prior matches (device CSR), see HostBatchView
  const uint32_t* prior_off = nullptr;
  const uint32_t* prior_tok = nullptr;
  const uint32_t* prior_sid = nullptr;
  uint32_t        prior_max_len = 0;  // longest prior match
};

int  gpu_device_count();
// What the GPU suffix sort leaves in HBM for the upload that follows: the token buffer, the sentence positions and the sorted
// suffix array, so that fmx_index_sort -> fmx_index_finalize -> fmx_index_to_device moves nothing across PCIe twice.
struct SortResident {
  int       device = -1;
  uint32_t* d_buf = nullptr;       // h.buf + 8 spare zero words (what DeviceIndex keeps as `tok`)
  uint32_t* d_sent_pos = nullptr;
  uint32_t* d_sid = nullptr;       // sorted suffix array, parallel arrays
  uint16_t* d_pos = nullptr;
  uint64_t  n_buf = 0, n_sent = 0, n_suf = 0;
};
void gpu_sort_resident_free(SortResident*);
// throws unless `device` is a HIP ordinal of a gfx950 part
void gpu_check_device(int device);
// sorts h's suffix array on `device`; returns what stayed in HBM (nullptr for an empty index); the caller owns it
SortResident* gpu_suffix_sort(HostIndex& h, int device);
void* gpu_host_alloc(size_t bytes);  // pinned host memory (hipHostMalloc), nullptr on failure
void  gpu_host_free(void* p);

// what the device's raw-text tokenizer needs to know about the index (fmx_match_batch_raw_lines)
struct RawLineTok {
  uint32_t cas = 0, nbr = 0;   // the index's penalty tokens that change a word (lower case + case feature) / a number (placeholder)
  uint32_t num_id = 1;         // vocabulary id of the number placeholder
  uint32_t feat_real[4] = {0x7ffffffeu, 0x7ffffffeu, 0x7ffffffeu, 0x7ffffffeu};  // real-token ids of the features L, C, U, M
};

class DeviceIndex {
 public:
  struct Impl;
  DeviceIndex();
  ~DeviceIndex();
  DeviceIndex(const DeviceIndex&) = delete;
  DeviceIndex& operator=(const DeviceIndex&) = delete;

  // resident (optional, consumed): arrays the suffix sort left on the same device; they are adopted instead of uploaded again
  void     upload(const HostIndex& h, int device, SortResident* resident = nullptr);
  int      device() const;
  // test / debug: one of the device-built arrays (0 suf, 1 suf_sig, 2 suf_key, 3 gram, 4 gram5, 5 suf_b); returns its element count
  uint64_t download(int which, void* dst, uint64_t capacity_bytes) const;
  uint64_t bytes() const;
  uint32_t sub_queries() const;
  void     set_sub_queries(uint32_t n);
  void     set_host_sub_queries(uint32_t n);
  void     set_initial_segment(uint64_t n);
  void     set_heavy_min(uint32_t n);   // ranges at least this long use the bit-plane kernel (0 = never)
  void     set_probe_count(bool on);    // measurement: k_probe also counts the bytes / 128-byte lines it requests
  void     set_serial(bool on);         // measurement: no side stream, every kernel alone on the GPU
  void     set_turns(uint32_t n);        // host-buffer calls that may own the GPU's kernels at once (0 = no turns)
  void     set_device_turns(uint32_t n);  // chaining of the device-resident entry points only (default 0)
  void     set_expand_order(uint32_t n);
  void     set_b_grid(uint32_t n);
  void     set_heavy_grid(uint32_t n);  // workgroups of the bit-plane kernel
  void     set_light_max(uint32_t n);   // queries with at most n hits, all in short ranges, skip phase A (0 = none do)
  void     set_probe_split(bool on);    // the probe as head + tail kernels (default) or as one kernel
  void     set_light_flat(uint32_t n);  // ... and of those, queries with at most n hits are walked several to a wave (0 = none)
  void     set_contexts(uint32_t n);    // concurrent calls served at once (each owns a workspace)
  void     set_max_host_active(uint32_t n);  // host-buffer calls that may have work on the GPU at once, whatever "contexts" says (default 4): further ones wait on the host

  // every pointer in HBM of device(); sub_tok_bounds = q_off at each sub-batch boundary (+ q_off[nq])
  void match_device(const HostIndex& h, const Params& prm, const DeviceBatchView& b, Rec* d_out,
                    uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap, void* stream,
                    const std::vector<uint32_t>& sub_tok_bounds, Stats* st);
  // the two halves of match_device without statistics: begin enqueues every pass and returns, end waits (and re-runs a pass whose
  // workspace overflowed).  Several calls may be pending at once, each in its own context; end on the thread that began.
  struct Pending;
  Pending* match_device_begin(const HostIndex& h, const Params& prm, const DeviceBatchView& b, Rec* d_out, uint32_t* d_out_count,
                              uint32_t* d_out_total, uint32_t cap, void* stream, const std::vector<uint32_t>& sub_tok_bounds);
  void match_device_end(Pending*);
  void match_device_abandon(Pending*);
  // host buffers: H2D, match_device, D2H
  // d_keep / d_keep_count (optional, in the HBM of device()): the records are also left there
  void match_host(const HostIndex& h, const Params& prm, const HostBatchView& b, Rec* out, uint32_t* out_count,
                  uint32_t* out_total, uint32_t cap, Stats* st, Rec* d_keep = nullptr, uint32_t* d_keep_count = nullptr);
  // pre-tokenised lines as they sit in a file (blank-separated tokens, one pattern per '\n'-terminated line): cut, looked up and
  // matched on the device
  void match_lines(const HostIndex& h, const Params& prm, const char* text, uint64_t n_bytes, uint64_t n_lines, Rec* out,
                   uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st);
  // raw text lines: tokenised by the device's byte-class machine where it applies (ASCII letters / digits / punctuation / blanks);
  // bad_lines = the lines with any other byte, for the host front-end
  void match_raw_lines(const HostIndex& h, const Params& prm, const char* text, uint64_t n_bytes, uint64_t n_lines, Rec* out,
                       uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st, const RawLineTok& tok,
                       std::vector<uint64_t>& bad_lines);
  // the two halves of match_host (the pipeline path) for ONE host thread that keeps several host-buffer calls in flight
  struct HostCall;
  HostCall* match_host_begin(const HostIndex& h, const Params& prm, const HostBatchView& b, Rec* out, uint32_t* out_count,
                             uint32_t* out_total, uint32_t cap, Rec* d_keep, uint32_t* d_keep_count);
  void match_host_end(HostCall*);
  void match_host_abandon(HostCall*);
  void     set_pair_finish(bool on);     // two queries per wave in the replay / top-N kernel
  void     set_real_hybrid(bool on);     // real-token indexes under unit costs: k_dp_tab<true> + float kernel on the marked pairs (A/B switch)
  void     set_small_path(bool on);      // micro-batches (<= 512 queries) in one kernel launch (fmx_small.hip.hh)
  // FuzzyMatch::subsequence (src/fuzzy_match.cc:250-365) for a batch of tokenised patterns; one record per query
  void subsequence_host(const HostIndex& h, const HostBatchView& b, uint32_t number_of_matches, bool no_perfect, int min_subseq_length,
                        float min_subseq_ratio, bool idf_weighting, Rec* out, uint32_t* out_found, uint32_t* out_pos);
  // q_off values at the sub-batch boundaries of a device-resident batch, and its longest pattern
  std::vector<uint32_t> sub_bounds_from_device(const uint32_t* d_q_off, uint64_t nq, uint32_t* p_max, void* stream);

 private:
  // the pipeline for a host batch (match_host tries the one-launch path for micro-batches first)
  void match_host_full(const HostIndex& h, const Params& prm, const HostBatchView& b, Rec* out, uint32_t* out_count,
                       uint32_t* out_total, uint32_t cap, Stats* st, Rec* d_keep, uint32_t* d_keep_count);
  bool match_host_small(const HostIndex& h, const Params& prm, const HostBatchView& b, Rec* out, uint32_t* out_count,
                        uint32_t* out_total, uint32_t cap, Stats* st);
  void match_lines_impl(const HostIndex& h, const Params& prm, const char* text, uint64_t n_bytes, uint64_t n_lines, Rec* out,
                        uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st, const RawLineTok* raw,
                        std::vector<uint64_t>* bad_lines);
  Impl* impl_;
};

}  // namespace fmx
