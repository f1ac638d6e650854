// fmx_gpu.hip — device residency of the index and the batched match() pipeline.
//
// HBM layout (immutable, SoA, replicated per GPU — SURVEY §8e):
//   suf      SufRec[n_suf]  8 B   {token offset, tokens left, sentence length}: one load per compare step
//   suf_sid  u32[n_suf]           sentence id, only read for hits that survive the filters
//   tok      u32[n_buf]           the reference's _sentence_buffer verbatim: [len, tok x len, 0]
//   sent_pos u32[n_sent]          _sentence_pos
//   qva      u32[vocab+1]         _quickVocabAccess
//   idf      f32[vocab]           logf(N / sfreq[w]) computed on the host with glibc (bit-exact scores)
//   real / itok / itok_dist       only for indexes whose real tokens differ from the normalised ones
#include "fmx_gpu.hh"

#include <string.h>
#include <cstring>
#include <functional>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <chrono>
#include <stdexcept>
#include <string>

#include "fmx_kernels.hip.hh"

namespace fmx {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      throw GpuError(std::string(#expr) + ": " + hipGetErrorString(e_));                            \
  } while (0)

int gpu_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  int usable = 0;
  for (int d = 0; d < n; d++) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, d) != hipSuccess) continue;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) == 0) usable++;
  }
  return usable;
}

template <class T> static T* dev_alloc(size_t n) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
  return (T*)p;
}

template <class T> static T* dev_upload(const std::vector<T>& v) {
  T* p = dev_alloc<T>(v.size());
  if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}

struct DeviceIndex::Impl {
  int      device = -1;
  DevIndex d{};
  uint64_t bytes = 0;
  uint64_t itok_version = ~0ull;
  // workspace (grown on demand, reused across calls; one call at a time per index)
  uint64_t cap_q = 0, cap_t = 0, cap_seg = 0;
  DevWork  w{};
  DevScratch sc{};
  unsigned long long* tile_sums = nullptr;
  unsigned long long* grand_total = nullptr;
  uint32_t* big_list = nullptr;
  uint32_t* mid_list = nullptr;
  uint32_t* fin_list = nullptr;
  uint32_t* contrast_list = nullptr;  // 2 x cap_q
  WorkItem* items = nullptr;
  WorkItem* items_b = nullptr;
  uint32_t  item_cap = 0;
  uint32_t  s_max = 4096;   // survivors kept per query before a retry with a larger segment
  // staging for the host-batch entry point
  uint32_t *d_qoff = nullptr, *d_qtok = nullptr, *d_qreal = nullptr, *d_qitok = nullptr;
  uint64_t  cap_hq = 0, cap_ht = 0;
  Rec*      d_out = nullptr;
  uint32_t *d_out_count = nullptr, *d_out_total = nullptr;
  uint64_t  cap_out = 0, cap_outq = 0;
  hipEvent_t ev[2 * kNumStages]{};
  bool       ev_ok = false;
  uint32_t   sub_queries = 1u << 20;  // queries per pipeline pass: large passes amortise the ~40 launches and their tails
  uint32_t   host_sub_queries = 1u << 18;  // pass size of the host-buffer entry point: result copies overlap the next pass
  hipStream_t copy_stream = nullptr;       // D2H of finished passes (non-blocking stream)
  hipStream_t side_stream = nullptr;       // the few-queries / long-latency tail kernels run beside the bulk kernels
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  uint16_t*  cmin_tab = nullptr;           // threshold table of k_cmin, (cmin_rows) x 256
  uint32_t   cmin_rows = 0;
  std::function<void(uint32_t)> on_launched;  // match_host's hook: pass `sb` is enqueued, passes < sb are complete
  uint64_t   init_seg = 1ull << 24;
};

DeviceIndex::DeviceIndex() : impl_(new Impl) {}

static void free_ws(DeviceIndex::Impl& m) {
  hipFree(m.w.q_m); hipFree(m.w.t_q); hipFree(m.w.t_lo); hipFree(m.w.t_cnt); hipFree(m.w.q_hits);
  hipFree(m.w.q_seg); hipFree(m.w.q_kept); hipFree(m.w.q_cand); hipFree(m.w.q_acc); hipFree(m.w.q_pb); hipFree(m.w.q_aux); hipFree(m.w.q_rank); hipFree(m.w.p_hist); hipFree(m.w.c_in); hipFree(m.w.c_off); hipFree(m.w.t_idf); hipFree(m.w.c_qi); hipFree(m.big_list); m.big_list = nullptr; hipFree(m.mid_list); m.mid_list = nullptr; hipFree(m.fin_list); m.fin_list = nullptr; hipFree(m.contrast_list); m.contrast_list = nullptr;
  hipFree(m.w.seg); hipFree(m.w.qb); hipFree(m.w.heap); hipFree(m.w.counters);
  hipFree(m.sc.keys); hipFree(m.sc.vals); hipFree(m.sc.f0); hipFree(m.sc.f1);
  hipFree(m.tile_sums); hipFree(m.grand_total); hipFree(m.w.q_cap);
  m.w = DevWork{}; m.sc = DevScratch{};
  m.tile_sums = m.grand_total = nullptr;
  m.cap_q = m.cap_t = m.cap_seg = 0;
}

DeviceIndex::~DeviceIndex() {
  if (!impl_) return;
  Impl& m = *impl_;
  if (m.device >= 0) {
    (void)hipSetDevice(m.device);
    hipFree((void*)m.d.suf); hipFree((void*)m.d.suf_sig); hipFree((void*)m.d.suf_key); hipFree((void*)m.d.gram.g3); hipFree((void*)m.d.gram.g5); hipFree((void*)m.d.suf_b); hipFree((void*)m.d.tok); hipFree((void*)m.d.sent_pos);
    hipFree((void*)m.d.qva); hipFree((void*)m.d.idf); hipFree((void*)m.d.real); hipFree((void*)m.d.itok);
    hipFree((void*)m.d.itok_dist);
    free_ws(m);
    hipFree(m.d_qoff); hipFree(m.d_qtok); hipFree(m.d_qreal); hipFree(m.d_qitok);
    hipFree(m.d_out); hipFree(m.d_out_count); hipFree(m.d_out_total); hipFree(m.items); hipFree(m.items_b);
    if (m.copy_stream) { hipStreamDestroy(m.copy_stream); m.copy_stream = nullptr; }
    if (m.side_stream) { hipStreamDestroy(m.side_stream); m.side_stream = nullptr; hipEventDestroy(m.ev_fork); hipEventDestroy(m.ev_join); }
    hipFree(m.cmin_tab); m.cmin_tab = nullptr; m.cmin_rows = 0;
    if (m.ev_ok) for (auto& e : m.ev) hipEventDestroy(e);
  }
  delete impl_;
}

int      DeviceIndex::device() const { return impl_->device; }
uint64_t DeviceIndex::bytes() const { return impl_->bytes; }
void     DeviceIndex::set_sub_queries(uint32_t n) { impl_->sub_queries = std::max<uint32_t>(n, 1); }
void     DeviceIndex::set_host_sub_queries(uint32_t n) { impl_->host_sub_queries = std::max<uint32_t>(n, 1); }
void     DeviceIndex::set_initial_segment(uint64_t n) { impl_->init_seg = std::max<uint64_t>(n, 1024); }

static void upload_itok_table(DeviceIndex::Impl& m, const HostIndex& h) {
  if (m.itok_version == h.itok_version) return;
  hipFree((void*)m.d.itok_dist);
  m.d.itok_dist = nullptr;
  m.d.n_itok = (uint32_t)h.itok_strings.size();
  if (!h.itok_dist.empty()) m.d.itok_dist = dev_upload(h.itok_dist);
  m.itok_version = h.itok_version;
}

void DeviceIndex::upload(const HostIndex& h, int device) {
  Impl& m = *impl_;
  if (m.device >= 0) throw GpuError("index is already on a device");
  if (gpu_device_count() <= device || device < 0) throw GpuError("no usable gfx950 device " + std::to_string(device));
  HIP_TRY(hipSetDevice(device));
  m.device = device;
  m.d.suf = dev_upload(h.suf_rec);
  m.d.suf_b = dev_upload(h.suf_b);
  m.d.suf_sig = (const unsigned long long*)dev_upload(h.suf_sig);
  m.d.suf_key = dev_upload(h.suf_key);
  m.d.gram.g3 = h.gram.empty() ? nullptr : dev_upload(h.gram);
  m.d.gram.mask3 = h.gram.empty() ? 0u : (uint32_t)(h.gram.size() - 1);
  m.d.gram.g5 = h.gram5.empty() ? nullptr : dev_upload(h.gram5);
  m.d.gram.mask5 = h.gram5.empty() ? 0u : (uint32_t)(h.gram5.size() - 1);
  {  // eight spare tokens: the DP kernels read the sentence four tokens at a time, possibly past its end
    uint32_t* t = dev_alloc<uint32_t>(h.buf.size() + 8);
    HIP_TRY(hipMemset(t, 0, (h.buf.size() + 8) * sizeof(uint32_t)));
    if (!h.buf.empty()) HIP_TRY(hipMemcpy(t, h.buf.data(), h.buf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    m.d.tok = t;
  }
  m.d.sent_pos = dev_upload(h.sent_pos);
  m.d.qva = dev_upload(h.qva);
  m.d.idf = dev_upload(h.idf);
  m.d.n_qva = (uint32_t)h.qva.size();
  m.d.n_sent = h.num_sentences();
  m.d.n_suf = h.num_suffixes();
  uint32_t mx = 0;
  for (uint32_t sp : h.sent_pos) mx = std::max(mx, h.buf[sp]);
  m.d.max_slen = mx;
  m.bytes = h.suf_rec.size() * 8 + h.suf_sig.size() * 8 + h.suf_key.size() * sizeof(SufKey) + h.gram.size() * 16 + h.gram5.size() * 16 + h.suf_b.size() * 16 + h.buf.size() * 4 + h.sent_pos.size() * 4 +
            h.qva.size() * 4 + h.idf.size() * 4;
  if (h.real_differs) {
    m.d.real = dev_upload(h.real_ids);
    m.d.itok = dev_upload(h.itok_ids);
    m.bytes += h.real_ids.size() * 4 + h.itok_ids.size() * 4;
  }
  upload_itok_table(m, h);
  for (auto& e : m.ev) HIP_TRY(hipEventCreate(&e));
  m.ev_ok = true;
  HIP_TRY(hipDeviceSynchronize());
}

static void ensure_ws(DeviceIndex::Impl& m, uint64_t nq, uint64_t nt, uint64_t seg) {
  if (nq > m.cap_q || nt > m.cap_t || seg > m.cap_seg) {
    const uint64_t cq = std::max(nq, m.cap_q), ct = std::max(nt, m.cap_t), cs = std::max(seg, m.cap_seg);
    free_ws(m);
    m.w.q_m = dev_alloc<uint32_t>(cq);
    m.w.t_q = dev_alloc<uint32_t>(ct);
    m.w.t_lo = dev_alloc<uint32_t>(ct);
    m.w.t_cnt = dev_alloc<uint32_t>(ct);
    m.w.q_hits = dev_alloc<unsigned long long>(cq + 1);
    m.w.q_seg = dev_alloc<unsigned long long>(cq + 1);
    m.w.q_cap = dev_alloc<unsigned long long>(cq + 1);
    m.w.q_kept = dev_alloc<uint32_t>(cq);
    m.w.q_cand = dev_alloc<uint32_t>(cq);
    m.w.q_acc = dev_alloc<uint32_t>(cq);
    m.w.q_pb = dev_alloc<uint32_t>(cq);
    m.w.q_aux = dev_alloc<uint32_t>(cq);
    m.w.q_rank = dev_alloc<uint32_t>(cq);
    m.w.p_hist = dev_alloc<uint32_t>(2 * kPBins);
    m.w.c_in = dev_alloc<unsigned long long>(3 * (cq + 1));
    m.w.c_off = dev_alloc<unsigned long long>(3 * (cq + 1));
    m.w.t_idf = dev_alloc<float>(ct);
    m.w.c_qi = dev_alloc<unsigned long long>(cs);
    m.big_list = dev_alloc<uint32_t>(cq);
    m.mid_list = dev_alloc<uint32_t>(cq);
    m.fin_list = dev_alloc<uint32_t>(cq);
    m.contrast_list = dev_alloc<uint32_t>(2 * cq);
    m.w.seg = dev_alloc<Cand>(cs);
    m.w.qb = dev_alloc<QHit>(cs);
    m.w.heap = dev_alloc<float>(cs + cq + 1);
    m.w.counters = dev_alloc<unsigned long long>(16);
    m.w.seg_cap = cs;
    m.sc.keys = dev_alloc<unsigned long long>(2 * cs);
    m.sc.vals = dev_alloc<uint32_t>(2 * cs);
    m.sc.f0 = dev_alloc<float>(cs);
    m.sc.f1 = dev_alloc<float>(cs);
    m.tile_sums = dev_alloc<unsigned long long>(kScanTile);
    m.grand_total = dev_alloc<unsigned long long>(1);
    m.cap_q = cq; m.cap_t = ct; m.cap_seg = cs;
  }
}

const char* stage_name(int s) {
  static const char* names[kNumStages] = {"setup", "probe", "scan", "expand", "cand", "cand_big", "cand_scan", "dp", "finish"};
  return (s >= 0 && s < kNumStages) ? names[s] : "";
}

void DeviceIndex::match_device(const HostIndex& h, const Params& prm_in, const DeviceBatchView& bv, Rec* d_out,
                               uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap, void* stream_v,
                               const std::vector<uint32_t>& sub_tok_bounds, Stats* st) {
  Impl& m = *impl_;
  if (m.device < 0) throw GpuError("index is not on a device");
  HIP_TRY(hipSetDevice(m.device));
  hipStream_t stream = (hipStream_t)stream_v;
  upload_itok_table(m, h);
  Params prm = prm_in;
  if (cap == 0) throw GpuError("cap must be > 0");
  // cost model: real-token / penalty-token terms need the general kernel (rows in LDS); idf alone has a
  // register kernel
  const bool side_terms = (m.d.real && bv.q_real) || (bv.q_itok != nullptr) || (m.d.itok != nullptr);
  const bool use_idf = prm.idf_penalty != 0.f;
  const bool full = side_terms;
  const uint32_t n_sub = (uint32_t)sub_tok_bounds.size() - 1;
  for (uint32_t sb = 0; sb < n_sub; sb++) {
    DevBatch b;
    b.q_off = bv.q_off; b.q_tok = bv.q_tok; b.q_real = bv.q_real; b.q_itok = bv.q_itok;
    b.q_begin = sb * m.sub_queries;
    b.nq = (uint32_t)std::min<uint64_t>(m.sub_queries, bv.nq - b.q_begin);
    b.t_begin = sub_tok_bounds[sb];
    b.nt = sub_tok_bounds[sb + 1] - sub_tok_bounds[sb];
    uint64_t want_seg = std::max<uint64_t>(m.cap_seg, m.init_seg);
    uint32_t p_max = (uint32_t)std::min<uint64_t>(prm.max_tokens, std::max<uint32_t>(b.nt, 1));
    p_max = std::max<uint32_t>(1, std::min<uint32_t>(p_max, bv.p_max ? bv.p_max : p_max));
    for (int attempt = 0;; attempt++) {
      ensure_ws(m, b.nq, std::max<uint32_t>(b.nt, 1), want_seg);
      if (!m.items) { m.item_cap = 1u << 22; m.items = dev_alloc<WorkItem>(m.item_cap); m.items_b = dev_alloc<WorkItem>(m.item_cap); }
      DevWork w = m.w;
      HIP_TRY(hipMemsetAsync(w.q_hits, 0, (b.nq + 1) * sizeof(unsigned long long), stream));
      HIP_TRY(hipMemsetAsync(w.counters, 0, 16 * sizeof(unsigned long long), stream));
      HIP_TRY(hipMemsetAsync(w.q_pb, 0, b.nq * sizeof(uint32_t), stream));
      int e = 0;
      auto tick = [&]() { HIP_TRY(hipEventRecord(m.ev[e++], stream)); };
      tick();
      hipLaunchKernelGGL(k_setup, dim3((b.nq + 255) / 256), dim3(256), 0, stream, m.d, b, w, prm);
      if (prm.idf_penalty != 0.f) hipLaunchKernelGGL(k_tidf, dim3((b.nt + 255) / 256), dim3(256), 0, stream, m.d, b, w);
      HIP_TRY(hipMemsetAsync(w.p_hist, 0, 2 * kPBins * sizeof(uint32_t), stream));
      {
        const uint32_t per_block = kRankThreads * kRankPer;
        hipLaunchKernelGGL(k_phist, dim3((b.nq + per_block - 1) / per_block), dim3(kRankThreads), 0, stream, b, w);
        hipLaunchKernelGGL(k_prank, dim3((b.nq + per_block - 1) / per_block), dim3(kRankThreads), 0, stream, b, w);
      }
      tick(); tick();
      if (b.nt) hipLaunchKernelGGL(k_probe, dim3((b.nt + 63) / 64), dim3(64), 0, stream, m.d, b, w);
      tick(); tick();
      {
        hipLaunchKernelGGL(k_caps, dim3((b.nq + 1 + 255) / 256), dim3(256), 0, stream, b, w, m.items, m.item_cap, m.s_max);
        const uint32_t n = b.nq + 1;  // one extra so that q_seg[nq] = total
        const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
        if (tiles > (uint32_t)kScanTile) throw GpuError("sub-batch too large for the scan");
        hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(kScanThreads), 0, stream, w.q_cap, w.q_seg, m.tile_sums, n);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, stream, m.tile_sums, tiles, m.grand_total);
        hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(kScanThreads), 0, stream, w.q_seg, m.tile_sums, n);
      }
      tick(); tick();
      {
        // the signature bound needs theoretical_rejection_cover to be monotone in the cover: costs >= 0
        const int use_sig = prm.ins >= 0.f && prm.del >= 0.f && prm.rep >= 0.f;
        if (p_max + 1 > m.cmin_rows) {
          hipFree(m.cmin_tab);
          m.cmin_rows = p_max + 1;
          m.cmin_tab = dev_alloc<uint16_t>((size_t)m.cmin_rows * 256);
        }
        w.cmin_tab = m.cmin_tab;
        hipLaunchKernelGGL(k_cmin, dim3(p_max + 1), dim3(256), 0, stream, m.d.max_slen, prm, use_sig, m.cmin_tab);
        hipLaunchKernelGGL(k_expand_a, dim3(b.nq + kExpandExtraGrid), dim3(kWave), 0, stream, m.d, b, w, prm,
                           (const WorkItem*)m.items, m.item_cap, use_sig);
        hipLaunchKernelGGL(k_caps_b, dim3((b.nq + 255) / 256), dim3(256), 0, stream, b, w, m.items_b, m.item_cap);
        const size_t lds = (size_t)p_max * 4;
        if (lds > 64 * 1024) throw GpuError("patterns too long for the expand kernel's LDS staging");
        hipLaunchKernelGGL(k_expand_b, dim3(b.nq + kExpandExtraGrid), dim3(kWave), lds, stream, m.d, b, w, prm,
                           (const WorkItem*)m.items_b, m.item_cap);
      }
      tick(); tick();
      if (!m.side_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&m.side_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&m.ev_fork, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&m.ev_join, hipEventDisableTiming));
      }
      // fork: `side` continues after everything enqueued on `stream` so far; join: `stream` waits for `side`
      auto fork = [&]() { HIP_TRY(hipEventRecord(m.ev_fork, stream)); HIP_TRY(hipStreamWaitEvent(m.side_stream, m.ev_fork, 0)); };
      auto join = [&]() { HIP_TRY(hipEventRecord(m.ev_join, m.side_stream)); HIP_TRY(hipStreamWaitEvent(stream, m.ev_join, 0)); };
      hipLaunchKernelGGL(k_cand, dim3((b.nq + 3) / 4), dim3(256), 0, stream, b, w, m.big_list, m.mid_list);
      {
        // The sorter of the few huge survivor sets is one workgroup per query and leaves the GPU almost idle:
        // it runs on the side stream next to k_cand_mid (disjoint queries).
        // radix passes needed: sentence ids < n_sent, match lengths <= longest pattern
        uint32_t sid_bytes = 1, len_bytes = p_max < 255 ? 1u : 2u;
        while (sid_bytes < 4 && (m.d.n_sent >> (8 * sid_bytes)) != 0) sid_bytes++;
        fork();
        hipLaunchKernelGGL(k_cand_big, dim3(kBigGrid), dim3(kBigThreads), 0, m.side_stream, w, m.sc.keys,
                           (const uint32_t*)m.big_list, sid_bytes, len_bytes);
      }
      hipLaunchKernelGGL(k_cand_mid, dim3(1024), dim3(256), 0, stream, w, (const uint32_t*)m.mid_list);
      tick(); tick();
      join();  // stage "cand_big" = what is left of the big sorter after k_cand_mid
      tick(); tick();
      {
        // dense per-class candidate lists: three exclusive scans of the per-query candidate counts
        const uint32_t n = b.nq + 1;
        const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
        hipLaunchKernelGGL(k_classify, dim3((n + 255) / 256), dim3(256), 0, stream, b, w, full ? 1 : 0);
        for (int c = 0; c < 3; c++) {
          hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(kScanThreads), 0, stream, w.c_in + (size_t)c * n,
                             w.c_off + (size_t)c * n, m.tile_sums, n);
          hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, stream, m.tile_sums, tiles, m.grand_total);
          hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(kScanThreads), 0, stream, w.c_off + (size_t)c * n, m.tile_sums, n);
        }
        hipLaunchKernelGGL(k_cmap, dim3((b.nq + 255) / 256), dim3(256), 0, stream, b, w, full ? 1 : 0, m.fin_list);
      }
      tick(); tick();
      {
        const size_t lds = ((size_t)p_max + 1) * kWave * 4;
        if (lds > 150 * 1024)
          throw GpuError("patterns of " + std::to_string(p_max) + " tokens need more LDS than one CU has (limit ~590)");
        const unsigned grid = 256 * 16;  // grid-stride over the flat candidate list
        if (full) {
          HIP_TRY(hipFuncSetAttribute((const void*)k_dp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(k_dp<true>, dim3(grid), dim3(kWave), lds, stream, m.d, b, w, prm);
        } else {
          // by pattern length: row + pattern in 32 or 64 registers, longer ones with the row in LDS
          if (use_idf) {
            hipLaunchKernelGGL((k_dp_reg<32, true>), dim3(grid), dim3(kWave), 0, stream, m.d, b, w, prm);
            if (p_max > 32) hipLaunchKernelGGL((k_dp_reg<64, true>), dim3(grid), dim3(kWave), 0, stream, m.d, b, w, prm);
          } else if (prm.ins == prm.del && prm.del == prm.rep && prm.rep >= 0.f) {
            // unit costs: bit-parallel Levenshtein distance, then the float sum the DP would have produced
            // few registers: eight waves per SIMD hide the dependent loads in front of every candidate
            hipLaunchKernelGGL((k_dp_reg<32, false, true>), dim3(grid), dim3(kWave), 0, stream, m.d, b, w, prm);
            if (p_max > 32) hipLaunchKernelGGL((k_dp_reg<64, false, true>), dim3(grid), dim3(kWave), 0, stream, m.d, b, w, prm);
          } else {
            hipLaunchKernelGGL((k_dp_reg<32, false>), dim3(grid), dim3(kWave), 0, stream, m.d, b, w, prm);
            if (p_max > 32) hipLaunchKernelGGL((k_dp_reg<64, false>), dim3(grid), dim3(kWave), 0, stream, m.d, b, w, prm);
          }
          if (p_max > 64) {
            if (use_idf) {
              HIP_TRY(hipFuncSetAttribute((const void*)k_dp<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
              hipLaunchKernelGGL(k_dp<true>, dim3(grid), dim3(kWave), lds, stream, m.d, b, w, prm);
            } else {
              HIP_TRY(hipFuncSetAttribute((const void*)k_dp<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
              hipLaunchKernelGGL(k_dp<false>, dim3(grid), dim3(kWave), lds, stream, m.d, b, w, prm);
            }
          }
        }
      }
      tick(); tick();
      {
        // the queue of large candidate sets was filled by k_cmap: their (long, mostly sequential) replay runs on the
        // side stream next to the bulk kernel; the two touch disjoint queries
        fork();
        hipLaunchKernelGGL((k_finish<kLdsSortMax, true>), dim3(2048), dim3(kWave), 0, m.side_stream, m.d, b, w, prm, m.sc, cap,
                           d_out, d_out_count, d_out_total, m.contrast_list, m.fin_list);
        hipLaunchKernelGGL((k_finish<64, false>), dim3(b.nq), dim3(kWave), 0, stream, m.d, b, w, prm, m.sc, cap, d_out,
                           d_out_count, d_out_total, m.contrast_list, m.fin_list);
        join();
        if (prm.contrast > 0.f) {
          const uint32_t* cl = m.contrast_list;
          if (prm.nmatch >= 1 && prm.nmatch <= kMaxFlatRounds) {
            // flattened rounds: pick, then one edit distance per remaining accepted match of any query
            const uint32_t n = b.nq + 1;
            const uint32_t tiles = (n + kScanTile - 1) / kScanTile;
            hipLaunchKernelGGL(k_cacc, dim3((n + 255) / 256), dim3(256), 0, stream, b, w);
            for (int c = 0; c < 2; c++) {
              hipLaunchKernelGGL(k_scan_tiles, dim3(tiles), dim3(kScanThreads), 0, stream, w.c_in + (size_t)c * n,
                                 w.c_off + (size_t)c * n, m.tile_sums, n);
              hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanThreads), 0, stream, m.tile_sums, tiles, m.grand_total);
              hipLaunchKernelGGL(k_scan_add, dim3(tiles), dim3(kScanThreads), 0, stream, w.c_off + (size_t)c * n, m.tile_sums, n);
            }
            hipLaunchKernelGGL(k_camap, dim3(b.nq), dim3(kWave), 0, stream, b, w, m.sc);
            for (uint32_t r = 0; r < prm.nmatch; r++) {
              if (r) {
                hipLaunchKernelGGL(k_cdp<32>, dim3(256 * 16), dim3(kWave), 0, stream, m.d, b, w, prm, m.sc, r);
                if (m.d.max_slen > 32)
                  hipLaunchKernelGGL(k_cdp<64>, dim3(256 * 8), dim3(kWave), 0, stream, m.d, b, w, prm, m.sc, r);
              }
              hipLaunchKernelGGL(k_cpick, dim3((b.nq + 255) / 256), dim3(256), 0, stream, m.d, b, w, prm, m.sc, r, cap, d_out, d_out_count,
                                 d_out_total);
            }
          } else {
            hipLaunchKernelGGL(k_contrast<0>, dim3(b.nq), dim3(kWave), 0, stream, m.d, b, w, prm, m.sc, cap, d_out,
                               d_out_count, d_out_total, cl);
          }
          const bool flat = prm.nmatch >= 1 && prm.nmatch <= kMaxFlatRounds;
          if (m.d.max_slen > 32 && !flat)
            hipLaunchKernelGGL(k_contrast<1>, dim3(2048), dim3(kWave), 0, stream, m.d, b, w, prm, m.sc, cap, d_out,
                               d_out_count, d_out_total, cl);
          if (m.d.max_slen > 64) {
            const size_t lds = ((size_t)m.d.max_slen + 1) * kWave * 4;
            if (lds > 120 * 1024) throw GpuError("contrastive rerank: indexed sentences too long for the LDS DP rows");
            HIP_TRY(hipFuncSetAttribute((const void*)k_contrast<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(k_contrast<2>, dim3(512), dim3(kWave), lds, stream, m.d, b, w, prm, m.sc, cap, d_out,
                               d_out_count, d_out_total, cl);
          }
        }
      }
      tick();
      hipLaunchKernelGGL(k_stats, dim3((b.nq + 1023) / 1024), dim3(1024), 0, stream, w, b.nq);
      HIP_TRY(hipGetLastError());
      if (m.on_launched) m.on_launched(sb);  // host-buffer path: copy out finished passes while this one runs
      unsigned long long ctr[16];
      HIP_TRY(hipMemcpyAsync(ctr, w.counters, sizeof(ctr), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
      if (st) {
        for (int s = 0; s < kNumStages; s++) {
          float ms = 0.f;
          HIP_TRY(hipEventElapsedTime(&ms, m.ev[2 * s], m.ev[2 * s + 1]));
          st->ms_kernel[s] += ms;
          st->launches[s] += 1;
        }
      }
      bool retry = false;
      if (ctr[6] > m.w.seg_cap) { want_seg = ctr[6] + ctr[6] / 4; retry = true; }       // segments did not fit
      if (ctr[4] > m.s_max) {                                                          // a query had more survivors
        uint32_t s = m.s_max;
        while (s < ctr[4]) s <<= 1;
        m.s_max = s;
        retry = true;
      }
      if (std::max(ctr[5], ctr[12]) > m.item_cap) {                                    // work-item list too short
        hipFree(m.items); hipFree(m.items_b);
        const unsigned long long need = std::max(ctr[5], ctr[12]);
        m.item_cap = (uint32_t)std::min<unsigned long long>(need + need / 4, 0xffffffffull);
        m.items = dev_alloc<WorkItem>(m.item_cap);
        m.items_b = dev_alloc<WorkItem>(m.item_cap);
        retry = true;
      }
      if (retry) {
        if (attempt >= 6) throw GpuError("workspace kept overflowing");
        if (st) st->n_retries++;
        continue;
      }
      if (st) {
        st->n_hits += ctr[0]; st->n_kept += ctr[1]; st->n_candidates += ctr[2]; st->n_dp += ctr[3];
        st->n_big += ctr[7]; st->n_phase_b += ctr[8]; st->max_kept = std::max<uint64_t>(st->max_kept, ctr[9]);
      }
      break;
    }
  }
}

void DeviceIndex::match_host(const HostIndex& h, const Params& prm, const HostBatchView& hb, Rec* out,
                             uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st) {
  Impl& m = *impl_;
  if (m.device < 0) throw GpuError("index is not on a device");
  HIP_TRY(hipSetDevice(m.device));
  const auto t0 = std::chrono::steady_clock::now();
  const uint64_t nq = hb.nq;
  if (nq == 0) return;
  const uint64_t nt = hb.q_off[nq];
  if (nq + 1 > m.cap_hq) {
    hipFree(m.d_qoff);
    m.d_qoff = dev_alloc<uint32_t>(nq + 1);
    m.cap_hq = nq + 1;
  }
  if (nt + nq + 1 > m.cap_ht) {
    hipFree(m.d_qtok); hipFree(m.d_qreal); hipFree(m.d_qitok);
    m.d_qtok = dev_alloc<uint32_t>(nt + 1);
    m.d_qreal = dev_alloc<uint32_t>(nt + 1);
    m.d_qitok = dev_alloc<uint32_t>(nt + nq + 1);
    m.cap_ht = nt + nq + 1;
  }
  if (nq * cap > m.cap_out || nq > m.cap_outq) {
    hipFree(m.d_out); hipFree(m.d_out_count); hipFree(m.d_out_total);
    m.d_out = dev_alloc<Rec>(nq * cap);
    m.d_out_count = dev_alloc<uint32_t>(nq);
    m.d_out_total = dev_alloc<uint32_t>(nq);
    m.cap_out = nq * cap; m.cap_outq = nq;
  }
  const auto h0 = std::chrono::steady_clock::now();
  HIP_TRY(hipMemcpy(m.d_qoff, hb.q_off, (nq + 1) * 4, hipMemcpyHostToDevice));
  if (nt) HIP_TRY(hipMemcpy(m.d_qtok, hb.q_tok, nt * 4, hipMemcpyHostToDevice));
  if (hb.q_real && nt) HIP_TRY(hipMemcpy(m.d_qreal, hb.q_real, nt * 4, hipMemcpyHostToDevice));
  if (hb.q_itok) HIP_TRY(hipMemcpy(m.d_qitok, hb.q_itok, (nt + nq) * 4, hipMemcpyHostToDevice));
  const auto h1 = std::chrono::steady_clock::now();
  DeviceBatchView bv;
  bv.nq = nq; bv.q_off = m.d_qoff; bv.q_tok = m.d_qtok;
  bv.q_real = hb.q_real ? m.d_qreal : nullptr;
  bv.q_itok = hb.q_itok ? m.d_qitok : nullptr;
  uint32_t p_max = 0;
  for (uint64_t q = 0; q < nq; q++) p_max = std::max(p_max, hb.q_off[q + 1] - hb.q_off[q]);
  bv.p_max = std::max<uint32_t>(p_max, 1);
  // Passes of host_sub_queries queries: the results of pass i travel to the host on their own stream while
  // pass i + 1 computes, so only the last pass's copy is exposed.
  const uint32_t saved_sub = m.sub_queries;
  m.sub_queries = std::min(saved_sub, m.host_sub_queries);
  std::vector<uint32_t> bounds;
  for (uint64_t q = 0; q < nq; q += m.sub_queries) bounds.push_back(hb.q_off[q]);
  bounds.push_back(hb.q_off[nq]);
  if (!m.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&m.copy_stream, hipStreamNonBlocking));
  uint64_t copied_q = 0;
  double ms_copy = 0.0;
  auto copy_upto = [&](uint64_t q_end) {
    if (q_end <= copied_q) return;
    const auto c0 = std::chrono::steady_clock::now();
    const uint64_t n = q_end - copied_q;
    HIP_TRY(hipMemcpyAsync(out + copied_q * cap, m.d_out + copied_q * cap, n * cap * sizeof(Rec), hipMemcpyDeviceToHost, m.copy_stream));
    HIP_TRY(hipMemcpyAsync(out_count + copied_q, m.d_out_count + copied_q, n * 4, hipMemcpyDeviceToHost, m.copy_stream));
    if (out_total) HIP_TRY(hipMemcpyAsync(out_total + copied_q, m.d_out_total + copied_q, n * 4, hipMemcpyDeviceToHost, m.copy_stream));
    HIP_TRY(hipStreamSynchronize(m.copy_stream));
    copied_q = q_end;
    ms_copy += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
  };
  const uint32_t sub = m.sub_queries;
  m.on_launched = [&](uint32_t sb) { copy_upto(std::min<uint64_t>((uint64_t)sb * sub, nq)); };
  try {
    match_device(h, prm, bv, m.d_out, m.d_out_count, m.d_out_total, cap, nullptr, bounds, st);
  } catch (...) {
    m.on_launched = nullptr;
    m.sub_queries = saved_sub;
    throw;
  }
  m.on_launched = nullptr;
  m.sub_queries = saved_sub;
  copy_upto(nq);
  const auto h3 = std::chrono::steady_clock::now();
  if (st) {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    st->ms_h2d += ms(h0, h1);
    st->ms_d2h += ms_copy;  // mostly hidden behind the following pass
    st->ms_total += ms(t0, h3);
  }
}

__global__ void k_pmax(const uint32_t* q_off, unsigned long long nq, uint32_t* out) {
  const unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t p = q < nq ? q_off[q + 1] - q_off[q] : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) p = max(p, (uint32_t)__shfl_xor(p, d, 64));
  if ((threadIdx.x & 63) == 0 && p) atomicMax(out, p);
}

std::vector<uint32_t> DeviceIndex::sub_bounds_from_device(const uint32_t* d_q_off, uint64_t nq, uint32_t* p_max,
                                                          void* stream_v) {
  Impl& m = *impl_;
  HIP_TRY(hipSetDevice(m.device));
  hipStream_t stream = (hipStream_t)stream_v;
  std::vector<uint32_t> bounds;
  for (uint64_t q = 0; q < nq; q += m.sub_queries) bounds.push_back(0);
  bounds.push_back(0);
  size_t k = 0;
  for (uint64_t q = 0; q < nq; q += m.sub_queries, k++)
    HIP_TRY(hipMemcpyAsync(&bounds[k], d_q_off + q, 4, hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipMemcpyAsync(&bounds[k], d_q_off + nq, 4, hipMemcpyDeviceToHost, stream));
  if (p_max && *p_max == 0 && nq) {
    ensure_ws(m, 1, 1, 1024);
    uint32_t* d_pm = (uint32_t*)m.grand_total;
    HIP_TRY(hipMemsetAsync(d_pm, 0, 4, stream));
    hipLaunchKernelGGL(k_pmax, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, d_q_off, nq, d_pm);
    HIP_TRY(hipMemcpyAsync(p_max, d_pm, 4, hipMemcpyDeviceToHost, stream));
  }
  HIP_TRY(hipStreamSynchronize(stream));
  return bounds;
}

// ---------------------------------------------------------------------------------------------
// GPU suffix sort (index build): rocPRIM merge sort of (prefix, key) pairs
// under SuffixArray::comp (src/suffix_array.cc:235-251).  Build-side utility, not the hot path.
// ---------------------------------------------------------------------------------------------
struct GpuSortKey {
  unsigned long long prefix;  // tok0 << 32 | tok1
  unsigned long long key;     // sid << 16 | pos
};

struct GpuSuffixLess {
  const uint32_t* buf;
  const uint32_t* sent_pos;
  __device__ bool operator()(const GpuSortKey& a, const GpuSortKey& b) const {
    if (a.prefix != b.prefix) return a.prefix < b.prefix;
    const uint32_t sa = (uint32_t)(a.key >> 16), sb = (uint32_t)(b.key >> 16);
    const uint32_t pa = (uint32_t)(a.key & 0xffff), pb = (uint32_t)(b.key & 0xffff);
    const uint32_t* ba = buf + sent_pos[sa];
    const uint32_t* bb = buf + sent_pos[sb];
    const uint32_t la = ba[0] - (pa - 1), lb = bb[0] - (pb - 1);
    const uint32_t mn = la < lb ? la : lb;
    for (uint32_t i = 2; i < mn; i++) {
      const uint32_t x = ba[pa + i], y = bb[pb + i];
      if (x != y) return x < y;
    }
    if (la != lb) return la < lb;
    return sa < sb;
  }
};

__global__ void k_make_sort_keys(const uint32_t* buf, const uint32_t* sent_pos, const uint32_t* suf_sid,
                                 const uint16_t* suf_pos, GpuSortKey* keys, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t sid = suf_sid[i], pos = suf_pos[i];
  const uint32_t* b = buf + sent_pos[sid];
  const uint32_t rem = b[0] - (pos - 1);
  GpuSortKey k;
  k.prefix = ((unsigned long long)b[pos] << 32) | (rem > 1 ? b[pos + 1] : 0u);
  k.key = ((unsigned long long)sid << 16) | pos;
  keys[i] = k;
}

__global__ void k_extract_keys(const GpuSortKey* keys, unsigned long long* out, unsigned long long n) {
  const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = keys[i].key;
}

void gpu_suffix_sort(HostIndex& h, int device) {
  if (gpu_device_count() <= device || device < 0) throw GpuError("no usable gfx950 device " + std::to_string(device));
  HIP_TRY(hipSetDevice(device));
  const uint64_t n = h.suf_sid.size();
  if (n == 0) { h.adopt_sorted(nullptr, 0); return; }
  uint32_t* d_buf = dev_upload(h.buf);
  uint32_t* d_sp = dev_upload(h.sent_pos);
  uint32_t* d_sid = dev_upload(h.suf_sid);
  uint16_t* d_pos = dev_upload(h.suf_pos);
  GpuSortKey* d_in = dev_alloc<GpuSortKey>(n);
  GpuSortKey* d_out = dev_alloc<GpuSortKey>(n);
  hipLaunchKernelGGL(k_make_sort_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_buf, d_sp, d_sid, d_pos, d_in, n);
  size_t tmp_bytes = 0;
  GpuSuffixLess less{d_buf, d_sp};
  HIP_TRY(rocprim::merge_sort(nullptr, tmp_bytes, d_in, d_out, (size_t)n, less, (hipStream_t)0, false));
  void* tmp = nullptr;
  HIP_TRY(hipMalloc(&tmp, std::max<size_t>(tmp_bytes, 1)));
  HIP_TRY(rocprim::merge_sort(tmp, tmp_bytes, d_in, d_out, (size_t)n, less, (hipStream_t)0, false));
  unsigned long long* d_flat = (unsigned long long*)d_in;  // reuse
  hipLaunchKernelGGL(k_extract_keys, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_out, d_flat, n);
  std::vector<uint64_t> flat(n);
  HIP_TRY(hipMemcpy(flat.data(), d_flat, n * 8, hipMemcpyDeviceToHost));
  hipFree(tmp); hipFree(d_in); hipFree(d_out); hipFree(d_buf); hipFree(d_sp); hipFree(d_sid); hipFree(d_pos);
  h.adopt_sorted(flat.data(), n);
}

}  // namespace fmx
