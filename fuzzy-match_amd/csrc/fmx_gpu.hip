// fmx_gpu.hip — device residency of the index and the batched match() pipeline.
//
// HBM layout (immutable, SoA, replicated per GPU — SURVEY §8e): see DESIGN.md §3.
//
// Re-entrancy (the reference's match() is const and is called from N threads on one instance,
// cli/src/FuzzyMatch-cli.cc:139-147): the index arrays are immutable after upload(); everything a call
// mutates — workspace, staging buffers, streams, events — lives in a Ctx taken from a small pool, so
// concurrent calls overlap on the GPU instead of serialising on a lock.
#include "fmx_gpu.hh"

#include <string.h>
#include <cstring>
#include <functional>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <stdexcept>
#include <string>

#include "fmx_kernels.hip.hh"
#include "fmx_light.hip.hh"
#include "fmx_subseq.hip.hh"
#include "fmx_small.hip.hh"
#include "fmx_derive.hip.hh"
#include "fmx_vocab.hip.hh"

namespace fmx {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      throw GpuError(std::string(#expr) + ": " + hipGetErrorString(e_));                            \
  } while (0)

static bool is_gfx950(int ordinal) {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, ordinal) != hipSuccess) { (void)hipGetLastError(); return false; }
  return std::string(prop.gcnArchName).rfind("gfx950", 0) == 0;
}

// number of HIP ordinals; ordinals that are not gfx950 are rejected individually by check_device()
int gpu_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  int usable = 0;
  for (int d = 0; d < n; d++) usable += is_gfx950(d) ? 1 : 0;
  return usable;
}

// `device` is a HIP ordinal: it must exist and be a gfx950 part itself (a lower ordinal of another
// architecture does not make a higher one usable, and vice versa)
void gpu_check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
  if (device < 0 || device >= n || !is_gfx950(device))
    throw GpuError("no usable gfx950 device at HIP ordinal " + std::to_string(device));
}

void* gpu_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, std::max<size_t>(bytes, 1), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
void gpu_host_free(void* p) { if (p) (void)hipHostFree(p); }

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

// hipFree waits for EVERYTHING in flight on the device.  A workspace buffer that grows while other calls are in flight (the first
// calls on an index, a batch with a heavier query than any before) used to free its old block on the spot: every one of the ~10
// buffers of a context then cost a device-wide wait, and the other contexts followed as the learned sizes spread.  Outgrown blocks
// are parked instead and freed when the process is quiet: at the destruction of an index, or — should they ever add up to more
// than kParkedMax — after one explicit synchronisation.  (This was suspected of being the multi-second stall with five and more
// host-buffer calls in flight; it was not — that one is the GPU's scheduler time-slicing more than about a dozen recently active
// hardware queues, DESIGN.md §4 — but a growing context no longer stops the others.)
struct ParkedBlocks {
  std::mutex mu;
  std::vector<std::pair<void*, uint64_t>> blocks;
  uint64_t bytes = 0;
  static constexpr uint64_t kParkedMax = 8ull << 30;  // per device (round 4: 48 GiB for the whole process, freed under whichever device was current)
  void park(void* p, uint64_t n) {
    std::vector<std::pair<void*, uint64_t>> drop;
    {
      std::lock_guard<std::mutex> g(mu);
      blocks.emplace_back(p, n);
      bytes += n;
      if (bytes > kParkedMax) { drop.swap(blocks); bytes = 0; }
    }
    // (the current device is the owner's: a context only grows its buffers under hipSetDevice of its index's device)
    if (!drop.empty()) { (void)hipDeviceSynchronize(); for (auto& b : drop) (void)hipFree(b.first); }
  }
  bool empty() { std::lock_guard<std::mutex> g(mu); return blocks.empty(); }
  void drain() {  // the caller made its device current; nothing of this process is using the blocks any more (hipFree waits for the device)
    std::vector<std::pair<void*, uint64_t>> drop;
    { std::lock_guard<std::mutex> g(mu); drop.swap(blocks); bytes = 0; }
    for (auto& b : drop) (void)hipFree(b.first);
  }
};
// one list per device ordinal: a block is parked, and later freed, under the device that owns it
static ParkedBlocks& parked_blocks() {
  static ParkedBlocks* pb = new ParkedBlocks[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
  return pb[(dev >= 0 && dev < 64) ? dev : 0];
}

// device buffer that only ever grows, by at least 1.5x (first-call and heavy-tail latency cliffs: one buffer
// growing never frees the others); the outgrown block is parked, not freed (see ParkedBlocks)
template <class T> struct Grow {
  T*       p = nullptr;
  uint64_t cap = 0;
  bool ensure(uint64_t need) {  // true if reallocated
    if (need <= cap && p) return false;
    const uint64_t c = std::max<uint64_t>(need, cap + cap / 2);
    if (p) parked_blocks().park(p, cap * sizeof(T));
    p = nullptr; cap = 0;
    p = dev_alloc<T>(c);
    cap = c;
    return true;
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// device allocations owned by whoever holds the last reference (see Impl::tab_mu)
struct DevBlob {
  void* a = nullptr;
  void* b = nullptr;
  ~DevBlob() { if (a) (void)hipFree(a); if (b) (void)hipFree(b); }
};

// The streams of a process, per device (round 5).  A hardware queue costs even when it idles: the GPU's scheduler time-slices
// the queues of a process once more than about a dozen are in use or were in use a moment ago, and every kernel of every call then
// runs 30-100 x slower (DESIGN.md section 5 "Stalls"; seen again in the test suite, where the contexts of indexes EARLIER tests had
// left behind slowed a fresh index down thirty-fold).  Until round 4 every context of every index owned two compute streams and
// every index a pair of copy streams, so the number of queues grew with the "contexts" option and with the number of indexes alive.
// Now a context BORROWS its pair of compute streams (main lane: normal priority, side lane: high, see Ctx::init) for the length
// of a call from this pool, and every index of a device uses the pool's one pair of copy streams: the queues of a process are
// bounded by the calls that run at once — which admission control (Impl::max_host_active, "contexts") bounds — not by how many
// contexts or indexes exist.  A pair goes back idle (its call was waited for) and is handed out again as it is.
struct StreamPool {
  std::mutex mu;
  // Host-buffer calls of a device START at least `spacing` apart (round 5).  Five or six of them begun back to back on an idle
  // GPU — a caller filling its pipeline — made the FIRST of them take 13-340 ms instead of ~10 (tools/dbg/text_host_time.py: the
  // kernels run at their usual speed, the GPU idles in between); begun half a millisecond apart they never did.  In the steady
  // state calls begin a step apart and nothing waits.
  std::mutex pace_mu;
  std::chrono::steady_clock::time_point last_host_begin{};
  void pace(std::chrono::microseconds spacing) {
    if (spacing.count() <= 0) return;
    std::lock_guard<std::mutex> g(pace_mu);
    const auto due = last_host_begin + spacing;
    if (std::chrono::steady_clock::now() < due) std::this_thread::sleep_until(due);
    last_host_begin = std::chrono::steady_clock::now();
  }
  std::vector<std::pair<hipStream_t, hipStream_t>> idle;
  hipStream_t h2d = nullptr, d2h = nullptr;
  bool copy_tried = false;
  std::pair<hipStream_t, hipStream_t> take() {
    {
      std::lock_guard<std::mutex> g(mu);
      if (!idle.empty()) { auto p = idle.back(); idle.pop_back(); return p; }
    }
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    static const bool classes = [] { const char* e = getenv("FMX_STREAM_CLASSES"); return e ? atoi(e) != 0 : true; }();
    // Round 6 — the "cliff" of rounds 2-5, reproduced without the library (tools/microbench/queues.hip, profiles/r06_queues_cliff.md):
    // calls in flight that each bring a HIGH-PRIORITY stream collapse (4-7 ms per step -> 160-430 ms) as soon as FIVE or more such
    // streams are in use beside the normal class's eight hardware queues — six calls in the bare repro, seven in the library's shape;
    // with the same streams at normal priority, or ONE high-priority stream shared by all calls, twelve calls in flight take what four
    // take.  Every priority class has a pool of hardware queues of its own, and the firmware scheduler time-slices a process that keeps
    // more than about a dozen queues busy.  Until round 5 every call's side lane was high-priority ("so that main and side lane never
    // share a queue"): that was the cliff.  Now side lanes are normal-priority streams (FMX_HIGH_PRIO_LANES = how many pairs still get
    // a high one: 0): sixteen uncapped synchronous callers take 1.6 ms per call instead of 160-1000 (capped at four: 0.67), and
    // `value` / the device-resident leg are the same or 1-2 % better.
    // (End of round 6: ONE pair keeps a high-priority side lane, and it is the pair handed out first — `give` puts it on top.  Two
    // normal-priority lanes can land on one hardware queue and then take turns for as long as the pair lives: in two C3 runs out of
    // six the one-call-at-a-time leg ran the hit walk of the light queries in 1.92 ms instead of 1.34 with k_expand_b<false> "beside"
    // it in 0.41 instead of 0.58, and the finish stage in 0.61 instead of 0.41 — the sums of the lanes, 5.76 ms per pass instead of
    // 4.93.  A caller with one call at a time always gets the pair whose lanes cannot share a queue; two high-priority queues in use
    // — this one and the id look-up's — are far from the five of the cliff.  Testing a fresh pair with a spinning kernel instead
    // stalled the bench's text leg in four runs out of four: removed.)
    static const int high_lanes = [] { const char* e = getenv("FMX_HIGH_PRIO_LANES"); return e ? atoi(e) : 1; }();
    hipStream_t a = nullptr, b = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    bool high = false;
    { std::lock_guard<std::mutex> g(mu); high = classes && n_high < high_lanes; if (high) n_high++; }
    if (high) {
      HIP_TRY(hipStreamCreateWithPriority(&b, hipStreamNonBlocking, hi));
      std::lock_guard<std::mutex> g(mu);
      high_side.push_back(b);
    } else HIP_TRY(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    return {a, b};
  }
  int n_high = 0;  // pairs made with a high-priority side lane
  std::vector<hipStream_t> high_side;  // ... their side lanes
  void give(std::pair<hipStream_t, hipStream_t> p) {
    std::lock_guard<std::mutex> g(mu);
    // `take` pops the back: a pair with a high-priority side lane goes there, the others below the high ones
    if (std::find(high_side.begin(), high_side.end(), p.second) != high_side.end()) { idle.push_back(p); return; }
    size_t n_top = 0;
    while (n_top < idle.size() && std::find(high_side.begin(), high_side.end(), idle[idle.size() - 1 - n_top].second) != high_side.end()) n_top++;
    idle.insert(idle.end() - (std::ptrdiff_t)n_top, p);
  }
  // the device's pair of copy streams (low priority: a pool of hardware queues of its own); false: the caller makes its own
  bool copy_pair(hipStream_t& out_h2d, hipStream_t& out_d2h) {
    std::lock_guard<std::mutex> g(mu);
    if (!copy_tried) {
      copy_tried = true;
      int lo = 0, hi = 0;
      hipStream_t a = nullptr, b = nullptr;
      if (hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hipStreamCreateWithPriority(&a, hipStreamNonBlocking, lo) == hipSuccess &&
          hipStreamCreateWithPriority(&b, hipStreamNonBlocking, lo) == hipSuccess) { h2d = a; d2h = b; }
      else { if (a) (void)hipStreamDestroy(a); (void)hipGetLastError(); }
    }
    out_h2d = h2d; out_d2h = d2h;
    return h2d != nullptr;
  }
};
static StreamPool& stream_pool(int device) {
  static StreamPool* pools = new StreamPool[64];
  return pools[(device >= 0 && device < 64) ? device : 0];
}

constexpr int kMaxPassesInFlight = 2;
static const uint64_t kTurnMinQueries = [] { const char* e = getenv("FMX_TURN_MIN"); return e ? (uint64_t)atoll(e) : (uint64_t)262144; }();  // device-resident calls at least this large are chained on the GPU (Impl::turns), smaller ones overlap freely (design-time switch FMX_TURN_MIN)
static const uint64_t kTurnMinQueriesHost = [] { const char* e = getenv("FMX_TURN_MIN_HOST"); return e ? (uint64_t)atoll(e) : (uint64_t)32768; }();  // host-buffer calls: chained from here on (their copies interleave badly when the kernels of
                                                 // 100 k-query calls overlap freely: 46 M patterns/s against 86 M chained at C2)
constexpr int kScanSlots = 5;  // scans per pass: queue capacities, queued hits, candidate classes, contrast lists, pairs left to the float edit distance

// Everything one call mutates.
struct Ctx {
  // ---- workspace ----
  Grow<uint32_t> q_m, q_kept, q_cand, q_acc, q_pb, q_rank, q_scap, big_list, mid_list, fin_list, contrast_list, fb_list;
  Grow<unsigned long long> q_hits, q_hits_a, q_seg, q_qseg, q_cap, c_in, c_off, q_best;
  Grow<uint32_t> t_q, t_lo, t_cnt;
  Grow<float>    t_idf;
  Grow<unsigned long long> c_qi, keys;
  Grow<Cand>     seg;
  Grow<QHit>     qb;
  Grow<float>    heap, f0, f1;
  Grow<uint32_t> vals;
  // zeroed by one memset per pass: [counters | scan state + tickets | p_hist]
  Grow<unsigned long long> zero_blk;
  unsigned long long* counters_p() { return zero_blk.p; }
  unsigned long long* scan_state(int slot) { return zero_blk.p + kNumCounters + (size_t)slot * kLbMaxArrays * kLbMaxTiles; }
  uint32_t* scan_ticket(int slot) { return reinterpret_cast<uint32_t*>(zero_blk.p + kNumCounters + (size_t)kScanSlots * kLbMaxArrays * kLbMaxTiles) + slot; }
  uint32_t* p_hist_p() { return scan_ticket(0) + 2 * kScanSlots; }
  static constexpr size_t zero_words() { return kNumCounters + (size_t)kScanSlots * kLbMaxArrays * kLbMaxTiles + kScanSlots + kPBins + 2; }
  Grow<WorkItem> items, items_b;
  Grow<HeavyItem> items_h;
  Grow<ProbeItem> items_t;   // start positions queued by k_probe for k_probe_tail (one slot per token of the pass)
  Grow<uint16_t> cmin_tab;
  // what the table in cmin_tab was computed for (k_cmin is skipped while a pass asks for the same: the coalescer's
  // micro-batches repeat their parameters, and the kernel is 25 us of a ~250 us pass)
  float    cmin_key[4] = {0, 0, 0, 0};
  uint32_t cmin_pmax = 0, cmin_maxslen = 0;
  int      cmin_sig = -1;
  const uint16_t* cmin_ptr = nullptr;
  Grow<float>    g_rows;      // DP rows in HBM for patterns / sentences too long for the LDS rows
  uint32_t s_max = 4096;      // survivors kept per query before a retry with a larger segment
  uint32_t heavy_div = 16;    // share of a query's long-range hits its queue reserves room for (1 / heavy_div)
  uint64_t seg_cap = 0;
  uint64_t want_qb = 1ull << 24;  // queue entries (8 bytes each) wanted: grows when a pass reports more
  // ---- staging of the host-buffer entry point (each buffer tracks its own capacity) ----
  Grow<uint32_t> d_qoff, d_qtok, d_qreal, d_qitok, d_cnt, d_tot, d_prior;
  Grow<unsigned char> d_text, d_text_r;    // token spellings of a text batch (normalised tokens; real tokens)
  Grow<unsigned long long> d_vstate, d_vstate_r;  // spellings per tile of k_vocab_count, scanned in place
  Grow<unsigned long long> d_islot;        // penalty tokens of a text batch: slot ...
  Grow<uint32_t> d_iid;                    // ... and interned id
  Grow<uint32_t> d_scratch;                // a few scratch words (longest pattern of a line batch)
  Grow<unsigned char> d_lflag;             // raw-text batches: one byte per line, set by the device tokenizer for lines it does not cover
  unsigned long long* h_vtotal = nullptr;  // pinned: tokens k_vocab_lookup found
  Grow<Rec>      d_out;
  // ---- streams / events ----
  hipStream_t stream = nullptr, h2d = nullptr, d2h = nullptr, side = nullptr;
  hipEvent_t  ev[2 * kNumStages]{};   // [2 s], [2 s + 1]: start / end of stage s (the last stage's pair lives on the side stream)
  hipEvent_t  ev_fork = nullptr, ev_join = nullptr, ev_cmin = nullptr, ev_a = nullptr, ev_ids = nullptr, ev_in[kMaxPassesInFlight]{}, ev_done = nullptr;
  hipEvent_t  ev_chain = nullptr;  // the last kernel of a call's first group (what the next call waits for on the GPU: a plain event —
                                   // the ev_pass events are made for a sleeping host and are slow to wait for from another stream)
  hipEvent_t  ev_mid = nullptr;  // the memory-bound half of a pass (probe, long ranges) is through: the next call's probe may start (turns = 2)
  unsigned long long* h_ctr = nullptr;  // pinned: the pass counters land here
  char* h_small = nullptr; char* d_small = nullptr; size_t h_small_bytes = 0;  // pinned block of the one-launch path (host / device view)
  unsigned long long* h_ctr_group = nullptr;  // pinned: one slot per pass of an asynchronous group
  std::vector<hipEvent_t> ev_pass;      // "pass done" events of an asynchronous group
  bool ok = false;
  bool shared_copy = false;  // h2d / d2h belong to the device's stream pool (one pair for every context of every index), not to this context
  bool pooled = false;       // stream / side are borrowed from the device's stream pool for the length of a call (null in between)
  int  cu_group = -1;  // >= 0: the kernels of this context run on one half of the GPU's compute units (option "cu_split")

  // A stream whose kernels only run on half of the compute units: hipExtStreamCreateWithCUMask (bit i of the mask = compute
  // unit i in the enumeration the runtime spreads round-robin over the XCDs).  mode 1: the lower / upper half of the bits —
  // half of the CUs of EVERY XCD; mode 2: bits by XCD parity — four whole XCDs (each with its own L2) per group.
  static void create_masked(hipStream_t* st, int group, int mode) {
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    const uint32_t n_cu = (uint32_t)prop.multiProcessorCount;
    std::vector<uint32_t> mask((n_cu + 31) / 32, 0u);
    for (uint32_t i = 0; i < n_cu; i++) {
      const bool mine = mode == 2 ? ((i % 8) / 4 == (uint32_t)group) : ((i < n_cu / 2) == (group == 0));
      if (mine) mask[i / 32] |= 1u << (i % 32);
    }
    HIP_TRY(hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data()));
  }

  void init(hipStream_t shared_h2d = nullptr, hipStream_t shared_d2h = nullptr) {
    // HIP maps streams onto a few hardware queues (four per priority class by default) and everything on one hardware queue
    // runs in order: when `stream` and `side` land on the same queue the kernels meant to run side by side take turns (−9 %,
    // seen in about one process out of four), and a copy of another call waiting on a compute stream's queue stalls its
    // kernels.  The priority classes have separate queue pools, so the three roles get three classes: kernels of the main
    // lane normal, of the side lane high, copies low.  (The class itself does not change which workgroups are placed
    // first: measured.)
    int prio_low = 0, prio_high = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
    const bool classes = [] { const char* e = getenv("FMX_STREAM_CLASSES"); return e ? atoi(e) != 0 : true; }();
    const int cu_mode = [] { const char* e = getenv("FMX_CU_SPLIT"); return e ? atoi(e) : 0; }();
    if (cu_mode && cu_group >= 0) {
      create_masked(&stream, cu_group, cu_mode);
      create_masked(&side, cu_group, cu_mode);
      HIP_TRY(hipStreamCreateWithPriority(&h2d, hipStreamNonBlocking, prio_low));
      HIP_TRY(hipStreamCreateWithPriority(&d2h, hipStreamNonBlocking, prio_low));
    } else {
    static const bool pool_streams = [] { const char* e = getenv("FMX_STREAM_POOL"); return e ? atoi(e) != 0 : true; }();  // design-time switch
    pooled = pool_streams;  // the compute pair comes with the lease (Impl::acquire)
    if (!pooled) HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (classes) {
      if (shared_h2d && shared_d2h) { h2d = shared_h2d; d2h = shared_d2h; shared_copy = true; }
      else {
        HIP_TRY(hipStreamCreateWithPriority(&h2d, hipStreamNonBlocking, prio_low));
        HIP_TRY(hipStreamCreateWithPriority(&d2h, hipStreamNonBlocking, prio_low));
      }
      if (!pooled) HIP_TRY(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_high));
    } else {
      HIP_TRY(hipStreamCreateWithFlags(&h2d, hipStreamNonBlocking));
      HIP_TRY(hipStreamCreateWithFlags(&d2h, hipStreamNonBlocking));
      if (!pooled) HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
    }
    }
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_cmin, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_ids, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_mid, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_chain, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming | hipEventBlockingSync));  // the host sleeps on it (see ev_pass)
    for (auto& e : ev_in) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIP_TRY(hipHostMalloc((void**)&h_ctr, kNumCounters * sizeof(unsigned long long), hipHostMallocDefault));
    zero_blk.ensure(zero_words());
    ok = true;
  }
  void destroy() {
    q_m.release(); q_kept.release(); q_cand.release(); q_acc.release(); q_pb.release(); q_rank.release(); q_scap.release(); q_best.release();
    big_list.release(); mid_list.release(); fin_list.release(); contrast_list.release();
    q_hits.release(); q_hits_a.release(); q_seg.release(); q_qseg.release(); q_cap.release(); c_in.release(); c_off.release();
    t_q.release(); t_lo.release(); t_cnt.release(); t_idf.release(); c_qi.release(); keys.release(); seg.release(); qb.release();
    heap.release(); f0.release(); f1.release(); vals.release(); zero_blk.release();
    items.release(); items_b.release(); items_h.release(); items_t.release(); cmin_tab.release(); g_rows.release();
    d_qoff.release(); d_qtok.release(); d_qreal.release(); d_qitok.release(); d_text.release(); d_vstate.release(); d_text_r.release(); d_vstate_r.release(); d_islot.release(); d_iid.release(); d_scratch.release(); d_lflag.release(); d_cnt.release(); d_tot.release(); d_out.release(); d_prior.release();
    if (stream && !pooled) (void)hipStreamDestroy(stream);
    if (h2d && !shared_copy) (void)hipStreamDestroy(h2d);
    if (d2h && !shared_copy) (void)hipStreamDestroy(d2h);
    if (side && !pooled) (void)hipStreamDestroy(side);
    for (auto& e : ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : ev_in) if (e) (void)hipEventDestroy(e);
    if (ev_fork) (void)hipEventDestroy(ev_fork);
    if (ev_join) (void)hipEventDestroy(ev_join);
    if (ev_cmin) (void)hipEventDestroy(ev_cmin);
    if (ev_ids) (void)hipEventDestroy(ev_ids);
    if (ev_a) (void)hipEventDestroy(ev_a);
    if (ev_mid) (void)hipEventDestroy(ev_mid);
    if (ev_chain) (void)hipEventDestroy(ev_chain);
    if (ev_done) (void)hipEventDestroy(ev_done);
    if (h_ctr) (void)hipHostFree(h_ctr);
    if (h_vtotal) (void)hipHostFree(h_vtotal);
    if (h_small) (void)hipHostFree(h_small);
    if (h_ctr_group) (void)hipHostFree(h_ctr_group);
    for (auto& e : ev_pass) if (e) (void)hipEventDestroy(e);
  }
};

struct DeferredCall;

struct DeviceIndex::Impl {
  int      device = -1;
  uint64_t n_tok_words = 0;  // words of m.d.tok (device layout of the sentence buffer, without the eight spare words)
  DevIndex d{};
  uint64_t bytes = 0;
  std::atomic<uint64_t> itok_version{~0ull};
  // The two tables that may be replaced while calls are in flight — the penalty-string distance table (a query interned a new
  // string) and the device copy of the spelling table (words were added) — are reference-counted: a call pins the pair it was
  // enqueued with (pin_tables), a refresh swaps the pointers under tab_mu and the old memory is freed when its last call is
  // through.  No lock is held across a call: a thread with calls pending (fmx_match_batch_device_begin) may intern a string,
  // issue its first text batch or refresh a table without deadlocking against itself, and never starves another thread's refresh.
  std::mutex              tab_mu;
  std::mutex              refresh_mu;   // one refresh at a time (the upload happens outside tab_mu)
  std::shared_ptr<DevBlob> itok_hold, vocab_hold;
  // options
  uint32_t sub_queries = 1u << 20;       // queries per pipeline pass: large passes amortise the ~40 launches and their tails
  uint32_t host_sub_queries = 1u << 20;  // pass size of the host-buffer entry point (see match_host)
  // Host-buffer calls take turns on the GPU: kernels of several calls enqueued at once advance in step (each stream's next
  // kernel waits for its own predecessor, the GPU alternates between the streams), all calls finish together, copy their
  // records home together and fetch their next inputs together — with the GPU idle meanwhile (measured: 26 % of the time
  // with three calls in flight).  One call at a time owns the kernels, and the copies of the calls before and after run
  // beside it.  (Letting the next call in earlier, under the owner's last kernels, does not pay: workgroups of the oldest
  // kernel are placed first, so the newcomer's probe starves the owner's tail and the two finish together again.)
  std::mutex              turn_mu;
  std::condition_variable turn_cv;
  int                     turn_busy = 0;
  uint32_t                turns = 2;     // 0 = no turns; 1 = chained on the GPU (each call's first kernel waits for the event that ends
                                         // the previous call's kernels: no hand-over through the host); 2 = chained behind the
                                         // predecessor's memory-bound half (its probe and long-range kernels): the probe of a call,
                                         // bound by HBM, runs beside the hit walk / edit distances / replay of the call before it,
                                         // bound by instruction issue and latency; 3 = turns taken on the host
  uint32_t                device_turns = 0;  // the same for the device-resident entry points.  0 by default: with eight hardware
                                         // queues (GPU_MAX_HW_QUEUES, set at load) six unchained calls overlap better than any chaining
                                         // (C3 5.59 against 5.82 ms per step, C4 3.74 against 4.08), and with the runtime's default of four
                                         // queues and three calls in flight chaining made no difference (5.89 / 5.86)
  std::mutex              chain_mu;
  hipEvent_t              chain_last = nullptr;  // ends the kernels of the call enqueued last (owned by its context)
  // what passes needed so far, per 1024 queries, shared by all contexts: a context sizes its workspace for a full pass of the
  // call at hand up front (a pass that does not fit is re-run after a reallocation that stalls every call in flight)
  std::atomic<uint64_t> hint_seg_kq{0}, hint_qb_kq{0};
  // ... and the per-query limits a context learned the hard way (a re-run): the other contexts adopt them at their next call
  // instead of each re-running a pass of its own when the batch with the heaviest query reaches it
  std::atomic<uint32_t> hint_s_max{0}, hint_heavy_div{0};
  uint64_t init_seg = 1ull << 24;
  uint32_t heavy_min = kHeavyMinDefault; // suffix ranges at least this long go through the bit-plane kernel (0 = never)
  uint32_t max_ctx = 3;
  bool     probe_count = false;
  bool     serial = false;        // measurement runs: no side stream, every kernel alone on the GPU
  uint32_t expand_order = 1, b_grid = kExpandExtraGrid;
  // micro-batches in one kernel launch (fmx_small.hip.hh); FMX_SMALL_PATH=0 in the environment starts with it off (the test
  // suite does, so that its many small batches keep exercising the pipeline; tests/test_gpu_small.py turns it on)
  bool     pair_finish = [] { const char* e = getenv("FMX_PAIR_FINISH"); return e ? atoi(e) != 0 : true; }();  // design-time switch
  bool     small_path = [] { const char* e = getenv("FMX_SMALL_PATH"); return e ? atoi(e) != 0 : true; }();
  // real-token indexes, unit costs: bit-parallel kernel for the pairs without a real-token mismatch + float kernel for the rest, instead
  // of the float kernel for all (round 6: built, bit-identical, slower at C3 — profiles/r06_real_hybrid_ab.md —: off unless asked for)
  bool     real_hybrid = [] { const char* e = getenv("FMX_REAL_HYBRID"); return e ? atoi(e) != 0 : false; }();
  uint32_t heavy_grid = kHeavyGrid;      // workgroups of k_expand_h (it runs beside the light queries' kernel: it must not fill every wave slot)
  uint32_t light_max = kLightMaxDefault;  // queries with at most this many hits, all in short ranges, skip phase A (0 = off)
  // of those, queries with at most this many hits are walked several to a wave (k_expand_lf, round 5; 0 = all by k_expand_b<true>).
  // OFF by default: bit-identical, 13 % fewer VALU instructions, and SLOWER at C3 — the stage 1.40 -> 1.81 ms (flat 128), 1.59 (64):
  // at four waves per SIMD (118 VGPRs, 8 KB of LDS) the sub-tiles' chains of dependent LDS / memory round trips are not hidden
  // (profiles/r05_light_flat_ab.md)
  // the probe as two kernels (head: one table line per start position; tail: the bisection of what is left, in full waves)
  bool probe_split = [] { const char* e = getenv("FMX_PROBE_SPLIT"); return e ? atoi(e) != 0 : true; }();
  uint32_t light_flat = [] { const char* e = getenv("FMX_LIGHT_FLAT"); return e ? (uint32_t)std::min(std::max(atoi(e), 0), 256) : 0u; }();
  // context pool
  // One pair of copy streams for every context of the index: the DMA engines serialise the copies anyway, and a context then
  // brings two streams (hardware queues) of its own instead of four.  Measured with six contexts (14 queues instead of 24):
  // C3 PCIe-inclusive 6.49 -> 5.88 ms per step with four callers, and the device-resident calls — which never touch a copy
  // stream — gain too (C3 5.42 -> 5.35 ms, C2's 100 k-query calls 0.78 -> 0.67 ms): queues cost even when they idle.
  // (FMX_SHARED_COPY=0: a pair per context, as before.)
  hipStream_t             copy_h2d = nullptr, copy_d2h = nullptr;
  // device copy of the host's spelling table (text batches: fmx_vocab.hip.hh); replaced (under tab_mu) when the vocabulary
  // changed; a call reads the copy it pinned
  DevVocab                vocab{};
  uint64_t                vocab_built_for = ~0ull;
  // the index's interned real tokens as the same kind of table (HostIndex::real_table): fixed at upload (real tokens are
  // interned by derive()); only indexes whose real tokens differ from the normalised ones have one
  DevVocab                real_vocab{};
  DevBlob                 real_hold;
  std::mutex              word_mu;     // scratch words of sub_bounds_from_device
  std::vector<uint32_t*>  word_pool;
  std::mutex              pool_mu;
  std::condition_variable pool_cv;
  std::vector<std::unique_ptr<Ctx>> all;
  std::vector<Ctx*>       idle;
  // Admission (round 5).  How many calls may have work on the GPU at once is a property of the GPU, not of how many contexts a
  // caller asked for: every host-buffer call in flight keeps two compute queues and the index's pair of copy queues busy, and
  // with eight of them (sixteen compute queues) every kernel of every call crawls (600-900 ms per step at C3).  So at
  // most max_host_active host-buffer calls hold a context at any time whatever "contexts" says.  Four by default; five and six
  // run as well once their starts are paced (StreamPool::pace; sixteen synchronous callers capped at five or six take the time of
  // four, tests/test_gpu_text_batch.py with the test process's garbage collector held off), seven and more crawl.  bench.py raises
  // the option to five for its one-thread pipeline of _begin / _end calls;
  uint32_t                max_host_active = [] { const char* e = getenv("FMX_MAX_HOST_ACTIVE"); return e ? (uint32_t)std::max(atoi(e), 1) : 4u; }();
  uint32_t                host_active = 0;
  std::deque<DeferredCall*> deferred;

  bool may_take(bool host_kind) const {  // under pool_mu
    return (!host_kind || host_active < max_host_active) && (!idle.empty() || all.size() < max_ctx);
  }
  // wait = false: nullptr instead of waiting when no context may be taken (check and take are one critical section).
  // force: one more context than the pool's limit rather than a dead-lock (a deferred call ended before any slot was freed).
  Ctx* acquire(bool wait = true, bool host_kind = false, bool force = false) {
    std::unique_lock<std::mutex> lk(pool_mu);
    for (;;) {
      const bool admitted = force || !host_kind || host_active < max_host_active;
      if (admitted && !idle.empty()) {
        Ctx* c = idle.back(); idle.pop_back();
        if (host_kind) host_active++;
        lk.unlock();
        try { lend_streams(c); } catch (...) { lk.lock(); idle.push_back(c); if (host_kind) host_active--; pool_cv.notify_all(); throw; }
        return c;
      }
      if (admitted && (all.size() < max_ctx || force)) {
        if (host_kind) host_active++;
        all.emplace_back(new Ctx);
        Ctx* c = all.back().get();
        c->cu_group = (int)((all.size() - 1) % 2);
        static const bool share = [] { const char* e = getenv("FMX_SHARED_COPY"); return e ? atoi(e) != 0 : true; }();
        // one pair of copy streams per DEVICE (round 4: per index), from the device's stream pool; contexts fall back to a pair of their own
        if (share && !copy_h2d) (void)stream_pool(device).copy_pair(copy_h2d, copy_d2h);
        const hipStream_t sh = copy_h2d, sd = copy_d2h;
        lk.unlock();
        try { c->init(sh, sd); lend_streams(c); } catch (...) { lk.lock(); idle.push_back(c); if (host_kind) host_active--; pool_cv.notify_all(); throw; }
        return c;
      }
      if (!wait) return nullptr;
      pool_cv.wait(lk);
    }
  }
  // the compute streams of a call: borrowed from the device's pool with the context, given back with it
  void lend_streams(Ctx* c) {
    if (!c->pooled || c->stream) return;
    const auto p = stream_pool(device).take();
    c->stream = p.first; c->side = p.second;
  }
  void release(Ctx* c, bool host_kind = false) {
    if (c->pooled && c->stream) {  // (the call was waited for — or ~HostCall / ~Pending waited for what was left of it: the pair is idle)
      stream_pool(device).give({c->stream, c->side});
      c->stream = nullptr; c->side = nullptr;
    }
    bool quiet;
    {
      std::lock_guard<std::mutex> lk(pool_mu);
      idle.push_back(c);
      if (host_kind && host_active) host_active--;
      quiet = idle.size() == all.size() && deferred.empty();
    }
    // a quiet point of the index (no call holds a context): what its contexts outgrew is freed now, not kept until the index goes
    if (quiet && device >= 0 && !parked_blocks().empty()) parked_blocks().drain();
    pool_cv.notify_all();  // (waiters of both kinds, and _end calls waiting for a deferred start)
    start_deferred();
  }
  void start_deferred() noexcept;
};

// thrown by a lease that may not wait: the call is deferred, not refused
struct NoContext : std::runtime_error { NoContext() : std::runtime_error("no context may be taken now") {} };

struct CtxLease {
  DeviceIndex::Impl& m;
  Ctx* c;
  bool host_kind;
  explicit CtxLease(DeviceIndex::Impl& m_, bool no_wait = false, bool host_kind_ = false, bool force = false)
      : m(m_), c(m_.acquire(!no_wait, host_kind_, force)), host_kind(host_kind_) {
    if (!c) throw NoContext();
    if (!c->ok) { Ctx* bad = c; c = nullptr; m.release(bad, host_kind); throw GpuError("context initialisation failed earlier"); }
  }
  void release_now() { if (c) { Ctx* x = c; c = nullptr; m.release(x, host_kind); } }
  ~CtxLease() { release_now(); }
};

// A call begun through one of the _begin entry points when no context may be taken (every context busy, or the host-buffer calls
// at their limit): it waits on the host, in order, and is started by the thread that frees a slot — or, at the latest, by its own
// _end.  State changes under Impl::pool_mu, announced on pool_cv.
struct DeferredCall {
  enum State { kStarted, kQueued, kStarting, kFailed };
  State state = kStarted;
  bool host_kind = false;
  std::function<void(bool)> start;  // enqueues the call; the argument: take a context beyond the pool's limit if there is none
  std::exception_ptr err;
};

void DeviceIndex::Impl::start_deferred() noexcept {
  for (;;) {
    DeferredCall* d = nullptr;
    {
      std::lock_guard<std::mutex> lk(pool_mu);
      if (deferred.empty() || !may_take(deferred.front()->host_kind)) return;
      d = deferred.front();
      deferred.pop_front();
      d->state = DeferredCall::kStarting;
    }
    bool again = false;
    try { d->start(false); }
    catch (const NoContext&) { again = true; }  // another thread took the context in between: back to the head of the line
    catch (...) { d->err = std::current_exception(); }
    {
      std::lock_guard<std::mutex> lk(pool_mu);
      if (again) { d->state = DeferredCall::kQueued; deferred.push_front(d); }
      else d->state = d->err ? DeferredCall::kFailed : DeferredCall::kStarted;
    }
    pool_cv.notify_all();
    if (again) return;
  }
}

// _begin: run the call's start now, or queue it.  _end: make sure it has started (rethrows what the start threw).
static void begin_or_defer(DeviceIndex::Impl& m, DeferredCall& d) {
  {
    std::lock_guard<std::mutex> lk(m.pool_mu);
    if (!m.deferred.empty()) {  // calls are started in the order they were begun
      d.state = DeferredCall::kQueued;
      m.deferred.push_back(&d);
      return;
    }
  }
  try { d.start(false); d.state = DeferredCall::kStarted; }
  catch (const NoContext&) {
    { std::lock_guard<std::mutex> lk(m.pool_mu); d.state = DeferredCall::kQueued; m.deferred.push_back(&d); }
    m.start_deferred();  // (a slot may have been freed between the failed attempt and the queueing)
  }
}
static void ensure_started(DeviceIndex::Impl& m, DeferredCall& d) {
  std::unique_lock<std::mutex> lk(m.pool_mu);
  if (d.state == DeferredCall::kQueued) {
    for (auto it = m.deferred.begin(); it != m.deferred.end(); ++it)
      if (*it == &d) { m.deferred.erase(it); break; }
    d.state = DeferredCall::kStarting;
    lk.unlock();
    // the slots may all be held by pending calls of THIS thread, which it would end after this one: one context more than the
    // limit rather than a dead-lock
    try { d.start(true); } catch (...) { d.err = std::current_exception(); }
    lk.lock();
    d.state = d.err ? DeferredCall::kFailed : DeferredCall::kStarted;
    lk.unlock();
    m.pool_cv.notify_all();
  } else {
    m.pool_cv.wait(lk, [&] { return d.state == DeferredCall::kStarted || d.state == DeferredCall::kFailed; });
    lk.unlock();
  }
  if (d.err) { std::exception_ptr e = d.err; d.err = nullptr; std::rethrow_exception(e); }
}

DeviceIndex::DeviceIndex() : impl_(new Impl) {}

DeviceIndex::~DeviceIndex() {
  if (!impl_) return;
  Impl& m = *impl_;
  if (m.device >= 0) {
    (void)hipSetDevice(m.device);
    (void)hipDeviceSynchronize();
    hipFree((void*)m.d.suf); hipFree((void*)m.d.suf_sig); hipFree((void*)m.d.suf_key); hipFree((void*)m.d.gram.g3); hipFree((void*)m.d.gram.g5); hipFree((void*)m.d.gram.g7); hipFree((void*)m.d.suf_b); hipFree((void*)m.d.tok); hipFree((void*)m.d.sent_pos);
    hipFree((void*)m.d.qva); hipFree((void*)m.d.idf); hipFree((void*)m.d.real); hipFree((void*)m.d.itok);
    hipFree((void*)m.d.planes); hipFree((void*)m.d.thermo);
    for (auto& c : m.all) c->destroy();
    parked_blocks().drain();  // (the device was synchronised above)
    for (uint32_t* wp : m.word_pool) (void)hipFree(wp);
    m.word_pool.clear();
    m.itok_hold.reset(); m.vocab_hold.reset();
    m.copy_h2d = m.copy_d2h = nullptr;  // (they belong to the device's stream pool)
  }
  delete impl_;
}

int      DeviceIndex::device() const { return impl_->device; }
uint64_t DeviceIndex::bytes() const { return impl_->bytes; }
uint32_t DeviceIndex::sub_queries() const { return impl_->sub_queries; }
void     DeviceIndex::set_sub_queries(uint32_t n) { impl_->sub_queries = std::max<uint32_t>(n, 1); }
void     DeviceIndex::set_host_sub_queries(uint32_t n) { impl_->host_sub_queries = std::max<uint32_t>(n, 1); }
void     DeviceIndex::set_initial_segment(uint64_t n) { impl_->init_seg = std::max<uint64_t>(n, 1024); }
void     DeviceIndex::set_heavy_min(uint32_t n) { impl_->heavy_min = n; }
void     DeviceIndex::set_probe_count(bool on) { impl_->probe_count = on; }
void     DeviceIndex::set_serial(bool on) { impl_->serial = on; }
void     DeviceIndex::set_turns(uint32_t n) { impl_->turns = n; impl_->device_turns = n; }
void     DeviceIndex::set_device_turns(uint32_t n) { impl_->device_turns = n; }
void     DeviceIndex::set_small_path(bool on) { impl_->small_path = on; }
void     DeviceIndex::set_real_hybrid(bool on) { impl_->real_hybrid = on; }
void     DeviceIndex::set_pair_finish(bool on) { impl_->pair_finish = on; }
void     DeviceIndex::set_expand_order(uint32_t n) { impl_->expand_order = n; }
void     DeviceIndex::set_b_grid(uint32_t n) { impl_->b_grid = std::min<uint32_t>(std::max<uint32_t>(n, 64u), 65536u); }
void     DeviceIndex::set_heavy_grid(uint32_t n) { impl_->heavy_grid = std::min<uint32_t>(std::max<uint32_t>(n, 64u), 65536u); }
void     DeviceIndex::set_light_max(uint32_t n) { impl_->light_max = std::min<uint32_t>(n, kPhaseBChunk); }
void     DeviceIndex::set_probe_split(bool on) { impl_->probe_split = on; }
void     DeviceIndex::set_light_flat(uint32_t n) { impl_->light_flat = std::min<uint32_t>(n, 256u); }
void     DeviceIndex::set_contexts(uint32_t n) {
  {
    std::lock_guard<std::mutex> lk(impl_->pool_mu);
    impl_->max_ctx = std::max<uint32_t>(std::max<uint32_t>(n, 1), (uint32_t)impl_->all.size());
  }
  impl_->pool_cv.notify_all();
  impl_->start_deferred();
}
void     DeviceIndex::set_max_host_active(uint32_t n) {
  { std::lock_guard<std::mutex> lk(impl_->pool_mu); impl_->max_host_active = std::max<uint32_t>(n, 1); }
  impl_->pool_cv.notify_all();
  impl_->start_deferred();
}

// what a call pins for its lifetime: the penalty-string table and the spelling table it was enqueued with
struct TableSnap {
  std::shared_ptr<DevBlob> itok_hold, vocab_hold;
  const uint16_t* itok_dist = nullptr;
  uint32_t        n_itok = 0;
  DevVocab        vocab{};
};
static TableSnap pin_tables(DeviceIndex::Impl& m) {
  std::lock_guard<std::mutex> g(m.tab_mu);
  TableSnap s;
  s.itok_hold = m.itok_hold; s.vocab_hold = m.vocab_hold;
  s.itok_dist = m.d.itok_dist; s.n_itok = m.d.n_itok; s.vocab = m.vocab;
  return s;
}
// the index's device view as a call sees it: the immutable arrays + the pinned tables
static DevIndex dev_view(const DeviceIndex::Impl& m, const TableSnap& s) {
  DevIndex d = m.d;
  d.itok_dist = s.itok_dist; d.n_itok = s.n_itok;
  return d;
}

// the penalty-string distance table grows when queries intern new strings: the new table is uploaded beside the old one and
// swapped in; calls in flight keep the table they pinned
static void refresh_itok_table(DeviceIndex::Impl& m, const HostIndex& h) {
  if (m.itok_version.load() == h.itok_version.load()) return;
  std::lock_guard<std::mutex> one(m.refresh_mu);
  if (m.itok_version.load() == h.itok_version.load()) return;
  // a consistent (table, string count, version) triple: another thread may be interning a string right now
  std::vector<uint16_t> dist;
  uint32_t n = 0;
  uint64_t version = 0;
  h.snapshot_itok_dist(dist, n, version);
  std::shared_ptr<DevBlob> blob(new DevBlob);
  if (!dist.empty()) blob->a = dev_upload(dist);
  {
    std::lock_guard<std::mutex> g(m.tab_mu);
    m.itok_hold = blob;
    m.d.itok_dist = (const uint16_t*)blob->a;
    m.d.n_itok = n;
  }
  m.itok_version.store(version);
}

static DevVocab upload_vocab_table(const HostIndex::VocabTable& t, DevBlob& blob) {
  std::vector<ulonglong2> ent(t.n_slots);  // slot and head side by side: one 16-byte load per probe
  for (uint64_t k = 0; k < t.n_slots; k++) { ent[k].x = t.slot[k]; ent[k].y = t.head[k]; }
  ulonglong2* de = dev_alloc<ulonglong2>(t.n_slots);
  blob.a = de;
  unsigned char* da = dev_alloc<unsigned char>(t.arena_bytes + 16);
  blob.b = da;
  HIP_TRY(hipMemcpy(de, ent.data(), t.n_slots * sizeof(ulonglong2), hipMemcpyHostToDevice));
  HIP_TRY(hipMemset(da, 0, t.arena_bytes + 16));
  if (t.arena_bytes) HIP_TRY(hipMemcpy(da, t.arena, t.arena_bytes, hipMemcpyHostToDevice));
  DevVocab v{};
  v.entry = de; v.arena = da; v.mask = t.n_slots - 1;
  return v;
}

// the device copy of the host's spelling table (text batches): uploaded on first use, replaced when words were added
static void refresh_vocab_table(DeviceIndex::Impl& m, const HostIndex& h) {
  const HostIndex::VocabTable t = h.vocab_table();
  {
    std::lock_guard<std::mutex> g(m.tab_mu);
    if (m.vocab.entry && m.vocab_built_for == t.built_for) return;
  }
  std::lock_guard<std::mutex> one(m.refresh_mu);
  {
    std::lock_guard<std::mutex> g(m.tab_mu);
    if (m.vocab.entry && m.vocab_built_for == t.built_for) return;
  }
  if (t.n_slots == 0) throw std::invalid_argument("text batch on an index without spellings (fmx_index_set_forms)");
  std::shared_ptr<DevBlob> blob(new DevBlob);
  const DevVocab v = upload_vocab_table(t, *blob);
  std::lock_guard<std::mutex> g(m.tab_mu);
  m.vocab_hold = blob;
  m.vocab = v;
  m.vocab_built_for = t.built_for;
}

constexpr size_t kDynLdsMax = 150 * 1024;  // dynamic LDS the row kernels may ask for (160 KB per CU on gfx950)

// The accelerator arrays from the reference's arrays, on the device: kernels in fmx_derive.hip.hh.
// m.d.tok and m.d.sent_pos are already resident; the suffix array goes up (6 bytes per suffix) and is dropped afterwards.
// Returns the per-sentence 112-bucket signatures (two words per sentence, device memory): the caller builds the bit planes
// from them and frees them.
static unsigned long long* derive_on_device(DeviceIndex::Impl& m, const HostIndex& h, SortResident* res) {
  const uint64_t n = h.num_suffixes();
  const uint32_t n_sent = h.num_sentences();
  m.d.suf = nullptr; m.d.suf_b = nullptr; m.d.suf_sig = nullptr; m.d.suf_key = nullptr;
  m.d.gram.g3 = nullptr; m.d.gram.g5 = nullptr; m.d.gram.g7 = nullptr; m.d.gram.mask3 = 0; m.d.gram.n_b5 = 0; m.d.gram.mask7 = 0;
  SufRec* suf = dev_alloc<SufRec>(n);
  SufRecB* suf_b = dev_alloc<SufRecB>(n);
  unsigned long long* suf_sig = dev_alloc<unsigned long long>(n);
  SufKey* suf_key = dev_alloc<SufKey>(n);
  m.d.suf = suf; m.d.suf_b = suf_b; m.d.suf_sig = suf_sig; m.d.suf_key = suf_key;
  m.bytes += n * (sizeof(SufRec) + sizeof(SufRecB) + 8 + sizeof(SufKey));
  if (n == 0) return nullptr;
  unsigned long long* d_ssig = dev_alloc<unsigned long long>(2ull * n_sent);
  {
    // the suffix array: still in HBM when the sort ran on this device just before, else uploaded (6 bytes per suffix)
    uint32_t* d_sid = res ? res->d_sid : dev_upload(h.suf_sid);
    uint16_t* d_pos = res ? res->d_pos : dev_upload(h.suf_pos);
    if (res) { res->d_sid = nullptr; res->d_pos = nullptr; }
    hipLaunchKernelGGL(k_dv_sent_sig, dim3((n_sent + 255) / 256), dim3(256), 0, 0, m.d.tok, m.d.sent_pos, n_sent, d_ssig);
    hipLaunchKernelGGL(k_dv_suffix, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, m.d.tok, m.d.sent_pos, (const uint32_t*)d_sid,
                       (const uint16_t*)d_pos, (const unsigned long long*)d_ssig, (unsigned long long)n, suf, suf_b, suf_sig, suf_key);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    hipFree(d_sid); hipFree(d_pos);
  }
  // ---- n-gram tables ----
  const uint32_t n32 = (uint32_t)n;
  uint32_t* bnd3 = dev_alloc<uint32_t>(n);
  uint32_t* bnd5 = nullptr;
  uint32_t* bnd7 = nullptr;
  uint32_t* lo3 = nullptr;
  unsigned long long* d_count = dev_alloc<unsigned long long>(1);
  void* tmp = nullptr;
  auto cleanup = [&]() { hipFree(bnd3); hipFree(bnd5); hipFree(bnd7); hipFree(lo3); hipFree(d_count); hipFree(tmp); };
  try {
    const DvTriBoundary tri{suf, suf_key, m.d.tok};
    // stream compaction of the indices i in [0, n) with pred(i), in order (round 6: hand-written — ballots per 64 indices, the
    // pass's look-back scan over their counts, a scatter — in place of rocprim::select, the last library call of the derive)
    const uint32_t n_ch = (uint32_t)((n + kWave - 1) / kWave);
    hipFree(tmp); tmp = nullptr;
    HIP_TRY(hipMalloc(&tmp, ((size_t)3 * (n_ch + 1) + kLbMaxTiles + 2) * sizeof(unsigned long long)));
    unsigned long long* sel_flags = (unsigned long long*)tmp;
    unsigned long long* sel_cnt = sel_flags + (n_ch + 1);
    unsigned long long* sel_offs = sel_cnt + (n_ch + 1);
    unsigned long long* sel_state = sel_offs + (n_ch + 1);
    auto select = [&](auto pred, uint32_t* out) -> uint32_t {
      HIP_TRY(hipMemsetAsync(sel_state, 0, ((size_t)kLbMaxTiles + 2) * sizeof(unsigned long long), 0));
      hipLaunchKernelGGL(k_dv_mark<decltype(pred)>, dim3(std::min<uint32_t>(n_ch + 1, 65536u)), dim3(kWave), 0, 0, pred, (unsigned long long)n, n_ch, sel_flags, sel_cnt);
      const uint32_t n_sc = n_ch + 1;
      if ((n_sc + kScanTile - 1) / kScanTile > kLbMaxTiles) throw UnsupportedError("index too large for the derive's scan");
      hipLaunchKernelGGL(k_scan_lb<1>, dim3((n_sc + kScanTile - 1) / kScanTile), dim3(kScanThreads), 0, 0, (const unsigned long long*)sel_cnt, sel_offs, (size_t)n_sc, n_sc,
                         sel_state, reinterpret_cast<uint32_t*>(sel_state + kLbMaxTiles));
      hipLaunchKernelGGL(k_dv_scatter, dim3(std::min<uint32_t>(n_ch, 65536u)), dim3(kWave), 0, 0, (unsigned long long)n, (const unsigned long long*)sel_flags,
                         (const unsigned long long*)sel_offs, out);
      HIP_TRY(hipGetLastError());
      unsigned long long num = 0;
      HIP_TRY(hipMemcpy(&num, sel_offs + n_ch, sizeof(num), hipMemcpyDeviceToHost));
      return (uint32_t)num;
    };
    auto read_count = [&]() { unsigned long long v = 0; HIP_TRY(hipMemcpy(&v, d_count, 8, hipMemcpyDeviceToHost)); return v; };
    auto table_cap = [](unsigned long long groups) { unsigned long long cap = 16; while (cap < groups * 2) cap <<= 1; return cap; };
    const uint32_t n_b3 = select(tri, bnd3);
    HIP_TRY(hipMemset(d_count, 0, 8));
    hipLaunchKernelGGL(k_dv_count_starts, dim3(std::min<uint32_t>((n_b3 + 255) / 256, 2048u)), dim3(256), 0, 0, (const uint32_t*)bnd3, n_b3, (const SufRec*)suf, 3u, d_count);
    const unsigned long long groups = read_count();
    const unsigned long long cap3 = table_cap(groups);
    if (groups != 0 && cap3 <= (1ull << 31)) {  // else: the keyed search is used instead (same rule as HostIndex::build_gram_table)
      GramEntry* g3 = dev_alloc<GramEntry>(cap3);
      m.d.gram.g3 = g3; m.d.gram.mask3 = (uint32_t)(cap3 - 1);
      m.bytes += cap3 * sizeof(GramEntry);
      hipLaunchKernelGGL(k_dv_fill_empty, dim3((unsigned)((cap3 + 255) / 256)), dim3(256), 0, 0, g3, cap3);
      hipLaunchKernelGGL(k_dv_insert3, dim3((n_b3 + 255) / 256), dim3(256), 0, 0, (const uint32_t*)bnd3, n_b3, n32, (const SufRec*)suf,
                         (const SufKey*)suf_key, m.d.tok, g3, (uint32_t)(cap3 - 1));
      // the 5-gram buckets: every distinct 5-gram (fmx_core.hh: G5Bucket)
      bnd5 = dev_alloc<uint32_t>(n);
      const DvFiveBoundary five{tri};
      const uint32_t n_b5 = select(five, bnd5);
      HIP_TRY(hipMemset(d_count, 0, 8));
      hipLaunchKernelGGL(k_dv_count_starts, dim3(std::min<uint32_t>((n_b5 + 255) / 256, 2048u)), dim3(256), 0, 0, (const uint32_t*)bnd5, n_b5, (const SufRec*)suf, 5u, d_count);
      const unsigned long long groups5 = read_count();
      const unsigned long long nb5 = std::max<unsigned long long>(16, (groups5 * 16 + kG5FillPer16 - 1) / kG5FillPer16);
      if (groups5 != 0 && nb5 <= 0xffffffffull) {
        G5Bucket* g5 = dev_alloc<G5Bucket>(nb5);
        m.d.gram.g5 = g5; m.d.gram.n_b5 = (uint32_t)nb5;
        m.bytes += nb5 * sizeof(G5Bucket);
        HIP_TRY(hipMemset(g5, 0, nb5 * sizeof(G5Bucket)));
        hipLaunchKernelGGL(k_dv_insert5x, dim3((n_b5 + 255) / 256), dim3(256), 0, 0, (const uint32_t*)bnd5, n_b5, n32, (const SufRec*)suf,
                           (const SufKey*)suf_key, m.d.tok, g5, (uint32_t)nb5);
        // second level: the distinct 7-grams of the 5-gram groups with more than kDeepGramMin suffixes
        bnd7 = dev_alloc<uint32_t>(n);
        const uint32_t n_b7 = select(DvSevenBoundary{five}, bnd7);
        lo3 = dev_alloc<uint32_t>(n_b7);
        HIP_TRY(hipMemset(d_count, 0, 8));
        hipLaunchKernelGGL(k_dv_mark_deep, dim3((n_b7 + 255) / 256), dim3(256), 0, 0, (const uint32_t*)bnd7, n_b7, (const uint32_t*)bnd5, n_b5, n32,
                           (const SufRec*)suf, 5u, 7u, lo3, d_count);
        const unsigned long long groups7 = read_count();
        const unsigned long long cap7 = table_cap(groups7);
        if (groups7 != 0 && cap7 <= (1ull << 31)) {
          GramEntry* g7 = dev_alloc<GramEntry>(cap7);
          m.d.gram.g7 = g7; m.d.gram.mask7 = (uint32_t)(cap7 - 1);
          m.bytes += cap7 * sizeof(GramEntry);
          hipLaunchKernelGGL(k_dv_fill_empty, dim3((unsigned)((cap7 + 255) / 256)), dim3(256), 0, 0, g7, cap7);
          hipLaunchKernelGGL(k_dv_insert_deep, dim3((n_b7 + 255) / 256), dim3(256), 0, 0, (const uint32_t*)bnd7, n_b7, n32, (const uint32_t*)lo3,
                             (const SufKey*)suf_key, 4u, g7, (uint32_t)(cap7 - 1));
        }
      }
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
  } catch (...) {
    cleanup();
    hipFree(d_ssig);
    throw;
  }
  cleanup();
  return d_ssig;
}

// test / debug: a derived array as it sits in HBM
uint64_t DeviceIndex::download(int which, void* dst, uint64_t capacity_bytes) const {
  const Impl& m = *impl_;
  if (m.device < 0) throw GpuError("index is not on a device");
  HIP_TRY(hipSetDevice(m.device));
  const void* src = nullptr;
  uint64_t count = 0, elem = 0;
  switch (which) {
    case 0: src = m.d.suf; count = m.d.n_suf; elem = sizeof(SufRec); break;
    case 1: src = m.d.suf_sig; count = m.d.n_suf; elem = 8; break;
    case 2: src = m.d.suf_key; count = m.d.n_suf; elem = sizeof(SufKey); break;
    case 3: src = m.d.gram.g3; count = m.d.gram.g3 ? (uint64_t)m.d.gram.mask3 + 1 : 0; elem = sizeof(GramEntry); break;
    case 4: src = m.d.gram.g5; count = m.d.gram.g5 ? (uint64_t)m.d.gram.n_b5 : 0; elem = sizeof(G5Bucket); break;
    case 8: src = m.d.gram.g7; count = m.d.gram.g7 ? (uint64_t)m.d.gram.mask7 + 1 : 0; elem = sizeof(GramEntry); break;
    case 5: src = m.d.suf_b; count = m.d.n_suf; elem = sizeof(SufRecB); break;
    case 6: src = m.d.sent_pos; count = m.d.n_sent; elem = 4; break;   // positions in the device layout of the sentence buffer
    case 7: src = m.d.tok; count = m.n_tok_words; elem = 4; break;
    default: throw std::invalid_argument("unknown device array");
  }
  if (dst && count) {
    if (count * elem > capacity_bytes) throw std::invalid_argument("destination too small");
    HIP_TRY(hipMemcpy(dst, src, count * elem, hipMemcpyDeviceToHost));
  }
  return count;
}

void DeviceIndex::upload(const HostIndex& h, int device, SortResident* res) {
  Impl& m = *impl_;
  if (m.device >= 0) throw GpuError("index is already on a device");
  gpu_check_device(device);
  HIP_TRY(hipSetDevice(device));
  m.device = device;
  const uint64_t n_suf = h.num_suffixes();
  if (n_suf >= 0xffffffffull) throw UnsupportedError("more than 2^32 - 1 suffixes");
  if (res && (res->device != device || res->n_buf != h.buf.size() || res->n_sent != h.sent_pos.size() || res->n_suf != n_suf || !res->d_buf ||
              !res->d_sent_pos || !res->d_sid || !res->d_pos))
    res = nullptr;  // not what this index looks like now: ignore it
  uint32_t* d_tok = nullptr;
  if (res) {  // left in HBM by the suffix sort (with the eight spare words)
    m.d.tok = res->d_buf; res->d_buf = nullptr;
    m.d.sent_pos = res->d_sent_pos; res->d_sent_pos = nullptr;
  } else {  // eight spare tokens: the DP kernels read the sentence four tokens at a time, possibly past its end
    d_tok = dev_alloc<uint32_t>(h.buf.size() + kTokSpare);
    HIP_TRY(hipMemset(d_tok, 0, (h.buf.size() + kTokSpare) * sizeof(uint32_t)));
    if (!h.buf.empty()) HIP_TRY(hipMemcpy(d_tok, h.buf.data(), h.buf.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    m.d.tok = d_tok;
    m.d.sent_pos = dev_upload(h.sent_pos);
  }
  m.n_tok_words = h.buf.size();
  // Device layout of the sentence buffer: a sentence [len, tokens, 0] that fits one 128-byte line does not straddle two (the
  // expand and edit-distance kernels read a candidate's sentence at random: one line instead of ~1.8).  Every derived array
  // holds offsets into THIS layout (they are all built on the device from m.d.tok / m.d.sent_pos); the host arrays and the
  // .fmi keep the reference's packed _sentence_buffer.  d_old_pos = the packed positions (for the per-slot twins below).
  uint32_t* d_old_pos = nullptr;
  static const bool pad_tok = [] { const char* e = getenv("FMX_PAD_TOK"); return e ? atoi(e) != 0 : true; }();
  if (pad_tok && !h.sent_pos.empty()) {
    std::vector<uint32_t> pos(h.sent_pos.size());
    uint64_t at = 0;
    for (size_t s2 = 0; s2 < h.sent_pos.size(); s2++) {
      const uint64_t size = (uint64_t)h.buf[h.sent_pos[s2]] + 2;
      if ((at & 31) != 0 && (at & 31) + size > 32) at = (at + 31) & ~31ull;
      pos[s2] = (uint32_t)at;
      at += size;
      if (at >= 0xfffffff0ull) break;
    }
    if (at < 0xfffffff0ull) {
      uint32_t* d_new_pos = dev_upload(pos);
      uint32_t* d_new = dev_alloc<uint32_t>(at + kTokSpare);
      HIP_TRY(hipMemset(d_new, 0, (at + kTokSpare) * sizeof(uint32_t)));
      const uint32_t ns = (uint32_t)h.sent_pos.size();
      hipLaunchKernelGGL(k_relayout, dim3((ns + 255) / 256), dim3(256), 0, 0, m.d.tok, m.d.sent_pos, m.d.tok, m.d.sent_pos, (const uint32_t*)d_new_pos, ns, d_new);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipDeviceSynchronize());
      hipFree((void*)m.d.tok);
      d_old_pos = const_cast<uint32_t*>(m.d.sent_pos);
      m.d.tok = d_new; m.d.sent_pos = d_new_pos;
      m.n_tok_words = at;
    }
  }
  m.d.qva = dev_upload(h.qva);
  m.d.idf = dev_upload(h.idf);
  m.d.n_qva = (uint32_t)h.qva.size();
  m.d.n_sent = h.num_sentences();
  m.d.n_suf = h.num_suffixes();
  uint32_t mx = 0;
  for (uint32_t sp : h.sent_pos) mx = std::max(mx, h.buf[sp]);
  m.d.max_slen = mx;
  m.bytes = m.n_tok_words * 4 + h.sent_pos.size() * 4 + h.qva.size() * 4 + h.idf.size() * 4;
  unsigned long long* d_ssig = derive_on_device(m, h, res);
  // bit planes of the signatures and thermometer planes of the stored lengths, built on the device from suf_sig
  // (DESIGN.md §3): plane-major, one 64-bit word per (row, block of 64 consecutive suffixes)
  m.d.n_blk = (m.d.n_suf + 63) / 64;
  m.d.planes = nullptr;
  m.d.thermo = nullptr;
  m.d.n_thermo = std::min<uint32_t>(mx, kSigLenClamp);
  if (m.d.n_suf) {
    unsigned long long* pl = dev_alloc<unsigned long long>((size_t)kPlaneRows * m.d.n_blk);
    unsigned long long* th = dev_alloc<unsigned long long>((size_t)std::max<uint32_t>(m.d.n_thermo, 1) * m.d.n_blk);
    hipLaunchKernelGGL(k_build_planes, dim3((unsigned)m.d.n_blk), dim3(kWave), 0, 0, m.d.suf_sig, m.d.suf_b, (const unsigned long long*)d_ssig, m.d.n_suf,
                       m.d.n_blk, pl, th, m.d.n_thermo);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    m.d.planes = pl;
    m.d.thermo = th;
    m.bytes += (uint64_t)(kPlaneRows + m.d.n_thermo) * m.d.n_blk * 8;
  }
  hipFree(d_ssig);
  if (h.real_differs) {
    // (eight spare words like the token buffer: the register kernels read a sentence's real-token ids four at a time)
    for (auto pr : {std::make_pair(&m.d.real, &h.real_ids), std::make_pair(&m.d.itok, &h.itok_ids)}) {
      uint32_t* d = dev_alloc<uint32_t>(pr.second->size() + kTokSpare);
      HIP_TRY(hipMemset(d, 0, (pr.second->size() + kTokSpare) * sizeof(uint32_t)));
      if (!pr.second->empty()) HIP_TRY(hipMemcpy(d, pr.second->data(), pr.second->size() * sizeof(uint32_t), hipMemcpyHostToDevice));
      *pr.first = d;
    }
    if (d_old_pos) {  // the per-slot twins of the token buffer follow it into the device layout
      const uint32_t ns = (uint32_t)h.sent_pos.size();
      for (const uint32_t** twin : {&m.d.real, &m.d.itok}) {
        uint32_t* d_new = dev_alloc<uint32_t>(m.n_tok_words + kTokSpare);
        HIP_TRY(hipMemset(d_new, 0, (m.n_tok_words + kTokSpare) * sizeof(uint32_t)));
        hipLaunchKernelGGL(k_relayout, dim3((ns + 255) / 256), dim3(256), 0, 0, m.d.tok, m.d.sent_pos, *twin, (const uint32_t*)d_old_pos, m.d.sent_pos, ns, d_new);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        hipFree((void*)*twin);
        *twin = d_new;
      }
    }
    m.bytes += 2 * m.n_tok_words * 4;
    m.real_vocab = upload_vocab_table(h.real_table(), m.real_hold);
  }
  if (d_old_pos) hipFree(d_old_pos);
  refresh_itok_table(m, h);
  // the LDS-row kernels may ask for up to kDynLdsMax of dynamic LDS: raise the limit once (not per call: the
  // attribute is process-wide and calls run concurrently)
  HIP_TRY(hipFuncSetAttribute((const void*)k_dp<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynLdsMax));
  HIP_TRY(hipFuncSetAttribute((const void*)k_dp<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynLdsMax));
  HIP_TRY(hipFuncSetAttribute((const void*)k_contrast<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynLdsMax));
  HIP_TRY(hipDeviceSynchronize());
}

static void ensure_ws(Ctx& c, uint64_t nq, uint64_t nt, uint64_t seg) {
  c.q_m.ensure(nq); c.q_kept.ensure(nq); c.q_cand.ensure(nq); c.q_acc.ensure(nq); c.q_pb.ensure(nq); c.q_best.ensure(3 * nq);
  c.q_rank.ensure(nq); c.q_scap.ensure(nq); c.big_list.ensure(nq); c.mid_list.ensure(nq); c.fin_list.ensure(nq); c.contrast_list.ensure(2 * nq); c.fb_list.ensure(4 * nq);
  c.q_hits.ensure(nq + 1); c.q_hits_a.ensure(nq + 1); c.q_seg.ensure(3 * (nq + 1)); c.q_qseg.ensure(2 * (nq + 1)); c.q_cap.ensure(3 * (nq + 1));
  c.c_in.ensure(3 * (nq + 1)); c.c_off.ensure(3 * (nq + 1));
  c.t_q.ensure(nt); c.t_lo.ensure(nt); c.t_cnt.ensure(nt); c.t_idf.ensure(nt); c.items_t.ensure((uint64_t)kTailLists * probe_tail_list_cap((uint32_t)std::min<uint64_t>(nt, 0xffffffffull)));
  if (seg > c.seg_cap) {
    // (the candidate lists name segment slots with 32 bits; a pass is at most a few million queries)
    if (seg > 0xffffffffull) throw UnsupportedError("a pass needs more than 2^32 - 1 segment entries: lower \"sub_queries\"");
    const uint64_t s = std::min<uint64_t>(std::max<uint64_t>(seg, c.seg_cap + c.seg_cap / 2), 0xffffffffull);
    c.c_qi.ensure(s); c.seg.ensure(s); c.f0.ensure(s); c.f1.ensure(s);
    c.keys.ensure(2 * s); c.vals.ensure(2 * s);
    c.seg_cap = s;
  }
  c.heap.ensure(2 * c.seg_cap + 2);
  c.qb.ensure(c.want_qb);
  if (!c.items.p) { c.items.ensure(1u << 22); c.items_b.ensure(1u << 22); c.items_h.ensure(1u << 20); }
}

static DevWork work_view(Ctx& c) {
  DevWork w{};
  w.q_m = c.q_m.p; w.t_q = c.t_q.p; w.t_lo = c.t_lo.p; w.t_cnt = c.t_cnt.p; w.q_hits = c.q_hits.p; w.q_hits_a = c.q_hits_a.p;
  w.q_cap = c.q_cap.p; w.q_seg = c.q_seg.p; w.q_qseg = c.q_qseg.p; w.qb_cap = c.qb.cap; w.q_kept = c.q_kept.p; w.q_cand = c.q_cand.p; w.q_acc = c.q_acc.p; w.q_pb = c.q_pb.p; w.q_scap = c.q_scap.p;
  w.q_best = c.q_best.p; w.q_rank = c.q_rank.p; w.p_hist = c.p_hist_p(); w.cmin_tab = c.cmin_tab.p; w.c_in = c.c_in.p;
  w.c_off = c.c_off.p; w.c_qi = c.c_qi.p; w.t_idf = c.t_idf.p; w.seg = c.seg.p; w.qb = c.qb.p; w.heap = c.heap.p;
  w.seg_cap = c.seg_cap; w.counters = c.counters_p();
  return w;
}

const char* stage_name(int s) {
  // 0..10 follow one another on the main stream (their sum is the pass): expand_a = k_expand_a (phase A of the queries that are
  // not light), expand_l = k_expand_b<true> (the light queries: ranges walked and reduced by one kernel), expand_wait = what the
  // side stream still has to do then.  On the side stream, beside them: expand_h = k_expand_h, expand_b = second scan +
  // k_expand_b<false> (the queries that are not light).
  static const char* names[kNumStages] = {"setup", "probe", "scan", "expand_a", "expand_l", "expand_wait", "cand", "cand_big", "cand_scan", "dp",
                                          "finish", "expand_h", "expand_b"};
  return (s >= 0 && s < kNumStages) ? names[s] : "";
}

// hooks of the host-buffer path around each pass (sub-batch) of the pipeline
struct PassHooks {
  std::function<void(uint32_t)> before;        // inputs of pass sb must be on their way; make `stream` wait for them
  std::function<void(uint32_t, hipEvent_t)> after;  // the records of pass sb are final once the event (or, if null, the stream) is done
  std::function<void()> computed;              // the kernels of every pass of a group are through (their records may still be travelling)
  std::function<void(hipEvent_t)> enqueued;    // every pass of the first group is enqueued; the event = its last kernel
};

struct PassEnv {
  DeviceIndex::Impl& m;
  Ctx& c;
  TableSnap snap;  // the replaceable tables this call pinned
  DevIndex d;      // the index as this call's kernels see it
  Params prm;
  bool full, use_idf;
  Rec* d_out; uint32_t* d_out_count; uint32_t* d_out_total;
  uint32_t cap;
  hipStream_t stream;
  uint32_t prior_max_len;  // longest prior match of the batch (rows of the general contrast kernel must hold it)
};

// grid of the flattened rerank rounds (grid-stride over the flat list): design-time switch FMX_CROUND_GRID = workgroups per CU
static unsigned cround_grid(int p) {
  static const int per_cu = [] { const char* e = getenv("FMX_CROUND_GRID"); return e ? atoi(e) : 0; }();
  if (per_cu > 0) return 256u * (unsigned)per_cu;
  return p == 32 ? 256u * 64u : 256u * 32u;  // (16 / 8 per CU until round 3: at 64 the stage is 0.3 ms shorter — six waves per SIMD instead of four)
}
// every launch of one pass, in stream order; the pass counters land in h_ctr_slot (pinned).  No host synchronisation.
static void enqueue_pass(PassEnv& E, const DevBatch& b, uint32_t p_max, bool timing, unsigned long long* h_ctr_slot) {
  DeviceIndex::Impl& m = E.m;
  Ctx& c = E.c;
  const Params& prm = E.prm;
  const bool full = E.full, use_idf = E.use_idf;
  // length classes of the candidate lists (dp_class): real tokens on both sides -> the register kernels with the real-token term
  // (round 6: what an index built under the reference's default -p nbr,tag,cas runs), pairs with penalty tokens deferred to the general one
  static const bool real_reg = [] { const char* e = getenv("FMX_REAL_REG"); return e ? atoi(e) != 0 : true; }();  // design-time switch
  const int dp_mode = !full ? 0 : ((real_reg && E.d.real && b.q_real) ? (use_idf ? 3 : 2) : 1);
  Rec* d_out = E.d_out; uint32_t* d_out_count = E.d_out_count; uint32_t* d_out_total = E.d_out_total;
  const uint32_t cap = E.cap;
  hipStream_t stream = E.stream;
  // (round 6) the sorter of the few huge survivor sets starts on the side lane as soon as k_expand_b<false> is through, beside the
  // light queries' hit walk, instead of after it (only with light queries: without them k_expand_b<false> is the whole stage)
  static const bool early_big_on = [] { const char* e = getenv("FMX_EARLY_BIG"); return e ? atoi(e) != 0 : true; }();  // design-time switch
  bool early_big = false, big_pending = false;
  DevWork w_big{};   // the pass's work view (filled below, before anything is launched)
  auto launch_big = [&](hipStream_t st) {
    const DevWork& w2 = w_big;
    // radix passes needed: sentence ids < n_sent, match lengths <= longest pattern
    uint32_t sid_bytes = 1, len_bytes = p_max < 255 ? 1u : 2u;
    while (sid_bytes < 4 && (E.d.n_sent >> (8 * sid_bytes)) != 0) sid_bytes++;
    hipLaunchKernelGGL(k_cand_route_big, dim3((b.nq + 255) / 256), dim3(256), 0, st, b, w2, c.big_list.p);
    hipLaunchKernelGGL(k_cand_big, dim3(kBigGrid), dim3(kBigThreads), 0, st, w2, c.keys.p, (const uint32_t*)c.big_list.p, sid_bytes, len_bytes);
  };
    DevWork w = work_view(c);
    const uint32_t item_cap = (uint32_t)std::min<uint64_t>(c.items.cap, 0xffffffffull);
    const uint32_t item_h_cap = (uint32_t)std::min<uint64_t>(c.items_h.cap, 0xffffffffull);
    // one memset per pass: counters, scan states / tickets, pattern-length histogram (the per-query accumulators are
    // zeroed by k_setup)
    HIP_TRY(hipMemsetAsync(c.zero_blk.p, 0, Ctx::zero_words() * sizeof(unsigned long long), stream));
    const uint32_t n_scan = b.nq + 1;  // one extra element so that out[nq] = total
    w.q_capl = w.q_cap + n_scan;   // second array of the first scan (input / output)
    w.q_segl = w.q_qseg + n_scan;
    w_big = w;
    const uint32_t scan_tiles = (n_scan + kScanTile - 1) / kScanTile;
    if (scan_tiles > kLbMaxTiles) throw std::invalid_argument("sub-batch too large for the scan");
    int e = 0;
    auto tick = [&]() { if (timing) HIP_TRY(hipEventRecord(c.ev[e], stream)); e++; };
    // the signature bound needs theoretical_rejection_cover to be monotone in the cover: costs >= 0
    const int use_sig = prm.ins >= 0.f && prm.del >= 0.f && prm.rep >= 0.f;
    // suffix ranges of at least heavy_min hits are walked through the signature bit planes (k_expand_h)
    const uint32_t heavy_min = (use_sig && E.d.planes && m.heavy_min) ? std::max<uint32_t>(m.heavy_min, 256u) : 0xffffffffu;
    // fork: `side` continues after everything enqueued on `stream` so far; join: `stream` waits for `side`
    const hipStream_t side = m.serial ? stream : c.side;
    auto fork = [&]() { if (!m.serial) { HIP_TRY(hipEventRecord(c.ev_fork, stream)); HIP_TRY(hipStreamWaitEvent(c.side, c.ev_fork, 0)); } };
    auto join = [&]() { if (!m.serial) { HIP_TRY(hipEventRecord(c.ev_join, c.side)); HIP_TRY(hipStreamWaitEvent(stream, c.ev_join, 0)); } };
    tick();
    // the phase-A threshold table depends on the parameters only: computed on the side stream while the probe runs
    // (after the fork: the previous pass, which read the table, is complete)
    const float cmin_key[4] = {prm.fuzzy, prm.ins, prm.del, prm.rep};
    const bool cmin_cached = c.cmin_ptr == c.cmin_tab.p && c.cmin_sig == use_sig && c.cmin_maxslen == E.d.max_slen && c.cmin_pmax >= p_max &&
                             memcmp(c.cmin_key, cmin_key, sizeof cmin_key) == 0;
    // beside the probe, on the side stream: the threshold table (unless the pass before asked for the same) and the pattern-length
    // ranks (only k_classify, after the expand stage's join, reads them)
    fork();
    if (!cmin_cached) {
      hipLaunchKernelGGL(k_cmin, dim3(p_max + 1), dim3(256), 0, side, E.d.max_slen, prm, use_sig, c.cmin_tab.p);
      if (!m.serial) HIP_TRY(hipEventRecord(c.ev_cmin, c.side));
      memcpy(c.cmin_key, cmin_key, sizeof cmin_key);
      c.cmin_ptr = c.cmin_tab.p; c.cmin_sig = use_sig; c.cmin_maxslen = E.d.max_slen; c.cmin_pmax = p_max;
    }
    {
      const uint32_t per_block = kRankThreads * kRankPer;
      hipLaunchKernelGGL(k_phist, dim3((b.nq + per_block - 1) / per_block), dim3(kRankThreads), 0, side, b, w);
      hipLaunchKernelGGL(k_prank, dim3((b.nq + per_block - 1) / per_block), dim3(kRankThreads), 0, side, b, w);
    }
    hipLaunchKernelGGL(k_setup, dim3((b.nq + 1 + 255) / 256), dim3(256), 0, stream, E.d, b, w, prm);
    if (prm.idf_penalty != 0.f) hipLaunchKernelGGL(k_tidf, dim3((b.nt + 255) / 256), dim3(256), 0, stream, E.d, b, w);
    tick(); tick();
    if (b.nt) {
      const dim3 pg((b.nt + kProbeBlock - 1) / kProbeBlock), pb(kProbeBlock);
      // (the tail's persistent grid shrinks with the pass: a fifth of the tokens at most is queued, kTailLists lists)
      const uint32_t tail_per = std::min<uint32_t>(kProbeTailPer, std::max<uint32_t>(1u, (uint32_t)(b.nt / (5ull * kWave * kTailLists)) + 1u));
      const dim3 tg(kTailLists * tail_per);
      if (m.probe_split) {
        // head: a lane per start position, one table line; tail: a lane per start position the head could not decide
        if (m.probe_count) {  // measurement runs only: the same kernels, also counting the bytes / lines they request
          hipLaunchKernelGGL((k_probe<true, true>), pg, pb, 0, stream, E.d, b, w, heavy_min, c.items_h.p, item_h_cap, c.items_t.p);
          hipLaunchKernelGGL(k_probe_tail<true>, tg, dim3(kWave), 0, stream, E.d, b, w, heavy_min, c.items_h.p, item_h_cap, (const ProbeItem*)c.items_t.p);
        } else {
          hipLaunchKernelGGL((k_probe<false, true>), pg, pb, 0, stream, E.d, b, w, heavy_min, c.items_h.p, item_h_cap, c.items_t.p);
          hipLaunchKernelGGL(k_probe_tail<false>, tg, dim3(kWave), 0, stream, E.d, b, w, heavy_min, c.items_h.p, item_h_cap, (const ProbeItem*)c.items_t.p);
        }
      } else if (m.probe_count) {
        hipLaunchKernelGGL((k_probe<true, false>), pg, pb, 0, stream, E.d, b, w, heavy_min, c.items_h.p, item_h_cap, c.items_t.p);
      } else {
        hipLaunchKernelGGL((k_probe<false, false>), pg, pb, 0, stream, E.d, b, w, heavy_min, c.items_h.p, item_h_cap, c.items_t.p);
      }
    }
    tick(); tick();
    // light queries (k_caps_scan): at most light_max hits, all in short ranges, pattern of at most one wave
    const uint32_t light_max = b.nq ? std::min<uint32_t>(m.light_max, kPhaseBChunk) : 0u;
    early_big = early_big_on && light_max != 0;
    // two queries per wave in the last kernel (k_finish_pair): no contrastive rerank, lowest_costs fits half a wave
    // (with or without a contrastive factor; any contrast_buffer: beyond 30 entries the pair kernel takes 31 candidates per query)
    const bool pair_finish = m.pair_finish && prm.cbuf > 0;
    const uint32_t pair_max = prm.cbuf <= (int)kPairMax - 2 ? kPairMax : kPairMax - 1;
    {
      // capacities, one scan over two arrays (queue offsets of the queries phase A walks, segment offsets of the light queries)
      // and the work-item list of phase A in one launch
      hipLaunchKernelGGL(k_caps_scan, dim3(scan_tiles), dim3(kScanThreads), 0, stream, b, w, c.items.p, item_cap, c.s_max, c.heavy_div, light_max,
                         c.scan_state(0), c.scan_ticket(0));
    }
    tick(); tick();
    {
      // Two lanes from here to the candidate stage:
      //   main   k_expand_a (phase A of the queries that are not light, next to k_expand_h), then k_expand_b<true>: the light
      //          queries — ranges walked, filtered and reduced to candidates by one kernel; their segments are known already
      //   side   k_expand_h (the long ranges, through the bit planes), then — once k_expand_a is done too — the second scan
      //          (segments of the other queries, sized by the hits phase A really queued) and k_expand_b<false> for them
      // The light kernel is a per-query instruction stream, k_expand_h an HBM stream: they share the GPU well.
      const size_t lds = (size_t)p_max * 4;
      if (lds > 64 * 1024) throw UnsupportedError("patterns too long for the expand kernel's LDS staging");
      if (!m.serial && !cmin_cached) HIP_TRY(hipStreamWaitEvent(stream, c.ev_cmin, 0));
      // persistent grids shrink with the pass (a micro-batch of the coalescer pays ~25 us for 8192 workgroups that find nothing to do)
      const uint32_t grid_h = std::min<uint32_t>(m.heavy_grid, std::max<uint32_t>(256u, 4u * b.nq));
      const uint32_t grid_l = std::min<uint32_t>(kExpandExtraGrid, std::max<uint32_t>(256u, b.nq));
      const uint32_t grid_b = std::min<uint32_t>(m.b_grid, std::max<uint32_t>(256u, b.nq));
      auto side_tick = [&](int stage, int end) { if (timing) HIP_TRY(hipEventRecord(c.ev[2 * stage + end], side)); };
      // the light queries: those of at most flat_max hits several to a wave (k_expand_lf, fmx_light.hip.hh), the rest a wave each
      const uint32_t flat_max = std::min<uint32_t>(m.light_flat, light_max);
      auto launch_light = [&]() {
        if (flat_max) {
          const uint32_t G = b.nq >= (1u << 19) ? 32u : (b.nq >= (1u << 17) ? 16u : 8u);  // queries per wave: small passes still fill the GPU
          const uint32_t grid = (b.nq + G - 1) / G;
          if (flat_max <= 128) hipLaunchKernelGGL((k_expand_lf<128, 128>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm, G, flat_max);
          else hipLaunchKernelGGL((k_expand_lf<128, 256>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm, G, flat_max);
        }
        if (flat_max < light_max) {
          // (round 6) in parts when the big sorter waits on the side lane: workgroups of the oldest kernel are placed first, and one
          // launch of a million workgroups lets nothing younger in until it is nearly through; between two parts the side lane gets its turn
          static const uint32_t parts_env = [] { const char* e = getenv("FMX_LIGHT_PARTS"); return e ? (uint32_t)atoi(e) : 2u; }();  // design-time switch (1 / 2 / 3 / 6 parts at C3: stage 1.65 / 1.43 / 1.49 / 1.49 ms)
          const uint32_t parts = (early_big && b.nq >= 65536u) ? std::max(1u, parts_env) : 1u;
          for (uint32_t pt_ = 0; pt_ < parts; pt_++) {
            const uint32_t q0 = (uint32_t)((uint64_t)b.nq * pt_ / parts), q1 = (uint32_t)((uint64_t)b.nq * (pt_ + 1) / parts);
            if (q1 > q0)
              hipLaunchKernelGGL(k_expand_b<true>, dim3(q1 - q0), dim3(kWave), lds, stream, E.d, b, w, prm, (const WorkItem*)nullptr, q0, b.nq, flat_max);
          }
        }
      };
      auto second_scan = [&](hipStream_t st) {
        // the hits phase A really queued size the segments of everything downstream: second scan, q_cap -> q_seg, placed
        // behind the light queries' segments; the work-item list of k_expand_b<false> is laid out by the same
        // scan — second array, two strides on — instead of by atomics
        hipLaunchKernelGGL(k_seg_scan, dim3(scan_tiles), dim3(kScanThreads), 0, st, b, w, c.items_b.p, item_cap, light_max ? 1u : 0u, light_max ? 0u : 1u,
                           c.scan_state(1), c.scan_ticket(1));
      };
      fork();
      side_tick(kStageExpandH, 0);
      if (heavy_min != 0xffffffffu)
        hipLaunchKernelGGL(k_expand_h, dim3(grid_h), dim3(kWave), 0, side, E.d, b, w, prm, (const HeavyItem*)c.items_h.p, item_h_cap);
      side_tick(kStageExpandH, 1);
      // phase A of the queries that are not light: block q = query q (light_max == 0) or a work-item list
      hipLaunchKernelGGL(k_expand_a, dim3((light_max ? 0u : b.nq) + grid_l), dim3(kWave), 0, stream, E.d, b, w, prm,
                         (const WorkItem*)c.items.p, item_cap, use_sig, light_max ? 0u : b.nq);
      if (m.expand_order == 0) {
        if (!m.serial) { HIP_TRY(hipEventRecord(c.ev_a, stream)); HIP_TRY(hipStreamWaitEvent(c.side, c.ev_a, 0)); }
        tick(); tick();
        side_tick(kStageExpandB, 0);
        second_scan(side);
        hipLaunchKernelGGL(k_expand_b<false>, dim3((light_max ? 0u : b.nq) + grid_b), dim3(kWave), lds, side, E.d, b, w, prm,
                           (const WorkItem*)c.items_b.p, item_cap, light_max ? 0u : b.nq, 0u);
        side_tick(kStageExpandB, 1);
        if (early_big) launch_big(side);
        if (light_max) launch_light();
        tick(); tick();
      } else {
        // long ranges first (k_expand_h next to k_expand_a), then the two k_expand_b side by side: the one for the queries
        // that are not light is a chain of memory latencies on a small grid and must reach the GPU FIRST — workgroups of
        // the oldest kernel are placed first, and the light kernel's million workgroups would starve anything younger
        join();
        HIP_TRY(hipEventRecord(c.ev_mid, stream));  // probe + long ranges done: what follows is bound by instruction issue and latency
        tick(); tick();
        second_scan(stream);
        fork();
        side_tick(kStageExpandB, 0);
        hipLaunchKernelGGL(k_expand_b<false>, dim3((light_max ? 0u : b.nq) + grid_b), dim3(kWave), lds, side, E.d, b, w, prm,
                           (const WorkItem*)c.items_b.p, item_cap, light_max ? 0u : b.nq, 0u);
        side_tick(kStageExpandB, 1);
        if (early_big) {
          // the main lane goes on when k_expand_b<false> is through (ev_a); the sorter behind it is joined where its result is
          // needed, in front of k_classify: the small candidate kernels run beside its tail
          if (!m.serial) { HIP_TRY(hipEventRecord(c.ev_a, side)); big_pending = true; }
          launch_big(side);
        }
        if (light_max) {
          if (!m.serial) hipLaunchKernelGGL(k_lag, dim3(1), dim3(kWave), 0, stream, w.counters);
          launch_light();
        }
        tick(); tick();
      }
      // stage "expand_wait": what k_expand_b<false> still has to do when the light kernel is through
      if (big_pending) HIP_TRY(hipStreamWaitEvent(stream, c.ev_a, 0));
      else join();
    }
    tick(); tick();
    // fin_list is free until k_cmap fills it: it carries k_cand's few queries
    hipLaunchKernelGGL(k_cand_route, dim3((b.nq + 255) / 256), dim3(256), 0, stream, b, w, c.big_list.p, c.mid_list.p, c.fin_list.p, early_big ? 1 : 0);
    hipLaunchKernelGGL(k_cand, dim3(std::min<uint32_t>(2048u, std::max<uint32_t>(64u, b.nq))), dim3(kWave), 0, stream, w, (const uint32_t*)c.fin_list.p);
    {
      // The sorter of the few huge survivor sets is one workgroup per query and leaves the GPU almost idle:
      // it runs on the side stream next to k_cand_mid (disjoint queries).
      // radix passes needed: sentence ids < n_sent, match lengths <= longest pattern
      uint32_t sid_bytes = 1, len_bytes = p_max < 255 ? 1u : 2u;
      while (sid_bytes < 4 && (E.d.n_sent >> (8 * sid_bytes)) != 0) sid_bytes++;
      if (!early_big) {
        fork();
        hipLaunchKernelGGL(k_cand_big, dim3(kBigGrid), dim3(kBigThreads), 0, side, w, c.keys.p,
                           (const uint32_t*)c.big_list.p, sid_bytes, len_bytes);
      }
    }
    hipLaunchKernelGGL(k_cand_mid, dim3(1024), dim3(256), 0, stream, w, (const uint32_t*)c.mid_list.p);
    tick(); tick();
    join();  // stage "cand_big" = what is left of the big sorter after k_cand_mid
    tick(); tick();
    {
      // dense per-class candidate lists: three exclusive scans of the per-query candidate counts
      hipLaunchKernelGGL(k_classify, dim3((n_scan + 255) / 256), dim3(256), 0, stream, b, w, dp_mode);
      hipLaunchKernelGGL(k_scan_lb<3>, dim3(scan_tiles), dim3(kScanThreads), 0, stream, w.c_in, w.c_off, (size_t)n_scan, n_scan,
                         c.scan_state(2), c.scan_ticket(2));
      hipLaunchKernelGGL(k_cmap, dim3((b.nq + 1023) / 1024), dim3(1024), 0, stream, b, w, dp_mode, c.fin_list.p, pair_finish ? pair_max : (uint32_t)kWave);
    }
    tick(); tick();
    {
      const size_t lds = ((size_t)p_max + 1) * kWave * 4;
      const bool rows_in_hbm = lds > kDynLdsMax;  // patterns too long for LDS rows: same kernel over rows in HBM
      static const unsigned dp_per_cu = [] { const char* e = getenv("FMX_DP_GRID"); return e ? (unsigned)atoi(e) : 16u; }();  // design-time switch
      const unsigned grid = 256 * dp_per_cu;  // grid-stride over the flat candidate list
      const unsigned grid_g = 1024;
      if (rows_in_hbm) c.g_rows.ensure((uint64_t)grid_g * kWave * (p_max + 1));
      auto launch_rows = [&](bool with_side, const unsigned long long* ext_list = nullptr) {
        if (with_side) {
          if (rows_in_hbm) hipLaunchKernelGGL((k_dp<true, true>), dim3(grid_g), dim3(kWave), 0, stream, E.d, b, w, prm, c.g_rows.p, p_max + 1, ext_list);
          else hipLaunchKernelGGL((k_dp<true, false>), dim3(grid), dim3(kWave), lds, stream, E.d, b, w, prm, (float*)nullptr, 0u, ext_list);
        } else {
          if (rows_in_hbm) hipLaunchKernelGGL((k_dp<false, true>), dim3(grid_g), dim3(kWave), 0, stream, E.d, b, w, prm, c.g_rows.p, p_max + 1, ext_list);
          else hipLaunchKernelGGL((k_dp<false, false>), dim3(grid), dim3(kWave), lds, stream, E.d, b, w, prm, (float*)nullptr, 0u, ext_list);
        }
      };
      if (dp_mode >= 2) {
        // real tokens: rows + both pattern streams in registers; pairs with penalty tokens (only looked for when the index has
        // interned penalty strings) are handed over in c.keys — free between the candidate sorters and k_finish — to the general kernel
        const int check_itok = E.d.n_itok > 1 ? 1 : 0;
        const bool two = p_max > 32;
        if (two) fork();
        const bool hybrid_on = m.real_hybrid;  // option "real_hybrid" (off by default)
        if (dp_mode == 3) {
          if (two) hipLaunchKernelGGL((k_dp_reg<48, true, false, true>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm, c.keys.p, check_itok);
          if (p_max > 48) hipLaunchKernelGGL((k_dp_reg<64, true, false, true, 49>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm, c.keys.p, check_itok);
          hipLaunchKernelGGL((k_dp_reg<32, true, false, true>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm, c.keys.p, check_itok);
        } else {
          if (two) hipLaunchKernelGGL((k_dp_reg<48, false, false, true>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm, c.keys.p, check_itok);
          if (p_max > 48) hipLaunchKernelGGL((k_dp_reg<64, false, false, true, 49>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm, c.keys.p, check_itok);
          const bool unit = prm.ins == prm.del && prm.del == prm.rep && prm.rep >= 0.f;
          if (hybrid_on && unit && !check_itok && c.seg_cap >= 4096) {
            // unit costs, no penalty tokens anywhere: the bit-parallel kernel answers the pairs in which no equal word meets a
            // different real token (their cost is the ids' cost) and marks the others; a scan + a scatter make the marked ones a
            // dense list in list order; the float kernel computes those.  c.keys: [0, seg_cap) the list, then flags / counts / offsets.
            const uint32_t n_chunks = (uint32_t)(c.seg_cap / kWave) + 1;
            unsigned long long* flags = c.keys.p + c.seg_cap;
            unsigned long long* cnt = flags + (n_chunks + 1);
            unsigned long long* offs = cnt + (n_chunks + 1);
            HIP_TRY(hipMemsetAsync(cnt, 0, (size_t)(n_chunks + 1) * sizeof(unsigned long long), stream));
            hipLaunchKernelGGL(k_dp_tab<true>, dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm, flags, cnt, n_chunks);
            const uint32_t n_sc = n_chunks + 1;
            hipLaunchKernelGGL(k_scan_lb<1>, dim3((n_sc + kScanTile - 1) / kScanTile), dim3(kScanThreads), 0, stream, (const unsigned long long*)cnt, offs, (size_t)n_sc, n_sc,
                               c.scan_state(4), c.scan_ticket(4));
            hipLaunchKernelGGL(k_dp_compact, dim3(2048), dim3(256), 0, stream, b, w, (const unsigned long long*)flags, (const unsigned long long*)offs, c.keys.p);
            hipLaunchKernelGGL((k_dp_reg<32, false, false, true>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm, (unsigned long long*)nullptr, 0,
                               (const unsigned long long*)c.keys.p, (const unsigned long long*)(offs + n_chunks));
          } else
            hipLaunchKernelGGL((k_dp_reg<32, false, false, true>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm, c.keys.p, check_itok);
        }
        if (two) join();
        if (p_max > 64) launch_rows(true);
        if (check_itok) launch_rows(true, c.keys.p);
      } else if (dp_mode == 1) {
        launch_rows(true);
      } else {
        // by pattern length: row + pattern in 32 or 64 registers, longer ones with the row in LDS
        // the length classes are disjoint lists: the kernels of the (few) patterns of 33..48 / 49..64 tokens run on the side stream
        // beside the bulk one instead of behind it (a small pass waited ~0.15 ms for a grid that fills a tenth of the GPU)
        const bool two = p_max > 32;
        if (two) fork();
        if (use_idf) {
          if (two) hipLaunchKernelGGL((k_dp_reg<48, true>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm);
          if (p_max > 48) hipLaunchKernelGGL((k_dp_reg<64, true, false, false, 49>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm);
          hipLaunchKernelGGL((k_dp_reg<32, true>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm);
        } else if (prm.ins == prm.del && prm.del == prm.rep && prm.rep >= 0.f) {
          // unit costs: bit-parallel Levenshtein distance, then the float sum the DP would have produced
          // few registers: eight waves per SIMD hide the dependent loads in front of every candidate
          if (two) hipLaunchKernelGGL((k_dp_reg<64, false, true>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm);
          // the 32-token class: match masks from a per-wave LDS table of the queries' patterns (k_dp_tab, round 5) instead of
          // two compare instructions per pattern token and row (design-time switch FMX_DP_TAB=0: the register kernel)
          static const bool dp_tab = [] { const char* e = getenv("FMX_DP_TAB"); return e ? atoi(e) != 0 : true; }();
          if (dp_tab) hipLaunchKernelGGL(k_dp_tab<false>, dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm);
          else hipLaunchKernelGGL((k_dp_reg<32, false, true>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm);
        } else {
          if (two) hipLaunchKernelGGL((k_dp_reg<48, false>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm);
          if (p_max > 48) hipLaunchKernelGGL((k_dp_reg<64, false, false, false, 49>), dim3(grid), dim3(kWave), 0, side, E.d, b, w, prm);
          hipLaunchKernelGGL((k_dp_reg<32, false>), dim3(grid), dim3(kWave), 0, stream, E.d, b, w, prm);
        }
        if (p_max > 64) launch_rows(use_idf);
        if (two) join();
      }
    }
    tick(); tick();
    {
      // the queue of large candidate sets was filled by k_cmap: their (long, mostly sequential) replay runs on the
      // side stream next to the bulk kernel; the two touch disjoint queries
      fork();
      hipLaunchKernelGGL((k_finish<kLdsSortMax, true>), dim3(2048), dim3(kWave), 0, side, E.d, b, w, prm, DevScratch{c.keys.p, c.vals.p, c.f0.p, c.f1.p}, cap,
                         d_out, d_out_count, d_out_total, c.contrast_list.p, c.fin_list.p);
      if (pair_finish)
        hipLaunchKernelGGL(k_finish_pair, dim3((b.nq + 3) / 4), dim3(kWave), 0, stream, b, w, prm, cap, d_out, d_out_count, d_out_total, pair_max,
                           c.contrast_list.p);
      else
        hipLaunchKernelGGL((k_finish<64, false>), dim3(b.nq), dim3(kWave), 0, stream, E.d, b, w, prm, DevScratch{c.keys.p, c.vals.p, c.f0.p, c.f1.p}, cap, d_out,
                         d_out_count, d_out_total, c.contrast_list.p, c.fin_list.p);
      join();
      const DevScratch sc{c.keys.p, c.vals.p, c.f0.p, c.f1.p};
      if (prm.contrast > 0.f) {
        const uint32_t* cl = c.contrast_list.p;
        const bool flat = prm.nmatch >= 1 && prm.nmatch <= kMaxFlatRounds;
        const uint32_t row_tokens = std::max<uint32_t>(E.d.max_slen, E.prior_max_len);  // longest "pattern" side of the rerank's DP
        if (flat) {
          // accepted sets of up to 32 matches (sentences of up to 32 tokens): every round inside one launch, a group of 8 / 16 / 32
          // lanes per query with the picked sentence's match masks in a small LDS table (k_crerank_g); the big / cand lists of the
          // candidate stage are free by now and carry the three query lists
          static const bool groups_on = [] { const char* e = getenv("FMX_RERANK_GROUPS"); return e ? atoi(e) != 0 : true; }();  // design-time switch
          static const bool g32_side = [] { const char* e = getenv("FMX_G32_SIDE"); return e ? atoi(e) != 0 : true; }();  // design-time switch
          if (groups_on) hipLaunchKernelGGL(k_cgrp_route, dim3((b.nq + 1023) / 1024), dim3(1024), 0, stream, b, w, c.big_list.p, c.mid_list.p, c.fin_list.p, c.contrast_list.p);
          // The flattened rounds (the rest: sentences of 33-64 tokens; without the groups everything) are chains of small launches —
          // pick, then one edit distance per remaining accepted match of any query; round r: the match picked in round r - 1 emits
          // itself, the others bid for round r's pick.  They touch other queries than the group kernels, so with the groups on the
          // chain runs on the side stream BESIDE them (until round 4 it followed them: 0.9 ms of the stage).  The look-back scan stays
          // in front of the fork: beside a kernel that fills every CU its tiles wait for predecessors that are not resident (0.5 ms
          // instead of 25 us).
          hipLaunchKernelGGL(k_cacc, dim3((n_scan + 255) / 256), dim3(256), 0, stream, b, w);
          hipLaunchKernelGGL(k_scan_lb<2>, dim3(scan_tiles), dim3(kScanThreads), 0, stream, w.c_in, w.c_off, (size_t)n_scan, n_scan,
                             c.scan_state(3), c.scan_ticket(3));
          if (groups_on) {
            fork();
            hipLaunchKernelGGL(k_camap, dim3((b.nq + kWave / kCamapGroup - 1) / (kWave / kCamapGroup)), dim3(kWave), 0, side, E.d, b, w, sc, prm, d_out_count, d_out_total);
            const unsigned gg = std::min<unsigned>(256u * 32u, std::max<unsigned>(64u, b.nq / 4u));
            // (round 6) per-query dictionary ids: the group kernels' DICT variant first; the few groups whose sentences do not fit their
            // dictionary come back in fb lists, which the hash-table variant takes in small launches behind
            static const bool dict_on = [] { const char* e = getenv("FMX_RERANK_DICT"); return e ? atoi(e) != 0 : true; }();  // design-time switch
            uint32_t* fb = c.fb_list.p;
            const uint32_t nqs = b.nq;
            auto grp = [&](auto kern_d, auto kern_h, unsigned grid_, hipStream_t st, const uint32_t* lst, uint32_t which) {
              if (dict_on) {
                hipLaunchKernelGGL(kern_d, dim3(grid_), dim3(kWave), 0, st, E.d, b, w, prm, lst, 28u + which, cap, d_out, d_out_count, d_out_total, fb + (size_t)which * nqs,
                                   (uint32_t)kCtrRerankFb + which);
                hipLaunchKernelGGL(kern_h, dim3(std::max(64u, grid_ / 16u)), dim3(kWave), 0, st, E.d, b, w, prm, (const uint32_t*)(fb + (size_t)which * nqs),
                                   (uint32_t)kCtrRerankFb + which, cap, d_out, d_out_count, d_out_total, (uint32_t*)nullptr, 0u);
              } else
                hipLaunchKernelGGL(kern_h, dim3(grid_), dim3(kWave), 0, st, E.d, b, w, prm, lst, 28u + which, cap, d_out, d_out_count, d_out_total, (uint32_t*)nullptr, 0u);
            };
            grp(k_crerank_g<8, true>, k_crerank_g<8, false>, gg, stream, (const uint32_t*)c.big_list.p, 0u);
            grp(k_crerank_g<16, true>, k_crerank_g<16, false>, gg, stream, (const uint32_t*)c.mid_list.p, 1u);
            if (E.d.max_slen > 32)
              for (uint32_t r = 1; r <= prm.nmatch; r++)
                hipLaunchKernelGGL(k_cround<64>, dim3(cround_grid(64)), dim3(kWave), 0, side, E.d, b, w, prm, sc, r, cap, d_out, d_out_count, d_out_total);
            // what is left of the 32-token class are sets of more than 64 accepted matches: a small grid
            for (uint32_t r = 1; r <= prm.nmatch; r++)
              hipLaunchKernelGGL(k_cround<32>, dim3(1024), dim3(kWave), 0, side, E.d, b, w, prm, sc, r, cap, d_out, d_out_count, d_out_total);
            grp(k_crerank_g<32, true>, k_crerank_g<32, false>, gg / 2, g32_side ? side : stream, (const uint32_t*)c.fin_list.p, 2u);
            grp(k_crerank_g<64, true>, k_crerank_g<64, false>, gg, stream, (const uint32_t*)c.contrast_list.p, 3u);
            join();
          } else {
            hipLaunchKernelGGL(k_camap, dim3((b.nq + kWave / kCamapGroup - 1) / (kWave / kCamapGroup)), dim3(kWave), 0, stream, E.d, b, w, sc, prm, d_out_count, d_out_total);
            if (E.d.max_slen > 32) {
              fork();
              for (uint32_t r = 1; r <= prm.nmatch; r++)
                hipLaunchKernelGGL(k_cround<64>, dim3(cround_grid(64)), dim3(kWave), 0, side, E.d, b, w, prm, sc, r, cap, d_out, d_out_count, d_out_total);
            }
            for (uint32_t r = 1; r <= prm.nmatch; r++)
              hipLaunchKernelGGL(k_cround<32>, dim3(cround_grid(32)), dim3(kWave), 0, stream, E.d, b, w, prm, sc, r, cap, d_out, d_out_count, d_out_total);
            if (E.d.max_slen > 32) join();
          }
        } else {
          hipLaunchKernelGGL(k_contrast<0>, dim3(b.nq), dim3(kWave), 0, stream, E.d, b, w, prm, sc, cap, d_out,
                             d_out_count, d_out_total, cl, (float*)nullptr, 0u);
        }
        if (E.d.max_slen > 32 && !flat)
          hipLaunchKernelGGL(k_contrast<1>, dim3(2048), dim3(kWave), 0, stream, E.d, b, w, prm, sc, cap, d_out,
                             d_out_count, d_out_total, cl, (float*)nullptr, 0u);
        {  // long sentences, and the queries with more accepted matches than the flattened rerank indexes (k_finish queues both)
          const size_t lds = ((size_t)row_tokens + 1) * kWave * 4;
          if (lds > kDynLdsMax - 8 * 1024) {  // indexed sentences too long for LDS rows: rows in HBM
            const unsigned grid_g = 512;
            c.g_rows.ensure((uint64_t)grid_g * kWave * (row_tokens + 1));
            hipLaunchKernelGGL(k_contrast<3>, dim3(grid_g), dim3(kWave), 0, stream, E.d, b, w, prm, sc, cap, d_out,
                               d_out_count, d_out_total, cl, c.g_rows.p, row_tokens + 1);
          } else {
            hipLaunchKernelGGL(k_contrast<2>, dim3(512), dim3(kWave), lds, stream, E.d, b, w, prm, sc, cap, d_out,
                               d_out_count, d_out_total, cl, (float*)nullptr, 0u);
          }
        }
      }
    }
    tick();
    hipLaunchKernelGGL(k_stats, dim3((b.nq + 1023) / 1024), dim3(1024), 0, stream, w, b.nq);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(h_ctr_slot, w.counters, kNumCounters * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
}

// the limits a pass was enqueued with (several passes of an asynchronous group run before any of their counters is read)
struct PassLimits {
  uint32_t s_max, heavy_div;
  uint64_t seg_cap, qb_cap, item_cap, item_h_cap;
};
static PassLimits limits_of(const Ctx& c) {
  return PassLimits{c.s_max, c.heavy_div, c.seg_cap, c.qb.cap, std::min<uint64_t>(c.items.cap, 0xffffffffull),
                    std::min<uint64_t>(c.items_h.cap, 0xffffffffull)};
}

// what the counters of a finished pass ask for: true = re-run it (something was too small for it under the limits `ran`
// it was enqueued with; the context's limits are raised, never lowered)
static void learn_sizes(DeviceIndex::Impl& m, const unsigned long long* ctr, uint64_t nq_pass) {
  auto raise = [](std::atomic<uint64_t>& a, uint64_t v) { uint64_t cur = a.load(); while (v > cur && !a.compare_exchange_weak(cur, v)) {} };
  if (!nq_pass) return;
  raise(m.hint_seg_kq, (std::max(ctr[24], ctr[6]) * 1024 + nq_pass - 1) / nq_pass);
  raise(m.hint_qb_kq, (std::max(ctr[25], ctr[21]) * 1024 + nq_pass - 1) / nq_pass);
}

static bool pass_needs_retry(DeviceIndex::Impl& m, Ctx& c, const unsigned long long* ctr, const PassLimits& ran, uint64_t& want_seg) {
  bool retry = false;
  if (ctr[6] > ran.seg_cap) { want_seg = std::max<uint64_t>(want_seg, ctr[6] + ctr[6] / 4); retry = true; }   // segments did not fit
  if (ctr[21] > ran.qb_cap) { c.want_qb = std::max<uint64_t>(c.want_qb, ctr[21] + ctr[21] / 4); retry = true; }  // queues did not fit
  if (ctr[4] > ran.s_max) {                                                        // a query had more survivors
    unsigned long long s = std::max<uint32_t>(c.s_max, ran.s_max);
    while (s < ctr[4] && s < (1ull << 31)) s <<= 1;
    if (ctr[4] > s) throw UnsupportedError("a query kept more than 2^31 hits");
    c.s_max = std::max<uint32_t>(c.s_max, (uint32_t)s);
    { uint32_t cur = m.hint_s_max.load(); while (c.s_max > cur && !m.hint_s_max.compare_exchange_weak(cur, c.s_max)) {} }
    retry = true;
  } else if (ctr[4] != 0) {                                                        // more of its long-range hits passed than reserved
    if (ran.heavy_div == 1) throw GpuError("a query queued more hits than it has");
    c.heavy_div = std::min<uint32_t>(c.heavy_div, ran.heavy_div > 4 ? 4u : 1u);
    { uint32_t cur = m.hint_heavy_div.load(); while ((cur == 0 || c.heavy_div < cur) && !m.hint_heavy_div.compare_exchange_weak(cur, c.heavy_div)) {} }
    retry = true;
  }
  if (ctr[26] != 0)  // k_setup: q_off[first query] / q_off[last query + 1] are not what the pass was enqueued with
    throw std::invalid_argument("device batch: total_tokens / offsets disagree with q_off on the device (q_off[0] must be 0 and q_off[nq] == total_tokens)");
  if (std::max(ctr[5], ctr[12]) > ran.item_cap) {                                  // work-item list too short
    const unsigned long long need = std::max(ctr[5], ctr[12]);
    const uint64_t nc = std::min<unsigned long long>(need + need / 4, 0xffffffffull);
    c.items.ensure(nc); c.items_b.ensure(nc);
    retry = true;
  }
  if (ctr[15] > ran.item_h_cap) {                                                  // heavy-range item list too short
    c.items_h.ensure(std::min<unsigned long long>(ctr[15] + ctr[15] / 4, 0xffffffffull));
    retry = true;
  }
  return retry;
}

static void add_counters(Stats* st, const unsigned long long* ctr) {
  if (!st) return;
  st->n_hits += ctr[0]; st->n_kept += ctr[1]; st->n_candidates += ctr[2]; st->n_dp += ctr[3];
  st->n_big += ctr[7]; st->n_phase_b += ctr[8]; st->max_kept = std::max<uint64_t>(st->max_kept, ctr[9]);
  st->n_heavy_hits += ctr[16]; st->n_heavy_pass += ctr[17]; st->plane_words += ctr[18];
  st->probe_bytes += ctr[19]; st->probe_lines += ctr[20]; st->n_light_hits += ctr[22]; st->n_rerank += ctr[27];
}

constexpr uint32_t kAsyncGroup = 32;  // passes enqueued before the host looks at their counters

// One call's passes over a context.  async: the passes of a group are enqueued back to back (the host runs ahead of the GPU,
// no bubble between passes); their counters are checked afterwards and a pass whose workspace was too small is re-run
// (first calls only).  Otherwise (device entry point with stage timing) each pass is synchronised and timed.
// begin_group / end_group are the two halves of a group, so that a caller may return to its own caller in between
// (fmx_match_batch_device_begin / _end: several calls enqueued by ONE host thread).
struct PassRun {
  DeviceIndex::Impl& m;
  Ctx& c;
  DeviceBatchView bv;
  std::vector<uint64_t> sub_q_bounds;
  std::vector<uint32_t> sub_tok_bounds;
  Stats* st;
  PassHooks hooks;  // members may be empty
  PassEnv E;
  uint32_t n_sub;
  uint64_t nq_eff, nt_eff;
  PassLimits ran{};
  hipStream_t stream;

  PassRun(DeviceIndex::Impl& m_, Ctx& c_, const Params& prm_in, const DeviceBatchView& bv_, Rec* d_out, uint32_t* d_out_count,
          uint32_t* d_out_total, uint32_t cap, hipStream_t stream_, std::vector<uint64_t> q_bounds, std::vector<uint32_t> tok_bounds,
          Stats* st_, const PassHooks* hooks_)
      : m(m_), c(c_), bv(bv_), sub_q_bounds(std::move(q_bounds)), sub_tok_bounds(std::move(tok_bounds)), st(st_),
        hooks(hooks_ ? *hooks_ : PassHooks{}),
        // cost model: real-token / penalty-token terms need the general kernel (DP rows in LDS / HBM); idf alone has a register kernel
        E{m_, c_, TableSnap{}, DevIndex{}, prm_in, (m_.d.real && bv_.q_real) || (bv_.q_itok != nullptr) || (m_.d.itok != nullptr), prm_in.idf_penalty != 0.f, d_out,
          d_out_count, d_out_total, cap, stream_, bv_.prior_max_len},
        stream(stream_) {
    E.snap = pin_tables(m);
    E.d = dev_view(m, E.snap);
    if (cap == 0) throw std::invalid_argument("cap must be > 0");
    if (!c.qb.p) c.want_qb = std::min<uint64_t>(c.want_qb, std::max<uint64_t>(m.init_seg, 4096));  // a small initial segment also starts the queues small
    n_sub = (uint32_t)sub_tok_bounds.size() - 1;
    // the largest pass this call could be cut into (the cut depends on what else is in flight): the workspace is sized for it
    nq_eff = std::min<uint64_t>(std::max<uint64_t>(bv.nq, 1), m.sub_queries);
    nt_eff = (uint64_t)((double)(sub_tok_bounds.back() - sub_tok_bounds.front()) * ((double)nq_eff / (double)std::max<uint64_t>(bv.nq, 1)) * 1.05) + 64;
    c.want_qb = std::max<uint64_t>(c.want_qb, hinted(m.hint_qb_kq));
    c.s_max = std::max<uint32_t>(c.s_max, m.hint_s_max.load());
    if (const uint32_t hd = m.hint_heavy_div.load()) c.heavy_div = std::min<uint32_t>(c.heavy_div, hd);
  }
  uint64_t hinted(const std::atomic<uint64_t>& kq) const { const uint64_t h = kq.load(); return h ? (h * nq_eff / 1024) + (h * nq_eff / 1024) / 8 + 1024 : 0ull; }
  DevBatch batch_of(uint32_t sb) const {
    DevBatch b;
    b.q_off = bv.q_off; b.q_tok = bv.q_tok; b.q_real = bv.q_real; b.q_itok = bv.q_itok;
    b.prior_off = bv.prior_off; b.prior_tok = bv.prior_tok; b.n_prior = bv.n_prior; b.prior_sid = bv.prior_sid;
    b.q_begin = (uint32_t)sub_q_bounds[sb];
    b.nq = (uint32_t)(sub_q_bounds[sb + 1] - sub_q_bounds[sb]);
    b.t_begin = sub_tok_bounds[sb];
    b.nt = sub_tok_bounds[sb + 1] - sub_tok_bounds[sb];
    return b;
  }
  uint32_t p_max_of(const DevBatch& b) const {
    uint32_t p_max = (uint32_t)std::min<uint64_t>(E.prm.max_tokens, std::max<uint32_t>(b.nt, 1));
    return std::max<uint32_t>(1, std::min<uint32_t>(p_max, bv.p_max ? bv.p_max : p_max));
  }
  uint64_t seg_for(const DevBatch& b) const {
    return std::max<uint64_t>(std::max<uint64_t>(c.seg_cap, hinted(m.hint_seg_kq)),
                              std::min<uint64_t>(m.init_seg, std::max<uint64_t>(4096, (uint64_t)b.nq * 64)));
  }
  // one pass, synchronised, re-run until its workspace fits
  void run_sync(uint32_t sb, bool call_before, uint64_t want_seg) {
    const DevBatch b = batch_of(sb);
    const uint32_t p_max = p_max_of(b);
    want_seg = std::max<uint64_t>(want_seg, seg_for(b));
    if (call_before && hooks.before) hooks.before(sb);
    for (int attempt = 0;; attempt++) {
      ensure_ws(c, std::max<uint64_t>(b.nq, nq_eff), std::max<uint64_t>(std::max<uint32_t>(b.nt, 1), nt_eff), want_seg);
      c.cmin_tab.ensure((uint64_t)(p_max + 1) * kCminRow);
      const PassLimits lim = limits_of(c);
      enqueue_pass(E, b, p_max, st != nullptr, c.h_ctr);
      HIP_TRY(hipStreamSynchronize(stream));
      if (st) {
        for (int s = 0; s < kNumStages; s++) {
          float ms = 0.f;
          HIP_TRY(hipEventElapsedTime(&ms, c.ev[2 * s], c.ev[2 * s + 1]));
          st->ms_kernel[s] += ms;
          st->launches[s] += 1;
        }
      }
      learn_sizes(m, c.h_ctr, b.nq);
      if (pass_needs_retry(m, c, c.h_ctr, lim, want_seg)) {
        if (attempt >= 6) throw GpuError("workspace kept overflowing");
        if (st) st->n_retries++;
        continue;
      }
      add_counters(st, c.h_ctr);
      break;
    }
    if (hooks.after) hooks.after(sb, nullptr);
  }
  void run_all_sync() { for (uint32_t sb = 0; sb < n_sub; sb++) run_sync(sb, true, 0); }
  void prepare_async() {
    if (!c.h_ctr_group) HIP_TRY(hipHostMalloc((void**)&c.h_ctr_group, (size_t)kAsyncGroup * kNumCounters * sizeof(unsigned long long), hipHostMallocDefault));
    while (c.ev_pass.size() < kAsyncGroup) {
      hipEvent_t e;
      // the host waits milliseconds on these: it sleeps instead of spinning (one process per GPU with three calls in flight
      // each would otherwise keep 24 cores of an 8-GPU node busy doing nothing)
      HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventBlockingSync));
      c.ev_pass.push_back(e);
    }
  }
  // every launch of passes [g0, g1), nothing waited for
  void begin_group(uint32_t g0, uint32_t g1) {
    // the workspace of the largest pass of the group, once (growing it later would wait for the passes in flight)
    uint64_t nq_max = 1, nt_max = 1, seg_max = 0;
    uint32_t pm_max = 1;
    for (uint32_t sb = g0; sb < g1; sb++) {
      const DevBatch b = batch_of(sb);
      nq_max = std::max<uint64_t>(nq_max, b.nq); nt_max = std::max<uint64_t>(nt_max, b.nt);
      seg_max = std::max<uint64_t>(seg_max, seg_for(b)); pm_max = std::max(pm_max, p_max_of(b));
    }
    ensure_ws(c, std::max(nq_max, nq_eff), std::max(nt_max, nt_eff), seg_max);
    c.cmin_tab.ensure((uint64_t)(pm_max + 1) * kCminRow);
    ran = limits_of(c);  // every pass of the group runs under these
    for (uint32_t sb = g0; sb < g1; sb++) {
      const DevBatch b = batch_of(sb);
      if (hooks.before) hooks.before(sb);
      enqueue_pass(E, b, p_max_of(b), false, c.h_ctr_group + (size_t)(sb - g0) * kNumCounters);
      HIP_TRY(hipEventRecord(c.ev_pass[sb - g0], stream));
    }
    if (g0 == 0 && hooks.enqueued) {
      HIP_TRY(hipEventRecord(c.ev_chain, stream));
      hooks.enqueued(c.ev_chain);
    }
  }
  // waits for the group, hands the records on, re-runs what did not fit
  void end_group(uint32_t g0, uint32_t g1) {
    // The records of a pass are sent home by the host once the pass is through, not queued behind an event: a copy that
    // waits in a DMA queue for its kernels holds up every copy queued after it — the inputs of the other calls in flight.
    for (uint32_t sb = g0; sb < g1; sb++) {
      HIP_TRY(hipEventSynchronize(c.ev_pass[sb - g0]));
      if (sb + 1 == g1 && hooks.computed) hooks.computed();
      if (hooks.after) hooks.after(sb, nullptr);
    }
    HIP_TRY(hipStreamSynchronize(stream));
    uint64_t want_seg = 0;
    std::vector<uint32_t> redo;
    for (uint32_t sb = g0; sb < g1; sb++) {
      const unsigned long long* ctr = c.h_ctr_group + (size_t)(sb - g0) * kNumCounters;
      learn_sizes(m, ctr, sub_q_bounds[sb + 1] - sub_q_bounds[sb]);
      if (pass_needs_retry(m, c, ctr, ran, want_seg)) redo.push_back(sb);
      else add_counters(st, ctr);
    }
    for (uint32_t sb : redo) {
      if (st) st->n_retries++;
      run_sync(sb, false, want_seg);
    }
  }
  void run_all_async() {
    prepare_async();
    for (uint32_t g0 = 0; g0 < n_sub; g0 += kAsyncGroup) {
      const uint32_t g1 = std::min<uint32_t>(g0 + kAsyncGroup, n_sub);
      begin_group(g0, g1);
      end_group(g0, g1);
    }
  }
};

static void run_passes(DeviceIndex::Impl& m, Ctx& c, const Params& prm_in, const DeviceBatchView& bv, Rec* d_out,
                       uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap, hipStream_t stream,
                       const std::vector<uint64_t>& sub_q_bounds, const std::vector<uint32_t>& sub_tok_bounds, Stats* st,
                       const PassHooks* hooks, bool async_ok) {
  PassRun r(m, c, prm_in, bv, d_out, d_out_count, d_out_total, cap, stream, sub_q_bounds, sub_tok_bounds, st, hooks);
  if (async_ok) r.run_all_async();
  else r.run_all_sync();
}

// fmx_match_batch_device_begin / _end: the two halves of the stats-less device call, so that ONE host thread can keep several
// calls in flight (each in its own context; chained on the GPU like concurrent callers are).
struct DeviceIndex::Pending {
  std::unique_ptr<CtxLease> lease;
  std::unique_ptr<PassRun> run;
  bool done = false;
  DeferredCall dc;  // begun when no context could be taken: queued on the host (Impl::deferred)
  ~Pending() { if (run && !done) (void)hipStreamSynchronize(run->stream); }  // never ended (an error in between): nothing may still run in the context when the lease goes
};

DeviceIndex::Pending* DeviceIndex::match_device_begin(const HostIndex& h, const Params& prm, const DeviceBatchView& bv, Rec* d_out,
                                                      uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap, void* stream_v,
                                                      const std::vector<uint32_t>& sub_tok_bounds) {
  Impl& m = *impl_;
  if (m.device < 0) throw GpuError("index is not on a device");
  HIP_TRY(hipSetDevice(m.device));
  std::vector<uint64_t> q_bounds;
  for (uint64_t q = 0; q < bv.nq; q += m.sub_queries) q_bounds.push_back(q);
  q_bounds.push_back(bv.nq);
  if (q_bounds.size() != sub_tok_bounds.size()) throw std::invalid_argument("sub-batch bounds do not match the batch");
  std::unique_ptr<Pending> p(new Pending);
  Pending* pp = p.get();
  Impl* mp = &m;
  const HostIndex* hp = &h;
  // everything that enqueues the call (a deferred call runs this on the thread that freed its slot)
  p->dc.host_kind = false;
  p->dc.start = [pp, mp, hp, prm, bv, d_out, d_out_count, d_out_total, cap, stream_v, q_bounds, sub_tok_bounds](bool force) {
    Impl& m = *mp;
    HIP_TRY(hipSetDevice(m.device));
    refresh_itok_table(m, *hp);
    pp->lease.reset(new CtxLease(m, /*no_wait=*/true, /*host_kind=*/false, force));  // NoContext instead of waiting: the busy contexts may all be pendings of this thread
    Ctx& c = *pp->lease->c;
    hipStream_t stream = stream_v ? (hipStream_t)stream_v : c.stream;  // NULL: the context's stream, the caller's inputs are complete
    const uint32_t turns = bv.nq >= kTurnMinQueries ? m.device_turns : 0u;
    std::unique_lock<std::mutex> chain_lk(m.chain_mu, std::defer_lock);
    if (turns == 1 || turns == 2) {
      chain_lk.lock();
      if (m.chain_last) HIP_TRY(hipStreamWaitEvent(stream, m.chain_last, 0));
    }
    PassHooks hooks;
    Ctx* cp = &c;
    // (chain_last belongs to chain_mu: written only by a chained call, which holds the lock until its launches are queued)
    const bool chain_held = chain_lk.owns_lock();
    hooks.enqueued = [mp, cp, turns, chain_held](hipEvent_t last) {
      if (chain_held) mp->chain_last = (turns == 2 && mp->expand_order != 0 && !mp->serial) ? cp->ev_mid : last;
    };
    pp->run.reset(new PassRun(m, c, prm, bv, d_out, d_out_count, d_out_total, cap, stream, q_bounds, sub_tok_bounds, nullptr, &hooks));
    pp->run->hooks.enqueued = nullptr;  // (re-armed below: the hook captures locals of this frame)
    pp->run->prepare_async();
    if (pp->run->n_sub <= kAsyncGroup) {
      pp->run->hooks.enqueued = hooks.enqueued;
      pp->run->begin_group(0, pp->run->n_sub);
      pp->run->hooks.enqueued = nullptr;
    } else {  // more passes than one group holds: the call runs to its end here
      pp->run->hooks.enqueued = hooks.enqueued;
      pp->run->run_all_async();
      pp->run->hooks.enqueued = nullptr;
      pp->done = true;
    }
  };
  begin_or_defer(m, p->dc);
  return p.release();
}

void DeviceIndex::match_device_end(Pending* pending) {
  std::unique_ptr<Pending> p(pending);
  if (!p) return;
  HIP_TRY(hipSetDevice(impl_->device));
  ensure_started(*impl_, p->dc);
  if (!p->done) { p->run->end_group(0, p->run->n_sub); p->done = true; }
}

void DeviceIndex::match_device_abandon(Pending* pending) {
  std::unique_ptr<Pending> p(pending);
  if (!p) return;
  try { ensure_started(*impl_, p->dc); } catch (...) {}
  // (~Pending waits for the stream)
}

void DeviceIndex::match_device(const HostIndex& h, const Params& prm, const DeviceBatchView& bv, Rec* d_out,
                               uint32_t* d_out_count, uint32_t* d_out_total, uint32_t cap, void* stream_v,
                               const std::vector<uint32_t>& sub_tok_bounds, Stats* st) {
  Impl& m = *impl_;
  if (m.device < 0) throw GpuError("index is not on a device");
  HIP_TRY(hipSetDevice(m.device));
  refresh_itok_table(m, h);
  CtxLease lease(m);
  Ctx& c = *lease.c;
  // the caller's stream carries the work (NULL = the context's own stream)
  hipStream_t stream = stream_v ? (hipStream_t)stream_v : c.stream;
  if (!stream_v) HIP_TRY(hipDeviceSynchronize());  // inputs may have been produced on other streams
  std::vector<uint64_t> q_bounds;
  for (uint64_t q = 0; q < bv.nq; q += m.sub_queries) q_bounds.push_back(q);
  q_bounds.push_back(bv.nq);
  if (q_bounds.size() != sub_tok_bounds.size()) throw std::invalid_argument("sub-batch bounds do not match the batch");
  if (st) {  // per-stage HIP-event times wanted: every pass synchronised and timed
    run_passes(m, c, prm, bv, d_out, d_out_count, d_out_total, cap, stream, q_bounds, sub_tok_bounds, st, nullptr, false);
    return;
  }
  // No statistics: the passes are enqueued back to back and large calls from several host threads are chained on the GPU like
  // the host-buffer path chains them (Impl::turns): each call's first kernel waits for the predecessor's chain event.
  const uint32_t turns = bv.nq >= kTurnMinQueries ? m.device_turns : 0u;
  std::unique_lock<std::mutex> chain_lk(m.chain_mu, std::defer_lock);
  if (turns == 1 || turns == 2) {
    chain_lk.lock();
    if (m.chain_last) HIP_TRY(hipStreamWaitEvent(stream, m.chain_last, 0));
  }
  PassHooks hooks;
  hooks.enqueued = [&](hipEvent_t last) {
    if (chain_lk.owns_lock()) { m.chain_last = (turns == 2 && m.expand_order != 0 && !m.serial) ? c.ev_mid : last; chain_lk.unlock(); }
  };
  run_passes(m, c, prm, bv, d_out, d_out_count, d_out_total, cap, stream, q_bounds, sub_tok_bounds, nullptr, &hooks, true);
  if (chain_lk.owns_lock()) chain_lk.unlock();
}

static bool is_pinned(const void* p) {
  if (!p) return true;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

// device-visible address of a pinned host buffer (nullptr if the buffer is pageable)
template <class T> static T* pinned_device_ptr(T* p) {
  if (!p) return nullptr;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return a.type == hipMemoryTypeHost ? (T*)a.devicePointer : nullptr;
}

// Small result arrays (counts, totals: 4 bytes per query) written straight into the caller's pinned buffers over PCIe: a
// 1 MB copy through the DMA engine costs ~150 us of set-up, two of them after the last pass are half of the call's tail.
__global__ void k_export_counts(const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ tot, uint32_t* __restrict__ h_cnt,
                                uint32_t* __restrict__ h_tot, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  h_cnt[i] = cnt[i];
  if (h_tot) h_tot[i] = tot[i];
}

// penalty tokens of a text batch: slot[k] (= q_off[q] + q + j, see fmx_batch::q_itok) gets the interned id id[k]
__global__ void k_scatter_itoks(const unsigned long long* __restrict__ slot, const uint32_t* __restrict__ id, unsigned long long n,
                                unsigned long long n_slots, uint32_t* __restrict__ q_itok) {
  const unsigned long long k = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n && slot[k] < n_slots) q_itok[slot[k]] = id[k];
}

__global__ void k_rebase(uint32_t* off, uint32_t n, uint32_t base) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) off[i] -= base;
}

// Host buffers in, host buffers out.  The batch is cut into passes of host_sub_queries queries; the inputs of pass
// i + 1 travel over PCIe (own stream) while pass i computes, and the records of pass i - 1 travel back (own stream)
// at the same time.  With pinned buffers (fmx_host_alloc) the copies are truly asynchronous; pageable buffers
// still work (the runtime stages them).
void DeviceIndex::match_host(const HostIndex& h, const Params& prm, const HostBatchView& hb, Rec* out,
                             uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st, Rec* d_keep,
                             uint32_t* d_keep_count) {
  if (!d_keep && !hb.text && match_host_small(h, prm, hb, out, out_count, out_total, cap, st)) return;
  match_host_full(h, prm, hb, out, out_count, out_total, cap, st, d_keep, d_keep_count);
}

// A micro-batch in one kernel launch (fmx_small.hip.hh): inputs and results live in a pinned block of the context that the
// kernel reads and writes directly (no copy commands: one launch, one synchronisation).  Queries the kernel does not take
// (long patterns, many hits, long ranges, many survivors) go through the pipeline as a sub-batch afterwards.
// Returns false when the batch or the parameters are not eligible.
bool DeviceIndex::match_host_small(const HostIndex& h, const Params& prm, const HostBatchView& hb, Rec* out, uint32_t* out_count,
                                   uint32_t* out_total, uint32_t cap, Stats* st) {
  Impl& m = *impl_;
  const uint64_t nq = hb.nq;
  if (!m.small_path || nq == 0 || nq > kSmallMaxQueries || hb.q_real || hb.q_itok || hb.n_prior || m.d.itok || cap == 0 || cap > 4096 ||
      !small_path_params_ok(prm) || m.device < 0)
    return false;
  const uint32_t base = hb.q_off[0];
  const uint64_t nt = hb.q_off[nq] - base;
  if (nt > (uint64_t)nq * 256) return false;
  const auto t0 = std::chrono::steady_clock::now();
  HIP_TRY(hipSetDevice(m.device));
  CtxLease lease(m);
  Ctx& c = *lease.c;
  // layout of the pinned block: offsets | tokens | counts | totals | deferred | records
  const size_t n_words = (nq + 1) + nt + 3 * nq;
  const size_t bytes = ((n_words * 4 + 15) & ~(size_t)15) + nq * cap * sizeof(Rec);
  if (bytes > c.h_small_bytes) {
    if (c.h_small) (void)hipHostFree(c.h_small);
    c.h_small = nullptr; c.h_small_bytes = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 1u << 20);
    HIP_TRY(hipHostMalloc((void**)&c.h_small, want, hipHostMallocDefault));
    HIP_TRY(hipHostGetDevicePointer((void**)&c.d_small, c.h_small, 0));
    c.h_small_bytes = want;
  }
  uint32_t* hw = (uint32_t*)c.h_small;
  uint32_t* h_off = hw; uint32_t* h_tok = h_off + nq + 1; uint32_t* h_cnt = h_tok + nt; uint32_t* h_tot = h_cnt + nq; uint32_t* h_def = h_tot + nq;
  Rec* h_rec = (Rec*)(c.h_small + ((n_words * 4 + 15) & ~(size_t)15));
  for (uint64_t q = 0; q <= nq; q++) h_off[q] = hb.q_off[q] - base;
  memcpy(h_tok, hb.q_tok + base, nt * 4);
  const size_t dshift = c.d_small - c.h_small;  // device view of the same block
  auto dv = [&](void* p) { return (void*)((char*)p + dshift); };
  SmallArgs a;
  a.q_off = (const uint32_t*)dv(h_off); a.q_tok = (const uint32_t*)dv(h_tok); a.nq = (uint32_t)nq; a.cap = cap;
  const int use_sig = prm.ins >= 0.f && prm.del >= 0.f && prm.rep >= 0.f;
  a.heavy_min = (use_sig && m.d.planes && m.heavy_min) ? std::max<uint32_t>(m.heavy_min, 256u) : 0xffffffffu;
  a.out = (Rec*)dv(h_rec); a.out_count = (uint32_t*)dv(h_cnt); a.out_total = (uint32_t*)dv(h_tot); a.deferred = (uint32_t*)dv(h_def);
  hipLaunchKernelGGL(k_match_small, dim3((unsigned)nq), dim3(kWave), 0, c.stream, m.d, a, prm);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c.stream));
  std::vector<uint32_t> rest;
  for (uint64_t q = 0; q < nq; q++) {
    if (h_def[q]) { rest.push_back((uint32_t)q); continue; }
    out_count[q] = h_cnt[q];
    if (out_total) out_total[q] = h_tot[q];
    memcpy(out + q * cap, h_rec + q * cap, (size_t)h_cnt[q] * sizeof(Rec));
  }
  if (!rest.empty()) {  // the pipeline for what is left, as a batch of its own
    std::vector<uint32_t> r_off(rest.size() + 1, 0), r_tok;
    for (size_t k = 0; k < rest.size(); k++) {
      const uint32_t q = rest[k];
      r_tok.insert(r_tok.end(), hb.q_tok + hb.q_off[q], hb.q_tok + hb.q_off[q + 1]);
      r_off[k + 1] = (uint32_t)r_tok.size();
    }
    if (r_tok.empty()) r_tok.push_back(0);
    std::vector<Rec> r_out(rest.size() * cap);
    std::vector<uint32_t> r_cnt(rest.size()), r_tot(rest.size());
    HostBatchView sub;
    sub.nq = rest.size(); sub.q_off = r_off.data(); sub.q_tok = r_tok.data();
    lease.release_now();  // the pipeline leases its own context
    match_host_full(h, prm, sub, r_out.data(), r_cnt.data(), r_tot.data(), cap, nullptr, nullptr, nullptr);
    for (size_t k = 0; k < rest.size(); k++) {
      const uint32_t q = rest[k];
      out_count[q] = r_cnt[k];
      if (out_total) out_total[q] = r_tot[k];
      memcpy(out + (size_t)q * cap, r_out.data() + k * cap, (size_t)r_cnt[k] * sizeof(Rec));
    }
  }
  if (st) {
    st->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    st->n_small += nq - rest.size();
  }
  return true;
}

// this call's turn on the GPU (see Impl::turns): its first inputs are already on their way
struct HostTurn {
  DeviceIndex::Impl& m; bool held = false;
  explicit HostTurn(DeviceIndex::Impl& m_) : m(m_) {}
  void acquire() {
    if (!m.turns) return;
    std::unique_lock<std::mutex> lk(m.turn_mu);
    m.turn_cv.wait(lk, [&] { return m.turn_busy < 1; });
    m.turn_busy++; held = true;
  }
  void release() {
    if (!held) return;
    { std::lock_guard<std::mutex> lk(m.turn_mu); m.turn_busy--; }
    held = false;
    m.turn_cv.notify_one();
  }
  ~HostTurn() { release(); }
};

// One host-buffer call (fmx_match_batch / _keep / _text): begin() enqueues the copies in, the id lookup of a text batch and every
// pass; end() waits, sends the records home and checks.  fmx_match_batch* run the two back to back; fmx_match_batch_text_begin /
// _end hand them to the caller, so that ONE host thread keeps several batches in flight (each in its own context) like the
// device entry point's _begin / _end do.
struct DeviceIndex::HostCall {
  DeviceIndex::Impl& m;
  const HostIndex& h;
  const Params prm;
  const HostBatchView hb;
  Rec* const out; uint32_t* const out_count; uint32_t* const out_total; const uint32_t cap;
  Stats* const st; Rec* const d_keep; uint32_t* const d_keep_count;
  const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  uint64_t nq = 0, nt = 0;
  uint32_t base = 0;
  TableSnap text_snap;
  std::unique_ptr<CtxLease> lease;
  Ctx* cp = nullptr;
  uint64_t text_tiles = 0, text_tiles_r = 0;
  bool text_real = false;
  std::vector<uint64_t> qb;
  uint32_t n_sub = 0;
  std::vector<uint32_t> bounds;
  DeviceBatchView bv;
  bool pinned = false, export_kernel = false;
  uint32_t* hd_count = nullptr; uint32_t* hd_total = nullptr;
  double ms_h2d = 0.0, ms_d2h = 0.0;
  hipStream_t h2d = nullptr, d2h = nullptr;
  HostTurn turn;
  bool first_sent = false, ran_all = false;
  uint32_t turns = 0;
  std::unique_lock<std::mutex> chain_lk;
  PassHooks hooks;
  std::unique_ptr<PassRun> run;
  DeferredCall dc;        // begun when no context could be taken: queued on the host (Impl::deferred)
  bool finished = false;  // end() ran through

  // Nothing of this call may still run, or still read the caller's buffers, when its context goes back to the pool and the error
  // reaches the caller (who may free them): a begin() or end() that threw half-way leaves copies and kernels in flight (round-4
  // advice: match_host_abandon existed and nobody called it).  Members are destroyed after this body: the lease last.
  ~HostCall() {
    if (cp && !finished) {
      (void)hipSetDevice(m.device);
      (void)hipStreamSynchronize(cp->stream);
      (void)hipStreamSynchronize(cp->side);
      if (h2d && h2d != cp->stream) (void)hipStreamSynchronize(h2d);
      if (d2h && d2h != cp->stream) (void)hipStreamSynchronize(d2h);
      (void)hipGetLastError();
    }
  }

  HostCall(DeviceIndex::Impl& m_, const HostIndex& h_, const Params& prm_, const HostBatchView& hb_, Rec* out_, uint32_t* out_count_,
           uint32_t* out_total_, uint32_t cap_, Stats* st_, Rec* d_keep_, uint32_t* d_keep_count_)
      : m(m_), h(h_), prm(prm_), hb(hb_), out(out_), out_count(out_count_), out_total(out_total_), cap(cap_), st(st_), d_keep(d_keep_),
        d_keep_count(d_keep_count_), turn(m_), chain_lk(m_.chain_mu, std::defer_lock) {}

  // inputs of pass s: offsets [s*sub, end] (the last pass also sends q_off[nq]), tokens of those queries
  void send(uint32_t s) {
    Ctx& c = *cp;
    const auto a = std::chrono::steady_clock::now();
    const uint64_t q0 = qb[s], q1 = qb[s + 1];
    // pass s reads offsets [q0, q1] inclusive; q0 itself already came with the previous pass
    const uint64_t o_lo = s == 0 ? q0 : q0 + 1, n_off = q1 - o_lo + 1;
    HIP_TRY(hipMemcpyAsync(c.d_qoff.p + o_lo, hb.q_off + o_lo, n_off * 4, hipMemcpyHostToDevice, h2d));
    if (base) hipLaunchKernelGGL(k_rebase, dim3((unsigned)((n_off + 255) / 256)), dim3(256), 0, h2d, c.d_qoff.p + o_lo, (uint32_t)n_off, base);
    const uint64_t t0_ = bounds[s], t1_ = bounds[s + 1];
    if (hb.text) {
      if (s == 0 && hb.n_text) {  // the whole text with the first pass; the ids are looked up on the compute stream (before_pass)
        HIP_TRY(hipMemsetAsync(c.d_text.p + hb.n_text, 0, 2 * kVocabChunk + 16, h2d));
        HIP_TRY(hipMemcpyAsync(c.d_text.p, hb.text, hb.n_text, hipMemcpyHostToDevice, h2d));
        if (text_real && hb.real_text && hb.n_real_text) {
          HIP_TRY(hipMemsetAsync(c.d_text_r.p + hb.n_real_text, 0, 2 * kVocabChunk + 16, h2d));
          HIP_TRY(hipMemcpyAsync(c.d_text_r.p, hb.real_text, hb.n_real_text, hipMemcpyHostToDevice, h2d));
        }
        if (hb.n_itoks) {  // the (few) penalty tokens of the batch: (slot, interned id) pairs, scattered into the zeroed slot array
          HIP_TRY(hipMemsetAsync(c.d_qitok.p, 0, (nt + nq + 1) * 4, h2d));
          HIP_TRY(hipMemcpyAsync(c.d_islot.p, hb.itok_slot, hb.n_itoks * 8, hipMemcpyHostToDevice, h2d));
          HIP_TRY(hipMemcpyAsync(c.d_iid.p, hb.itok_id, hb.n_itoks * 4, hipMemcpyHostToDevice, h2d));
          hipLaunchKernelGGL(k_scatter_itoks, dim3((unsigned)((hb.n_itoks + 255) / 256)), dim3(256), 0, h2d, (const unsigned long long*)c.d_islot.p,
                             (const uint32_t*)c.d_iid.p, hb.n_itoks, (unsigned long long)(nt + nq), c.d_qitok.p);
        }
      }
    } else if (t1_ > t0_) {
      HIP_TRY(hipMemcpyAsync(c.d_qtok.p + t0_, hb.q_tok + base + t0_, (t1_ - t0_) * 4, hipMemcpyHostToDevice, h2d));
      if (hb.q_real) HIP_TRY(hipMemcpyAsync(c.d_qreal.p + t0_, hb.q_real + base + t0_, (t1_ - t0_) * 4, hipMemcpyHostToDevice, h2d));
    }
    if (hb.q_itok)  // slot of (query q, position j) = q_off[q] + q + j: nt + nq entries, pass s owns [t0 + q0, t1 + q1)
      HIP_TRY(hipMemcpyAsync(c.d_qitok.p + t0_ + q0, hb.q_itok + base + t0_ + q0, (t1_ - t0_ + q1 - q0) * 4, hipMemcpyHostToDevice, h2d));
    HIP_TRY(hipEventRecord(c.ev_in[s % kMaxPassesInFlight], h2d));
    ms_h2d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
  }

  void before_pass(uint32_t s) {
    Ctx& c = *cp;
    if (s == 0 && !first_sent) send(0);
    HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_in[s % kMaxPassesInFlight], 0));
    if (s == 0 && hb.text && hb.n_text) {  // spellings -> ids, for every pass of the call
      // On the call's SIDE lane (high priority), which waits for the text only: the main lane may be chained behind the call
      // before this one (Impl::turns), and the look-up — 0.3 ms alone — used to queue behind that event and then run for 2-3 ms
      // beside the predecessor's kernels before this call's first kernel could start (round 5, traced).
      const hipStream_t ls = (m.serial || !c.side) ? c.stream : c.side;
      if (ls != c.stream) HIP_TRY(hipStreamWaitEvent(ls, c.ev_in[s % kMaxPassesInFlight], 0));
      unsigned long long* d_total = nullptr;
      HIP_TRY(hipHostGetDevicePointer((void**)&d_total, c.h_vtotal, 0));
      uint32_t* tile_count = reinterpret_cast<uint32_t*>(c.d_vstate.p);
      // a text with fewer spellings than the offsets count leaves the tail of the ids untouched: VOCAB_UNK there, not whatever the
      // buffer held (the call still fails after the synchronisation, but no kernel runs on garbage)
      // (the scan fills what the look-up will not reach: no memset of the whole array)
      hipLaunchKernelGGL(k_vocab_count, dim3((unsigned)text_tiles), dim3(kScanThreads), 0, ls, (const unsigned char*)c.d_text.p,
                         (unsigned long long)hb.n_text, tile_count);
      hipLaunchKernelGGL(k_vocab_scan, dim3(1), dim3(kScanThreads), 0, ls, tile_count, (uint32_t)text_tiles, d_total, c.d_qtok.p,
                         (unsigned long long)nt, 1u);
      hipLaunchKernelGGL(k_vocab_lookup, dim3((unsigned)text_tiles), dim3(kScanThreads), 0, ls, (const unsigned char*)c.d_text.p,
                         (unsigned long long)hb.n_text, (unsigned long long)nt, text_snap.vocab, c.d_qtok.p, (const uint32_t*)tile_count, 1u);
      if (text_real) {
        constexpr uint32_t kNoReal = 0x7ffffffeu;  // HostIndex::lookup_real's "not a real token of this index"
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)c.d_qreal.p, (int)kNoReal, nt, ls));
        if (hb.real_text) {  // a second stream of spellings, one per pattern token: counted (the call fails on a mismatch) and looked up
          uint32_t* tile_count_r = reinterpret_cast<uint32_t*>(c.d_vstate_r.p);
          hipLaunchKernelGGL(k_vocab_count, dim3((unsigned)text_tiles_r), dim3(kScanThreads), 0, ls, (const unsigned char*)c.d_text_r.p,
                             (unsigned long long)hb.n_real_text, tile_count_r);
          hipLaunchKernelGGL(k_vocab_scan, dim3(1), dim3(kScanThreads), 0, ls, tile_count_r, (uint32_t)text_tiles_r, d_total + 1);
          hipLaunchKernelGGL(k_vocab_lookup, dim3((unsigned)text_tiles_r), dim3(kScanThreads), 0, ls, (const unsigned char*)c.d_text_r.p,
                             (unsigned long long)hb.n_real_text, (unsigned long long)nt, m.real_vocab, c.d_qreal.p, (const uint32_t*)tile_count_r, kNoReal);
        } else {             // the pattern tokens' own spellings, against the real-token table (same token starts, same ranks)
          hipLaunchKernelGGL(k_vocab_lookup, dim3((unsigned)text_tiles), dim3(kScanThreads), 0, ls, (const unsigned char*)c.d_text.p,
                             (unsigned long long)hb.n_text, (unsigned long long)nt, m.real_vocab, c.d_qreal.p, (const uint32_t*)tile_count, kNoReal);
        }
      }
      if (ls != c.stream) { HIP_TRY(hipEventRecord(c.ev_ids, ls)); HIP_TRY(hipStreamWaitEvent(c.stream, c.ev_ids, 0)); }
    }
    if (s + 1 < n_sub) send(s + 1);  // travels while pass s computes
  }

  void after_pass(uint32_t s, hipEvent_t done) {  // the records of pass s go home once `done` (null: already complete)
    Ctx& c = *cp;
    const auto a = std::chrono::steady_clock::now();
    const uint64_t q0 = qb[s], q1 = qb[s + 1], n = q1 - q0;
    if (done) HIP_TRY(hipStreamWaitEvent(d2h, done, 0));
    // (the records stay with the DMA engine: a copy kernel writing them over PCIe was 1 ms per million queries slower)
    HIP_TRY(hipMemcpyAsync(out + q0 * cap, c.d_out.p + q0 * cap, n * cap * sizeof(Rec), hipMemcpyDeviceToHost, d2h));
    if (export_kernel) {
      hipLaunchKernelGGL(k_export_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d2h, (const uint32_t*)c.d_cnt.p + q0,
                         (const uint32_t*)c.d_tot.p + q0, hd_count + q0, hd_total ? hd_total + q0 : nullptr, (uint32_t)n);
    } else {
      HIP_TRY(hipMemcpyAsync(out_count + q0, c.d_cnt.p + q0, n * 4, hipMemcpyDeviceToHost, d2h));
      if (out_total) HIP_TRY(hipMemcpyAsync(out_total + q0, c.d_tot.p + q0, n * 4, hipMemcpyDeviceToHost, d2h));
    }
    if (d_keep) {  // a device-resident copy for a collective that follows (the multi-process RCCL gather)
      HIP_TRY(hipMemcpyAsync(d_keep + q0 * cap, c.d_out.p + q0 * cap, n * cap * sizeof(Rec), hipMemcpyDeviceToDevice, d2h));
      if (d_keep_count) HIP_TRY(hipMemcpyAsync(d_keep_count + q0, c.d_cnt.p + q0, n * 4, hipMemcpyDeviceToDevice, d2h));
    }
    ms_d2h += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
  }

  // no_wait: NoContext instead of waiting for a context (the busy ones may all be pending calls of the calling thread; the call is
  // then deferred); force: a context beyond the pool's limit if there is none
  void begin(bool no_wait, bool force = false) {
    if (m.device < 0) throw GpuError("index is not on a device");
    HIP_TRY(hipSetDevice(m.device));
    nq = hb.nq;
    if (nq == 0) { finished = true; return; }
    base = hb.q_off[0];
    nt = hb.q_off[nq] - base;
    if (hb.text && base != 0) throw std::invalid_argument("text batch: offsets must start at 0");
    refresh_itok_table(m, h);
    if (hb.text) refresh_vocab_table(m, h);
    text_snap = pin_tables(m);  // (the pass run pins its own pair; this one keeps the spelling table of the lookup alive)
    lease.reset(new CtxLease(m, no_wait, /*host_kind=*/true, force));
    cp = lease->c;
    h2d = cp->h2d; d2h = cp->d2h;  // (what ~HostCall waits for if anything below throws; set for good further down)
    Ctx& c = *cp;
    c.d_qoff.ensure(nq + 1);
    c.d_qtok.ensure(nt + 1);
    text_tiles = hb.text ? (hb.n_text + (uint64_t)kScanThreads * kVocabChunk - 1) / ((uint64_t)kScanThreads * kVocabChunk) : 0;
    // a text batch on an index with real tokens: they are interned on the device too — from the batch's second stream of
    // spellings (what _tokenize_and_normalize made of the sentence), or, without one, from the pattern tokens' own spellings
    // (match(const Tokens&, ...): the real sentence IS the pattern, src/fuzzy_match.cc:402-416)
    text_real = hb.text && m.d.real != nullptr;
    text_tiles_r = (text_real && hb.real_text) ? (hb.n_real_text + (uint64_t)kScanThreads * kVocabChunk - 1) / ((uint64_t)kScanThreads * kVocabChunk) : 0;
    if (hb.text) {
      if (hb.q_real || hb.q_itok) throw std::invalid_argument("text batch: real tokens / penalty tokens come as spellings and (slot, id) pairs, not as id arrays");
      if (text_real && hb.real_text) { c.d_text_r.ensure(hb.n_real_text + 2 * kVocabChunk + 16); c.d_vstate_r.ensure(text_tiles_r + 4); }
      c.d_text.ensure(hb.n_text + 2 * kVocabChunk + 16);
      c.d_vstate.ensure(text_tiles + 4);
      if (!c.h_vtotal) HIP_TRY(hipHostMalloc((void**)&c.h_vtotal, 4 * sizeof(unsigned long long), hipHostMallocDefault));
      c.h_vtotal[0] = ~0ull;
      c.h_vtotal[1] = ~0ull;
    }
    if (hb.q_real || text_real) c.d_qreal.ensure(nt + 1);
    if (hb.q_itok || (hb.text && hb.n_itoks)) c.d_qitok.ensure(nt + nq + 1);
    if (hb.text && hb.n_itoks) { c.d_islot.ensure(hb.n_itoks); c.d_iid.ensure(hb.n_itoks); }
    c.d_out.ensure(nq * cap);
    c.d_cnt.ensure(nq);
    c.d_tot.ensure(nq);
    // Pass schedule: uniform passes of host_sub_queries queries (default 2^20).  Cutting a call finer lets its own copies hide
    // behind its own kernels, but two half passes cost ~12 % more GPU time than one pass (fixed launches and tails, less
    // work per kernel to hide latencies with), which is what the overlap wins for a lone caller; and as soon as two calls
    // are in flight (the reference's N worker threads, bench.py's pipeline of three) each call's copies hide behind the
    // OTHER calls' kernels anyway: measured on C3, 8.1 ms per million queries with one pass per call against 9.2 with two.
    const uint64_t per_pass = std::max<uint32_t>(1, std::min(m.sub_queries, m.host_sub_queries));
    for (uint64_t q = 0; q < nq; q += per_pass) qb.push_back(q);
    qb.push_back(nq);
    n_sub = (uint32_t)qb.size() - 1;
    bounds.assign(n_sub + 1, 0);
    uint32_t p_max = 0;
    for (uint32_t s = 0; s < n_sub; s++) bounds[s] = hb.q_off[qb[s]] - base;
    bounds[n_sub] = (uint32_t)nt;
    for (uint64_t q = 0; q < nq; q++) p_max = std::max(p_max, hb.q_off[q + 1] - hb.q_off[q]);
    bv.nq = nq; bv.q_off = c.d_qoff.p; bv.q_tok = c.d_qtok.p;
    bv.q_real = (hb.q_real || text_real) ? c.d_qreal.p : nullptr;
    bv.q_itok = (hb.q_itok || (hb.text && hb.n_itoks)) ? c.d_qitok.p : nullptr;
    bv.p_max = std::max<uint32_t>(p_max, 1);
    if (hb.n_prior) {  // the caller's prior matches: offsets then tokens in one staging buffer, ahead of the first pass
      const uint32_t n_tok = hb.prior_off[hb.n_prior];
      c.d_prior.ensure((uint64_t)hb.n_prior + 1 + n_tok + 1 + hb.n_prior);
      HIP_TRY(hipMemcpyAsync(c.d_prior.p, hb.prior_off, ((size_t)hb.n_prior + 1) * 4, hipMemcpyHostToDevice, c.stream));
      if (n_tok) HIP_TRY(hipMemcpyAsync(c.d_prior.p + hb.n_prior + 1, hb.prior_tok, (size_t)n_tok * 4, hipMemcpyHostToDevice, c.stream));
      bv.n_prior = hb.n_prior; bv.prior_off = c.d_prior.p; bv.prior_tok = c.d_prior.p + hb.n_prior + 1;
      if (hb.prior_sid) {  // the ids behind the tokens (the rerank's memo is keyed on them)
        uint32_t* d_sid = c.d_prior.p + hb.n_prior + 1 + n_tok + 1;
        HIP_TRY(hipMemcpyAsync(d_sid, hb.prior_sid, (size_t)hb.n_prior * 4, hipMemcpyHostToDevice, c.stream));
        bv.prior_sid = d_sid;
      }
      for (uint32_t r = 0; r < hb.n_prior; r++) bv.prior_max_len = std::max(bv.prior_max_len, hb.prior_off[r + 1] - hb.prior_off[r]);
    }
    pinned = is_pinned(hb.q_off) && is_pinned(hb.text ? (const void*)hb.text : (const void*)hb.q_tok) && is_pinned(out) && is_pinned(out_count);
    hd_count = pinned_device_ptr(out_count);
    hd_total = out_total ? pinned_device_ptr(out_total) : nullptr;
    export_kernel = hd_count && (!out_total || hd_total);
    // (Round 6, A/B'd and removed — profiles/r06_records_home_ab.md: the emitting kernels storing the records THROUGH the mapped pointer of
    // a pinned `out` made the replay / rerank kernels wait for PCIe — value 186 -> 165 M patterns/s at C3; and the copies home cost less
    // than it looked: with NO copy at all a step takes 5.06 instead of 5.34 ms.)
    // Small single-pass calls put their copies on the compute stream: in order, no event between streams.  (With eight hardware
    // queues every stream has a queue of its own and a dependency between two of them is a signal round trip; a 100 k-query call
    // has four of them around ~1 ms of kernels, and its copies are too short to be worth overlapping with its own kernels —
    // other calls in flight fill the gaps.)
    static const uint64_t inline_max = [] { const char* e = getenv("FMX_HOST_INLINE_COPY"); return e ? (uint64_t)atoll(e) : (uint64_t)262144; }();
    const bool inline_copies = n_sub == 1 && nq <= inline_max && !m.serial;
    h2d = inline_copies ? c.stream : c.h2d;
    d2h = inline_copies ? c.stream : c.d2h;
    // (small calls — the coalescer's micro-batches, interactive callers, shards of a strongly scaled batch — do not fill the GPU
    // for long and simply overlap: measured at 100 k and 125 k queries per call, 103 M patterns/s against 95 M chained)
    static const int spacing_us = [] { const char* e = getenv("FMX_HOST_SPACING_US"); return e ? atoi(e) : 500; }();
    if (nq >= kTurnMinQueriesHost) stream_pool(m.device).pace(std::chrono::microseconds(spacing_us));
    turns = nq >= kTurnMinQueriesHost ? m.turns : 0u;
    if (no_wait && turns >= 3) turns = 0;  // (turns taken on the host would block the one thread that holds the other pending calls)
    if (turns == 1 || turns == 2) {
      // chained on the GPU: calls enqueue in the order they take the lock, each behind the end of its predecessor's kernels
      // (turns = 2: behind the predecessor's memory-bound half — its probe and long-range kernels — so that this call's probe,
      // bound by HBM, runs beside the predecessor's hit walk / edit distances / replay, bound by instruction issue)
      send(0); first_sent = true;
      chain_lk.lock();
      if (m.chain_last) HIP_TRY(hipStreamWaitEvent(c.stream, m.chain_last, 0));
    } else if (turns >= 3) {
      send(0); first_sent = true; turn.acquire();
    }
    hooks.before = [this](uint32_t s) { before_pass(s); };
    hooks.after = [this](uint32_t s, hipEvent_t done) { after_pass(s, done); };
    hooks.enqueued = [this](hipEvent_t last) {
      if (chain_lk.owns_lock()) { m.chain_last = (turns == 2 && m.expand_order != 0 && !m.serial) ? cp->ev_mid : last; chain_lk.unlock(); }
    };
    hooks.computed = [this]() { turn.release(); };
    // stage timing (st with events) is the device entry point's job: here the passes run back to back, unsynchronised
    // (the chain lock is given back on THIS thread whatever happens below: a deferred call is begun by whichever thread frees a slot,
    // and a std::mutex must be unlocked by the thread that locked it)
    try {
      run.reset(new PassRun(m, c, prm, bv, c.d_out.p, c.d_cnt.p, c.d_tot.p, cap, c.stream, qb, bounds, st, &hooks));
      run->prepare_async();
      if (n_sub <= kAsyncGroup) run->begin_group(0, n_sub);
      else { run->run_all_async(); ran_all = true; }  // more passes than one group holds: the call runs to its end here
    } catch (...) {
      if (chain_lk.owns_lock()) chain_lk.unlock();
      throw;
    }
    if (chain_lk.owns_lock()) chain_lk.unlock();  // (hooks.enqueued released it unless no pass was enqueued)
  }

  void end() {
    if (nq == 0) return;
    Ctx& c = *cp;
    HIP_TRY(hipSetDevice(m.device));
    if (!ran_all) run->end_group(0, n_sub);
    if (chain_lk.owns_lock()) chain_lk.unlock();
    turn.release();
    const auto a = std::chrono::steady_clock::now();
    HIP_TRY(hipEventRecord(c.ev_done, d2h));
    HIP_TRY(hipEventSynchronize(c.ev_done));
    if (hb.text && (hb.n_text ? *c.h_vtotal : 0ull) != nt)
      throw std::invalid_argument("text batch: the text holds " + std::to_string(*c.h_vtotal) + " spellings, the offsets count " + std::to_string(nt));
    if (text_real && hb.real_text && nt && c.h_vtotal[1] != nt)
      throw std::invalid_argument("text batch: the real-token text holds " + std::to_string(c.h_vtotal[1]) + " spellings, the offsets count " + std::to_string(nt));
    ms_d2h += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
    if (st) {
      st->ms_h2d += ms_h2d;   // host time spent enqueuing / staging; with pinned buffers the copies overlap the passes
      st->ms_d2h += ms_d2h;
      st->pinned = pinned ? 1 : 0;
      st->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    finished = true;
  }
};

void DeviceIndex::match_host_full(const HostIndex& h, const Params& prm, const HostBatchView& hb, Rec* out,
                                  uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st, Rec* d_keep,
                                  uint32_t* d_keep_count) {
  HostCall call(*impl_, h, prm, hb, out, out_count, out_total, cap, st, d_keep, d_keep_count);
  call.begin(false);
  call.end();
}

// The two halves of a host-buffer call for callers that keep several in flight from ONE thread (fmx_match_batch_text_begin / _end).
// The caller's buffers must stay valid until _end; _end on the thread that began.
DeviceIndex::HostCall* DeviceIndex::match_host_begin(const HostIndex& h, const Params& prm, const HostBatchView& hb, Rec* out, uint32_t* out_count,
                                                     uint32_t* out_total, uint32_t cap, Rec* d_keep, uint32_t* d_keep_count) {
  std::unique_ptr<HostCall> call(new HostCall(*impl_, h, prm, hb, out, out_count, out_total, cap, nullptr, d_keep, d_keep_count));
  HostCall* cp = call.get();
  call->dc.host_kind = true;
  call->dc.start = [cp](bool force) { cp->begin(true, force); };
  begin_or_defer(*impl_, call->dc);  // beyond the calls that may run at once: queued on the host, started when a slot is freed
  return call.release();
}

void DeviceIndex::match_host_end(HostCall* call_) {
  std::unique_ptr<HostCall> call(call_);
  if (!call) return;
  ensure_started(*impl_, call->dc);
  call->end();
}

void DeviceIndex::match_host_abandon(HostCall* call_) {
  std::unique_ptr<HostCall> call(call_);
  if (!call) return;
  try { ensure_started(*impl_, call->dc); } catch (...) {}
  // (~HostCall waits for the call's streams before the context goes back to the pool)
}

__global__ void k_pmax(const uint32_t* q_off, unsigned long long nq, uint32_t* out) {
  const unsigned long long q = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t p = q < nq ? q_off[q + 1] - q_off[q] : 0u;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) p = max(p, (uint32_t)__shfl_xor(p, d, 64));
  if ((threadIdx.x & 63) == 0 && p) atomicMax(out, p);
}

// ---------------------------------------------------------------------------------------------
// Pre-tokenised lines (fmx_match_batch_lines): the bytes of a file go to the device as they are; it cuts them into patterns
// and tokens (fmx_vocab.hip.hh: k_lines_count / k_lines_emit), looks the ids up (k_vocab_lookup) and runs the pipeline on the
// device-resident CSR.  The host reads back two numbers (tokens found, longest pattern) before the passes are enqueued.
// ---------------------------------------------------------------------------------------------
void DeviceIndex::match_lines(const HostIndex& h, const Params& prm, const char* text, uint64_t n_bytes, uint64_t n_lines, Rec* out,
                              uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st) {
  match_lines_impl(h, prm, text, n_bytes, n_lines, out, out_count, out_total, cap, st, nullptr, nullptr);
}

// raw text: the byte-class tokenizer on the device (fmx_vocab.hip.hh: k_raw_count / k_raw_emit); bad_lines = the lines it flagged
// (bytes outside its alphabet: the caller tokenises them on the host and overwrites their rows)
void DeviceIndex::match_raw_lines(const HostIndex& h, const Params& prm, const char* text, uint64_t n_bytes, uint64_t n_lines, Rec* out,
                                  uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st, const RawLineTok& tok,
                                  std::vector<uint64_t>& bad_lines) {
  match_lines_impl(h, prm, text, n_bytes, n_lines, out, out_count, out_total, cap, st, &tok, &bad_lines);
}

void DeviceIndex::match_lines_impl(const HostIndex& h, const Params& prm, const char* text, uint64_t n_bytes, uint64_t n_lines, Rec* out,
                                   uint32_t* out_count, uint32_t* out_total, uint32_t cap, Stats* st, const RawLineTok* raw,
                                   std::vector<uint64_t>* bad_lines) {
  Impl& m = *impl_;
  if (m.device < 0) throw GpuError("index is not on a device");
  HIP_TRY(hipSetDevice(m.device));
  const auto t0 = std::chrono::steady_clock::now();
  if (n_lines == 0) return;
  if (n_bytes >= 0xfffffff0ull) throw std::invalid_argument("line batch: more than 4 GB of text in one call");
  refresh_itok_table(m, h);
  refresh_vocab_table(m, h);
  const TableSnap text_snap = pin_tables(m);
  CtxLease lease(m, /*no_wait=*/false, /*host_kind=*/true);  // (a host-buffer call like any other: it counts against max_host_active)
  Ctx& c = *lease.c;
  const uint64_t tiles = (n_bytes + (uint64_t)kScanThreads * kVocabChunk - 1) / ((uint64_t)kScanThreads * kVocabChunk);
  // a token is at least one byte and its separator; raw text: every byte may be a token of its own (punctuation)
  const uint64_t tok_cap = raw ? n_bytes + 2 : n_bytes / 2 + 2;
  if (raw) c.d_lflag.ensure(n_lines + 8);
  c.d_text.ensure(n_bytes + 2 * kVocabChunk + 16);
  c.d_vstate.ensure(tiles + 4);
  c.d_vstate_r.ensure(tiles + 4);
  c.d_qoff.ensure(n_lines + 1);
  c.d_qtok.ensure(tok_cap);
  const bool real = m.d.real != nullptr;
  if (real) c.d_qreal.ensure(tok_cap);
  c.d_out.ensure(n_lines * cap);
  c.d_cnt.ensure(n_lines);
  c.d_tot.ensure(n_lines);
  if (!c.h_vtotal) HIP_TRY(hipHostMalloc((void**)&c.h_vtotal, 4 * sizeof(unsigned long long), hipHostMallocDefault));
  c.h_vtotal[0] = c.h_vtotal[1] = ~0ull;
  c.h_vtotal[2] = 0ull;
  c.h_vtotal[3] = 0ull;
  c.d_scratch.ensure(16);  // (a context-owned scratch word: no allocation, and no hipFree — a device-wide wait — inside a call)
  unsigned long long* d_total = nullptr;
  HIP_TRY(hipHostGetDevicePointer((void**)&d_total, c.h_vtotal, 0));
  hipStream_t s = c.stream;
  HIP_TRY(hipMemsetAsync(c.d_text.p + n_bytes, 0, 2 * kVocabChunk + 16, s));
  HIP_TRY(hipMemcpyAsync(c.d_text.p, text, n_bytes, hipMemcpyHostToDevice, s));
  uint32_t* tile_tok = reinterpret_cast<uint32_t*>(c.d_vstate.p);
  uint32_t* tile_nl = reinterpret_cast<uint32_t*>(c.d_vstate_r.p);
  if (raw) {
    RawTok rt;
    rt.cas = raw->cas; rt.nbr = raw->nbr; rt.num_id = raw->num_id;
    for (int k = 0; k < 4; k++) rt.feat_real[k] = raw->feat_real[k];
    HIP_TRY(hipMemsetAsync(c.d_lflag.p, 0, n_lines + 8, s));
    hipLaunchKernelGGL(k_raw_count, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, (const unsigned char*)c.d_text.p, (unsigned long long)n_bytes, tile_tok, tile_nl);
    hipLaunchKernelGGL(k_vocab_scan, dim3(1), dim3(kScanThreads), 0, s, tile_tok, (uint32_t)tiles, d_total);
    hipLaunchKernelGGL(k_vocab_scan, dim3(1), dim3(kScanThreads), 0, s, tile_nl, (uint32_t)tiles, d_total + 1);
    hipLaunchKernelGGL(k_raw_emit, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, (const unsigned char*)c.d_text.p, (unsigned long long)n_bytes,
                       (const uint32_t*)tile_tok, (const uint32_t*)tile_nl, c.d_qoff.p, (unsigned long long)n_lines, (unsigned long long)tok_cap,
                       text_snap.vocab, m.real_vocab, rt, c.d_qtok.p, real ? c.d_qreal.p : (uint32_t*)nullptr, c.d_lflag.p, d_total + 3);
  } else {
  hipLaunchKernelGGL(k_lines_count, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, (const unsigned char*)c.d_text.p, (unsigned long long)n_bytes, tile_tok, tile_nl);
  hipLaunchKernelGGL(k_vocab_scan, dim3(1), dim3(kScanThreads), 0, s, tile_tok, (uint32_t)tiles, d_total);
  hipLaunchKernelGGL(k_vocab_scan, dim3(1), dim3(kScanThreads), 0, s, tile_nl, (uint32_t)tiles, d_total + 1);
  hipLaunchKernelGGL(k_lines_emit, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, c.d_text.p, (unsigned long long)n_bytes, (const uint32_t*)tile_tok,
                     (const uint32_t*)tile_nl, c.d_qoff.p, (unsigned long long)n_lines);
  hipLaunchKernelGGL(k_vocab_lookup, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, (const unsigned char*)c.d_text.p, (unsigned long long)n_bytes,
                     (unsigned long long)tok_cap, text_snap.vocab, c.d_qtok.p, (const uint32_t*)tile_tok, 1u);
  if (real)  // match(const Tokens&, ...): the pattern's own spellings are its real tokens (src/fuzzy_match.cc:402-416)
    hipLaunchKernelGGL(k_vocab_lookup, dim3((unsigned)tiles), dim3(kScanThreads), 0, s, (const unsigned char*)c.d_text.p, (unsigned long long)n_bytes,
                       (unsigned long long)tok_cap, m.real_vocab, c.d_qreal.p, (const uint32_t*)tile_tok, 0x7ffffffeu);
  }
  HIP_TRY(hipGetLastError());
  // the passes need the token bounds and the longest pattern: read back with ONE wait (the token total is the end of the only
  // pass; a call of more lines than a pass holds reads the boundaries of its passes too)
  DeviceBatchView bv;
  bv.nq = n_lines; bv.q_off = c.d_qoff.p; bv.q_tok = c.d_qtok.p; bv.q_real = real ? c.d_qreal.p : nullptr;
  uint32_t p_max = 0;
  std::vector<uint32_t> bounds;
  if (n_lines <= m.sub_queries) {
    HIP_TRY(hipMemsetAsync(c.d_scratch.p, 0, 4, s));
    hipLaunchKernelGGL(k_pmax, dim3((unsigned)((n_lines + 255) / 256)), dim3(256), 0, s, (const uint32_t*)c.d_qoff.p, (unsigned long long)n_lines, c.d_scratch.p);
    HIP_TRY(hipMemcpyAsync(&c.h_vtotal[2], c.d_scratch.p, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    p_max = (uint32_t)c.h_vtotal[2];
    if (c.h_vtotal[0] > 0xffffffffull) throw std::invalid_argument("line batch: more than 2^32 - 1 tokens");
    bounds = {0u, (uint32_t)c.h_vtotal[0]};
  } else {
    bounds = sub_bounds_from_device(c.d_qoff.p, n_lines, &p_max, s);
  }
  if (c.h_vtotal[1] != n_lines)
    throw std::invalid_argument("line batch: the text holds " + std::to_string(c.h_vtotal[1]) + " line ends, the call names " + std::to_string(n_lines) + " lines");
  if (raw && bad_lines) {  // the lines the device tokenizer does not cover (their flags travel while the passes are enqueued)
    bad_lines->clear();
    if (c.h_vtotal[3] != 0) {
      std::vector<unsigned char> flags(n_lines);
      HIP_TRY(hipMemcpyAsync(flags.data(), c.d_lflag.p, n_lines, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      for (uint64_t l = 0; l < n_lines; l++) if (flags[l]) bad_lines->push_back(l);
    }
  }
  bv.p_max = std::max<uint32_t>(p_max, 1);
  std::vector<uint64_t> qb;
  for (uint64_t q = 0; q < n_lines; q += m.sub_queries) qb.push_back(q);
  qb.push_back(n_lines);
  uint32_t* const hd_count = pinned_device_ptr(out_count);
  uint32_t* const hd_total = out_total ? pinned_device_ptr(out_total) : nullptr;
  const bool export_kernel = hd_count && (!out_total || hd_total);
  PassHooks hooks;
  hooks.after = [&](uint32_t sb, hipEvent_t) {  // the records of a finished pass go home (same stream: in order behind its kernels)
    const uint64_t q0 = qb[sb], q1 = qb[sb + 1], n = q1 - q0;
    HIP_TRY(hipMemcpyAsync(out + q0 * cap, c.d_out.p + q0 * cap, n * cap * sizeof(Rec), hipMemcpyDeviceToHost, s));
    if (export_kernel) {
      hipLaunchKernelGGL(k_export_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const uint32_t*)c.d_cnt.p + q0,
                         (const uint32_t*)c.d_tot.p + q0, hd_count + q0, hd_total ? hd_total + q0 : nullptr, (uint32_t)n);
    } else {
      HIP_TRY(hipMemcpyAsync(out_count + q0, c.d_cnt.p + q0, n * 4, hipMemcpyDeviceToHost, s));
      if (out_total) HIP_TRY(hipMemcpyAsync(out_total + q0, c.d_tot.p + q0, n * 4, hipMemcpyDeviceToHost, s));
    }
  };
  run_passes(m, c, prm, bv, c.d_out.p, c.d_cnt.p, c.d_tot.p, cap, s, qb, bounds, st, &hooks, true);
  HIP_TRY(hipStreamSynchronize(s));
  if (st) st->ms_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

// ---------------------------------------------------------------------------------------------
// FuzzyMatch::subsequence (src/fuzzy_match.cc:250-365) for a batch of tokenised patterns.  The priority queue over the
// sub-n-grams is host prep (std::priority_queue<Subseq> = std::push_heap / pop_heap; libstdc++'s algorithm is restated
// so that equal keys pop in the reference's order); ranges, edit distances and the acceptance run on the device.
// ---------------------------------------------------------------------------------------------
namespace {
struct Subseq { float weight; uint32_t position, length; };
inline bool subseq_less(const Subseq& a, const Subseq& b) {  // src/fuzzy_match.cc:240-243
  return a.weight < b.weight || (a.weight == b.weight && a.position > b.position);
}
void sq_push_heap(Subseq* first, int64_t hole, int64_t top, Subseq value) {  // bits/stl_heap.h __push_heap
  int64_t parent = (hole - 1) / 2;
  while (hole > top && subseq_less(first[parent], value)) { first[hole] = first[parent]; hole = parent; parent = (hole - 1) / 2; }
  first[hole] = value;
}
void sq_adjust_heap(Subseq* first, int64_t hole, int64_t len, Subseq value) {  // bits/stl_heap.h __adjust_heap
  const int64_t top = hole;
  int64_t second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (subseq_less(first[second], first[second - 1])) second--;
    first[hole] = first[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) { second = 2 * (second + 1); first[hole] = first[second - 1]; hole = second - 1; }
  sq_push_heap(first, hole, top, value);
}
}  // namespace

void DeviceIndex::subsequence_host(const HostIndex& h, const HostBatchView& hb, uint32_t nm, bool no_perfect, int ml_in, float mr,
                                   bool idf_weighting, Rec* out, uint32_t* out_found, uint32_t* out_pos) {
  Impl& m = *impl_;
  if (m.device < 0) throw GpuError("index is not on a device");
  HIP_TRY(hipSetDevice(m.device));
  const uint64_t nq_all = hb.nq;
  for (uint64_t q = 0; q < nq_all; q++) { out_found[q] = 0; if (out_pos) out_pos[q] = 0; out[q] = Rec{0.f, 0.f, 0, 0}; }
  if (nq_all == 0 || nm == 0) return;  // candidates.size() < 0 never holds: nothing is ever scored (:312)
  refresh_itok_table(m, h);
  const TableSnap snap = pin_tables(m);
  const DevIndex D = dev_view(m, snap);
  CtxLease lease(m);
  Ctx& c = *lease.c;
  hipStream_t stream = c.stream;
  // per query: the sentences already seen = candidates (<= nm) + perfect matches skipped under no_perfect.  The list starts
  // small; a chunk of queries in which one overflows is run again with a list eight times as long (the reference has no limit;
  // at most every sentence of the index can be skipped)
  uint32_t seen_cap = nm + 1024;
  const uint64_t seen_max = (uint64_t)nm + h.num_sentences() + 1;
  constexpr uint64_t kChunkMax = 16384;    // queries per round trip: bounds the host-side queues
  uint64_t kChunk = kChunkMax;
  for (uint64_t q0 = 0; q0 < nq_all;) {
    // (a long seen list shrinks the chunk: the lists are nq x seen_cap words of HBM)
    while (kChunk > 1 && kChunk * (uint64_t)seen_cap > (1ull << 28)) kChunk /= 2;
    const uint64_t nq = std::min<uint64_t>(kChunk, nq_all - q0);
    const uint32_t base = hb.q_off[q0];
    const uint64_t nt = hb.q_off[q0 + nq] - base;
    std::vector<uint32_t> off(nq + 1);
    uint32_t p_max = 1;
    for (uint64_t q = 0; q <= nq; q++) off[q] = hb.q_off[q0 + q] - base;
    for (uint64_t q = 0; q < nq; q++) p_max = std::max(p_max, off[q + 1] - off[q]);
    const size_t lds = ((size_t)p_max + 1) * kWave * 4;
    const bool rows_in_hbm = lds > kDynLdsMax;  // patterns of more than ~590 tokens: the same kernel over rows in HBM
    const unsigned sub_grid_g = 1024;
    if (rows_in_hbm) c.g_rows.ensure((uint64_t)sub_grid_g * kWave * (p_max + 1));
    else HIP_TRY(hipFuncSetAttribute((const void*)k_sub_eval<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kDynLdsMax));
    c.d_qoff.ensure(nq + 1); c.d_qtok.ensure(nt + 1); c.t_q.ensure(nt + 1);
    HIP_TRY(hipMemcpyAsync(c.d_qoff.p, off.data(), (nq + 1) * 4, hipMemcpyHostToDevice, stream));
    if (nt) HIP_TRY(hipMemcpyAsync(c.d_qtok.p, hb.q_tok + base, nt * 4, hipMemcpyHostToDevice, stream));
    const uint32_t* d_real = nullptr;
    const uint32_t* d_itok = nullptr;
    if (hb.q_real) { c.d_qreal.ensure(nt + 1); if (nt) HIP_TRY(hipMemcpyAsync(c.d_qreal.p, hb.q_real + base, nt * 4, hipMemcpyHostToDevice, stream)); d_real = c.d_qreal.p; }
    if (hb.q_itok) { c.d_qitok.ensure(nt + nq + 1); HIP_TRY(hipMemcpyAsync(c.d_qitok.p, hb.q_itok + base + q0, (nt + nq) * 4, hipMemcpyHostToDevice, stream)); d_itok = c.d_qitok.p; }
    Grow<uint16_t> d_len; Grow<SubItem> d_items; Grow<SubState> d_st; Grow<uint32_t> d_seen;
    struct Free { Grow<uint16_t>& a; Grow<SubItem>& b; Grow<SubState>& c_; Grow<uint32_t>& d; ~Free() { a.release(); b.release(); c_.release(); d.release(); } } guard{d_len, d_items, d_st, d_seen};
    d_len.ensure(nt + 1); d_items.ensure(nq); d_st.ensure(nq); d_seen.ensure(nq * (uint64_t)seen_cap);
    std::vector<uint16_t> t_len(nt);
    if (nt) {
      hipLaunchKernelGGL(k_sub_tq, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, c.d_qoff.p, (uint32_t)nq, c.t_q.p);
      hipLaunchKernelGGL(k_sub_extent, dim3((unsigned)((nt + 63) / 64)), dim3(64), 0, stream, D, c.d_qoff.p, c.d_qtok.p, c.t_q.p, (uint32_t)nt, d_len.p);
      HIP_TRY(hipMemcpyAsync(t_len.data(), d_len.p, nt * 2, hipMemcpyDeviceToHost, stream));
    }
    std::vector<SubState> st(nq);
    for (auto& s : st) { s = SubState{}; s.md = 10000; }
    HIP_TRY(hipMemcpyAsync(d_st.p, st.data(), nq * sizeof(SubState), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    // the queues (src/fuzzy_match.cc:279-290)
    std::vector<std::vector<Subseq>> heaps(nq);
    for (uint64_t q = 0; q < nq; q++) {
      const uint32_t p = off[q + 1] - off[q];
      int ml = ml_in;
      if ((int)(mr * p) > ml) ml = (int)(mr * p);  // :267-268
      if ((int)p < ml) continue;                    // :270-271
      const uint32_t* w = hb.q_tok + base + off[q];
      auto& hp = heaps[q];
      for (uint32_t it = 0; it < p; it++) {
        float idf_weight = 0;
        for (uint32_t jt = it; jt < p; jt++) {
          const uint32_t wid = w[jt];
          if (wid == 1u || wid >= h.idf.size()) break;  // weight -1: ｟unk｠ (:283-284)
          // compute_idf_penalty (:372-390): HostIndex::idf holds logf(N / sfreq) per word
          idf_weight += idf_weighting ? h.idf[wid] : 1;
          if ((int)(jt - it + 1) >= ml) {
            const Subseq v{idf_weight, it, jt - it + 1};
            hp.push_back(v);
            sq_push_heap(hp.data(), (int64_t)hp.size() - 1, 0, v);
          }
        }
      }
    }
    std::vector<SubItem> items;
    for (;;) {
      items.clear();
      for (uint64_t q = 0; q < nq; q++) {
        auto& hp = heaps[q];
        if (st[q].md != 10000 || st[q].overflow) { hp.clear(); continue; }  // while (... && max_distance == 10000) (:303-304)
        while (!hp.empty()) {
          const Subseq top = hp[0];
          const Subseq last = hp.back();
          hp.pop_back();
          if (!hp.empty()) sq_adjust_heap(hp.data(), 0, (int64_t)hp.size(), last);
          // an empty range does nothing in the reference's loop; ranges are non-empty exactly up to the extent
          if (top.length <= t_len[off[q] + top.position]) { items.push_back(SubItem{(uint32_t)q, top.position, top.length}); break; }
        }
      }
      if (items.empty()) break;
      HIP_TRY(hipMemcpyAsync(d_items.p, items.data(), items.size() * sizeof(SubItem), hipMemcpyHostToDevice, stream));
      if (rows_in_hbm)
        hipLaunchKernelGGL(k_sub_eval<true>, dim3((unsigned)std::min<size_t>(items.size(), sub_grid_g)), dim3(kWave), 0, stream, D, c.d_qoff.p, c.d_qtok.p,
                           d_real, d_itok, (const SubItem*)d_items.p, (uint32_t)items.size(), d_st.p, d_seen.p, seen_cap, nm, no_perfect ? 1 : 0,
                           c.g_rows.p, p_max + 1);
      else
        hipLaunchKernelGGL(k_sub_eval<false>, dim3((unsigned)std::min<size_t>(items.size(), 65535)), dim3(kWave), lds, stream, D, c.d_qoff.p, c.d_qtok.p,
                           d_real, d_itok, (const SubItem*)d_items.p, (uint32_t)items.size(), d_st.p, d_seen.p, seen_cap, nm, no_perfect ? 1 : 0,
                           (float*)nullptr, 0u);
      HIP_TRY(hipGetLastError());
      HIP_TRY(hipMemcpyAsync(st.data(), d_st.p, nq * sizeof(SubState), hipMemcpyDeviceToHost, stream));
      HIP_TRY(hipStreamSynchronize(stream));
    }
    bool overflow = false;
    for (uint64_t q = 0; q < nq; q++) overflow = overflow || st[q].overflow;
    if (overflow) {  // run this chunk again with longer lists
      if (seen_cap >= seen_max) throw GpuError("subsequence(): the seen list overflowed at its largest size");
      seen_cap = (uint32_t)std::min<uint64_t>((uint64_t)seen_cap * 8, std::min<uint64_t>(seen_max, 0xffffffffull));
      continue;
    }
    for (uint64_t q = 0; q < nq; q++)
      if (st[q].md != 10000) { out[q0 + q] = st[q].best; out_found[q0 + q] = 1; if (out_pos) out_pos[q0 + q] = st[q].pos; }
    q0 += nq;
  }
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
  uint32_t* d_pm = nullptr;
  if (p_max && *p_max == 0 && nq) {
    {  // a scratch word from the index's pool (no allocation and no hipFree — a device-wide wait — per call)
      std::lock_guard<std::mutex> g(m.word_mu);
      if (!m.word_pool.empty()) { d_pm = m.word_pool.back(); m.word_pool.pop_back(); }
    }
    if (!d_pm) d_pm = dev_alloc<uint32_t>(1);
    HIP_TRY(hipMemsetAsync(d_pm, 0, 4, stream));
    hipLaunchKernelGGL(k_pmax, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, d_q_off, nq, d_pm);
    HIP_TRY(hipMemcpyAsync(p_max, d_pm, 4, hipMemcpyDeviceToHost, stream));
  }
  HIP_TRY(hipStreamSynchronize(stream));
  if (d_pm) { std::lock_guard<std::mutex> g(m.word_mu); m.word_pool.push_back(d_pm); }
  return bounds;
}

// the suffix sorter of the index build (gpu_suffix_sort): fmx_sort.hip

}  // namespace fmx

#ifdef FMX_LIGHT_STATS
extern "C" int fmx_debug_light_stats(unsigned long long* out, int reset) {
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(fmx::g_light_stats), sizeof(fmx::g_light_stats)) != hipSuccess) return -1;
  if (reset) { static unsigned long long z[1024]; if (hipMemcpyToSymbol(HIP_SYMBOL(fmx::g_light_stats), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

