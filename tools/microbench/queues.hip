// queues.hip — stand-alone repro of the "calls in flight" cliff (review item: seven host-buffer calls in flight take 107 ms per step,
// eight 600-900 ms, and it is not the number of compute queues).  No library code: N "calls" in flight from ONE host thread, each
//   H2D copy of its input (pinned -> HBM, shared copy stream)  ->  a chain of L dummy kernels on the call's own stream pair with an
//   event fork / join every other kernel (the pass's main + side lanes)  ->  D2H copy of its records (HBM -> pinned, shared stream)
// — the shape of fmx_match_batch_text_begin / _end.  The kernels spin for a fixed time on every CU, so a step costs the same GPU time
// whatever N is: any growth of ms / step with N is the runtime's / the scheduler's.
//   hipcc --offload-arch=gfx950 -O3 -o queues queues.hip && ./queues [mode ...]
// modes (any subset, default "all"): base | no_d2h | no_h2d | no_copies | one_stream | own_copy_streams | chained | small_copies |
//   prio_lane | memset_ctr | library_like (= chained + prio_lane + memset_ctr) | prio_shared | library_like_prio_shared
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k_spin(unsigned long long ticks, uint32_t* sink) {
  const unsigned long long t0 = wall_clock64();
  uint32_t x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = x * 1664525u + 1013904223u;
  if (x == 0x12345u && sink) sink[0] = x;
}

struct Mode {
  const char* name;
  bool h2d, d2h, two_streams, own_copy_streams, chained;
  size_t in_bytes, out_bytes;
  bool prio_lane = false;   // the id look-up of a text call: a short kernel on a HIGH-PRIORITY stream of the call, gated by its input copy
  bool memset_ctr = false;  // per pass: a memset of the counters and a 4 KB copy of them home on the compute stream
  bool prio_shared = false; // prio_lane with ONE high-priority stream for all calls instead of one per call
};

struct Call {
  hipStream_t main = nullptr, side = nullptr, h2d = nullptr, d2h = nullptr, prio = nullptr;
  hipEvent_t e_in, e_fork, e_join, e_k, e_mid, e_done, e_ids;
  char* h_ctr = nullptr;
  char *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
};

static double run(const Mode& md, int n_calls, int steps, int chain_len, unsigned long long ticks) {
  std::vector<Call> c(n_calls);
  hipStream_t sh = nullptr, sd = nullptr;
  if (getenv("QUEUES_COPY_LOW") && atoi(getenv("QUEUES_COPY_LOW"))) {
    int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithPriority(&sh, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithPriority(&sd, hipStreamNonBlocking, lo));
  } else {
    CK(hipStreamCreateWithFlags(&sh, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sd, hipStreamNonBlocking));
  }
  hipStream_t sp = nullptr;
  const int prio_k = getenv("QUEUES_PRIO_K") ? atoi(getenv("QUEUES_PRIO_K")) : 1 << 30;   // only the first K calls get a high-priority lane (the others: normal)
  const bool copy_low = getenv("QUEUES_COPY_LOW") && atoi(getenv("QUEUES_COPY_LOW"));     // the shared copy streams in the LOW priority class (a third pool of queues)
  int n_made = 0;
  if (md.prio_lane && md.prio_shared) { int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); CK(hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi)); }
  for (auto& k : c) {
    CK(hipStreamCreateWithFlags(&k.main, hipStreamNonBlocking));
    if (md.two_streams) CK(hipStreamCreateWithFlags(&k.side, hipStreamNonBlocking));
    if (md.own_copy_streams) { CK(hipStreamCreateWithFlags(&k.h2d, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&k.d2h, hipStreamNonBlocking)); }
    else { k.h2d = sh; k.d2h = sd; }
    if (md.prio_lane && !md.prio_shared) {
      int lo = 0, hi = 0; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
      if (n_made++ < prio_k) CK(hipStreamCreateWithPriority(&k.prio, hipStreamNonBlocking, hi));
      else CK(hipStreamCreateWithFlags(&k.prio, hipStreamNonBlocking));
    }
    if (md.prio_lane && md.prio_shared) k.prio = sp;
    for (hipEvent_t* e : {&k.e_in, &k.e_fork, &k.e_join, &k.e_k, &k.e_mid, &k.e_done, &k.e_ids}) CK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    CK(hipHostMalloc((void**)&k.h_ctr, 4096, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&k.h_in, md.in_bytes, hipHostMallocDefault));
    CK(hipHostMalloc((void**)&k.h_out, md.out_bytes, hipHostMallocDefault));
    CK(hipMalloc((void**)&k.d_in, md.in_bytes));
    CK(hipMalloc((void**)&k.d_out, md.out_bytes));
    memset(k.h_in, 1, md.in_bytes);
  }
  hipEvent_t chain_last = nullptr;
  auto begin = [&](int i) {
    Call& k = c[i % n_calls];
    if (md.h2d) {
      CK(hipMemcpyAsync(k.d_in, k.h_in, md.in_bytes, hipMemcpyHostToDevice, k.h2d));
      CK(hipEventRecord(k.e_in, k.h2d));
      CK(hipStreamWaitEvent(k.main, k.e_in, 0));
    }
    if (md.prio_lane) {  // the look-up: behind the input copy only, on the high-priority lane; the pass waits for it
      if (md.h2d) CK(hipStreamWaitEvent(k.prio, k.e_in, 0));
      for (int u = 0; u < 3; u++) hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, k.prio, ticks / 2, (uint32_t*)nullptr);
      CK(hipEventRecord(k.e_ids, k.prio));
      CK(hipStreamWaitEvent(k.main, k.e_ids, 0));
    }
    if (md.memset_ctr) CK(hipMemsetAsync(k.d_out, 0, 1 << 20, k.main));
    if (md.chained && chain_last) CK(hipStreamWaitEvent(k.main, chain_last, 0));
    for (int l = 0; l < chain_len; l++) {
      if (md.two_streams && (l & 1)) {  // fork: a kernel on the side lane beside one on the main lane, then join
        CK(hipEventRecord(k.e_fork, k.main));
        CK(hipStreamWaitEvent(k.side, k.e_fork, 0));
        hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 0, k.side, ticks / 2, (uint32_t*)nullptr);
        hipLaunchKernelGGL(k_spin, dim3(1024), dim3(256), 0, k.main, ticks / 2, (uint32_t*)nullptr);
        CK(hipEventRecord(k.e_join, k.side));
        CK(hipStreamWaitEvent(k.main, k.e_join, 0));
      } else
        hipLaunchKernelGGL(k_spin, dim3(2048), dim3(256), 0, k.main, ticks, (uint32_t*)nullptr);
      if (l == chain_len / 2) { CK(hipEventRecord(k.e_mid, k.main)); chain_last = k.e_mid; }
    }
    if (md.memset_ctr) CK(hipMemcpyAsync(k.h_ctr, k.d_out, 4096, hipMemcpyDeviceToHost, k.main));
    CK(hipEventRecord(k.e_k, k.main));
    if (md.d2h) {
      CK(hipStreamWaitEvent(k.d2h, k.e_k, 0));
      CK(hipMemcpyAsync(k.h_out, k.d_out, md.out_bytes, hipMemcpyDeviceToHost, k.d2h));
      CK(hipEventRecord(k.e_done, k.d2h));
    } else
      CK(hipEventRecord(k.e_done, k.main));
  };
  auto pipeline = [&](int n) {
    for (int i = 0; i < n_calls && i < n; i++) begin(i);
    for (int i = 0; i < n; i++) {
      CK(hipEventSynchronize(c[i % n_calls].e_done));
      if (i + n_calls < n) begin(i + n_calls);
    }
  };
  pipeline(2 * n_calls);
  CK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  pipeline(steps);
  CK(hipDeviceSynchronize());
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
  for (auto& k : c) {
    (void)hipStreamDestroy(k.main);
    if (k.side) hipStreamDestroy(k.side);
    if (k.prio && !md.prio_shared) hipStreamDestroy(k.prio);
    hipHostFree(k.h_ctr);
    if (md.own_copy_streams) { hipStreamDestroy(k.h2d); hipStreamDestroy(k.d2h); }
    for (hipEvent_t e : {k.e_in, k.e_fork, k.e_join, k.e_k, k.e_mid, k.e_done, k.e_ids}) hipEventDestroy(e);
    hipHostFree(k.h_in); hipHostFree(k.h_out); hipFree(k.d_in); hipFree(k.d_out);
  }
  hipStreamDestroy(sh); hipStreamDestroy(sd);
  if (sp) hipStreamDestroy(sp);
  return ms;
}

int main(int argc, char** argv) {
  const size_t IN = 87u << 20, OUT = 80u << 20;
  const Mode modes[] = {
      {"base", true, true, true, false, false, IN, OUT},
      {"no_d2h", true, false, true, false, false, IN, OUT},
      {"no_h2d", false, true, true, false, false, IN, OUT},
      {"no_copies", false, false, true, false, false, IN, OUT},
      {"one_stream", true, true, false, false, false, IN, OUT},
      {"own_copy_streams", true, true, true, true, false, IN, OUT},
      {"chained", true, true, true, false, true, IN, OUT},
      {"small_copies", true, true, true, false, false, 1u << 20, 1u << 20},
      {"prio_lane", true, true, true, false, false, IN, OUT, true, false},
      {"memset_ctr", true, true, true, false, false, IN, OUT, false, true},
      {"library_like", true, true, true, false, true, IN, OUT, true, true},
      {"prio_shared", true, true, true, false, false, IN, OUT, true, false, true},
      {"library_like_prio_shared", true, true, true, false, true, IN, OUT, true, true, true},
  };
  int clk_khz = 100000;  // wall_clock64 ticks at a constant 100 MHz on gfx9
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeWallClockRate, 0);
  const int chain_len = 28;
  const double kernel_us = 150.0;
  const unsigned long long ticks = (unsigned long long)(kernel_us * 1e-6 * clk_khz * 1e3);
  printf("# %d kernels of %.0f us per call (%.1f ms of GPU time per step), GPU_MAX_HW_QUEUES=%s, wall clock %d kHz\n", chain_len, kernel_us,
         chain_len * kernel_us / 1e3, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(default)", clk_khz);
  printf("mode,calls_in_flight,ms_per_step\n");
  for (const Mode& md : modes) {
    bool want = argc <= 1;
    for (int a = 1; a < argc; a++) want = want || md.name == std::string(argv[a]) || std::string(argv[a]) == "all";
    if (!want) continue;
    for (int n : {1, 2, 4, 5, 6, 7, 8, 10, 12}) {
      const double ms = run(md, n, n >= 7 ? 24 : 40, chain_len, ticks);
      printf("%s,%d,%.3f\n", md.name, n, ms);
      fflush(stdout);
      if (ms > 400.0) break;  // the cliff: larger N only takes longer
    }
  }
  return 0;
}
