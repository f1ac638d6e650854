// random_access.hip — what the memory system of one MI355X sustains for the access pattern of k_probe:
// independent or dependent random 16-byte loads over a multi-GB array.  Prints loads/s, the sector
// bandwidth they amount to, and the latency of a dependent chain at full occupancy.
//   hipcc --offload-arch=gfx950 -O3 -o random_access random_access.hip && ./random_access [GiB] [parts]
// parts > 1 (round 6): the same bytes as `parts` separate allocations, a lane's load going to a random one of them — does it matter
// for the address translation whether the probe's tables (5.3 + 1.1 + 0.3 + 4.0 + ... GB) live in one allocation or in several?
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

struct __attribute__((aligned(16))) Key { uint32_t t[4]; };

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// `chain` dependent steps per lane, `par` independent chains per lane
template <int PAR>
__global__ void __launch_bounds__(256) k_rand(const Key* a, uint64_t n, int chain, uint32_t* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x[PAR];
#pragma unroll
  for (int u = 0; u < PAR; u++) x[u] = mix(t * PAR + u + 1);
  for (int c = 0; c < chain; c++) {
    Key k[PAR];
#pragma unroll
    for (int u = 0; u < PAR; u++) k[u] = a[(uint64_t)x[u] % n];
#pragma unroll
    for (int u = 0; u < PAR; u++) x[u] = mix(x[u] + k[u].t[0] + k[u].t[3]);
  }
  uint32_t s = 0;
#pragma unroll
  for (int u = 0; u < PAR; u++) s += x[u];
  if (s == 0x12345678u) out[0] = s;
}

struct Parts { const Key* p[16]; };
template <int PAR>
__global__ void __launch_bounds__(256) k_rand_parts(Parts ps, int parts, uint64_t n_each, int chain, uint32_t* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x[PAR];
#pragma unroll
  for (int u = 0; u < PAR; u++) x[u] = mix(t * PAR + u + 1);
  for (int c = 0; c < chain; c++) {
    Key k[PAR];
#pragma unroll
    for (int u = 0; u < PAR; u++) k[u] = ps.p[(x[u] >> 8) % (uint32_t)parts][(uint64_t)mix(x[u]) % n_each];
#pragma unroll
    for (int u = 0; u < PAR; u++) x[u] = mix(x[u] + k[u].t[0] + k[u].t[3]);
  }
  uint32_t s = 0;
#pragma unroll
  for (int u = 0; u < PAR; u++) s += x[u];
  if (s == 0x12345678u) out[0] = s;
}

// binary-search-like: each step halves a window of `span` keys placed at a random base (locality of the later steps)
__global__ void __launch_bounds__(256) k_bisect(const Key* a, uint64_t n, int span, uint32_t* out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t x = mix(t + 1);
  uint64_t lo = (uint64_t)x % (n - span), hi = lo + span;
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    const Key k = a[mid];
    if ((k.t[0] ^ x) & 1) lo = mid + 1; else hi = mid;
    x = mix(x + k.t[1]);
  }
  if (x == 0x12345678u) out[0] = x;
}

__global__ void k_fill(Key* a, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    Key k; k.t[0] = mix((uint32_t)i); k.t[1] = mix((uint32_t)i + 7); k.t[2] = (uint32_t)i; k.t[3] = mix((uint32_t)(i >> 3));
    a[i] = k;
  }
}

template <class F>
static double time_ms(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; r++) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const double gib = argc > 1 ? atof(argv[1]) : 2.0;
  const uint64_t n = (uint64_t)(gib * (1ull << 30)) / sizeof(Key);
  const int parts = argc > 2 ? atoi(argv[2]) : 1;
  if (parts > 1 && parts <= 16) {
    Parts ps{};
    uint32_t* out2;
    const uint64_t n_each = n / parts;
    if (hipMalloc(&out2, 4) != hipSuccess) return 1;
    for (int k = 0; k < parts; k++) {
      Key* q;
      if (hipMalloc(&q, n_each * sizeof(Key)) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, q, n_each);
      ps.p[k] = q;
    }
    hipDeviceSynchronize();
    const uint32_t lanes2 = 25u << 20;
    printf("%d allocations of %.2f GiB, %u lanes\n", parts, gib / parts, lanes2);
    for (int chain : {1, 4}) {
      double ms1 = time_ms([&] { hipLaunchKernelGGL(k_rand_parts<1>, dim3(lanes2 / 256), dim3(256), 0, 0, ps, parts, n_each, chain, out2); }, 3);
      double ms4 = time_ms([&] { hipLaunchKernelGGL(k_rand_parts<4>, dim3(lanes2 / 256), dim3(256), 0, 0, ps, parts, n_each, chain, out2); }, 3);
      printf("chain %2d  par 1: %7.3f ms  %6.1f G loads/s   par 4: %7.3f ms  %6.1f G loads/s\n", chain, ms1, lanes2 * (double)chain / ms1 / 1e6, ms4,
             lanes2 * 4.0 * chain / ms4 / 1e6);
    }
    return 0;
  }
  Key* a; uint32_t* out;
  if (hipMalloc(&a, n * sizeof(Key)) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, a, n);
  hipDeviceSynchronize();
  const uint32_t lanes = 25u << 20;  // ~ one C3 step of probes
  const dim3 grid(lanes / 256), block(256);
  printf("array %.1f GiB (%llu keys of 16 B), %u lanes\n", gib, (unsigned long long)n, lanes);
  for (int chain : {1, 4, 12}) {
    double ms1 = time_ms([&] { hipLaunchKernelGGL(k_rand<1>, grid, block, 0, 0, a, n, chain, out); }, 3);
    double ms4 = time_ms([&] { hipLaunchKernelGGL(k_rand<4>, grid, block, 0, 0, a, n, chain, out); }, 3);
    printf("chain %2d  par 1: %7.3f ms  %6.1f G loads/s (%5.2f TB/s of 64-B sectors)   par 4: %7.3f ms  %6.1f G loads/s (%5.2f TB/s)\n",
           chain, ms1, lanes * (double)chain / ms1 / 1e6, lanes * (double)chain * 64 / ms1 / 1e9,
           ms4, lanes * 4.0 * chain / ms4 / 1e6, lanes * 4.0 * chain * 64 / ms4 / 1e9);
  }
  for (int span : {64, 4096, 1 << 20}) {
    int steps = 0; for (int s = span; s > 0; s >>= 1) steps++;
    double ms = time_ms([&] { hipLaunchKernelGGL(k_bisect, grid, block, 0, 0, a, n, span, out); }, 3);
    printf("bisect span %7d (%2d steps): %7.3f ms  %6.1f G probes/s  %6.1f G loads/s\n", span, steps, ms, lanes / ms / 1e6,
           lanes * (double)steps / ms / 1e6);
  }
  return 0;
}
