// pmc_calibrate.hip — known byte counts in the access patterns of the match() kernels, to calibrate rocprofv3's
// FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: "calibrate on a known byte count in your own access pattern").
// Every kernel touches an 8 GiB array exactly once (far beyond L2 + Infinity Cache), so the HBM bytes are known:
//   k_stream16   16 B per lane, consecutive lanes contiguous (k_expand_a's fast path)
//   k_stream8     8 B per lane, consecutive lanes contiguous (k_expand_h's plane rows)
//   k_random32   one 32 B load per lane at a random 32 B-aligned address, every 128 B line touched at most once
//                (k_probe's key loads): requested = 32 B, the memory system moves a line
//   k_random16   same with 16 B loads (trigram entries)
//   k_random16_pair  two 16 B loads per lane, one in each 64 B half of the lane's line: fill granularity of a random miss
//   k_write8      8 B per lane streaming store (queue appends)
// tools/pmc_calibrate.py runs this under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` and writes the factors.
//   hipcc --offload-arch=gfx950 -O3 -o pmc_calibrate pmc_calibrate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct __attribute__((aligned(16))) V16 { uint32_t t[4]; };
struct __attribute__((aligned(32))) V32 { uint32_t t[8]; };

__global__ void __launch_bounds__(256) k_stream16(const V16* a, uint64_t n, uint32_t* out) {
  uint32_t s = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += a[i].t[0] ^ a[i].t[3];
  if (s == 0x12345678u) out[0] = s;
}
__global__ void __launch_bounds__(256) k_stream8(const unsigned long long* a, uint64_t n, uint32_t* out) {
  unsigned long long s = 0;
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += a[i];
  if (s == 0x12345678ull) out[0] = (uint32_t)s;
}
// line l of the array is visited by exactly one lane: l = (t * odd) mod n_lines with n_lines a power of two is a bijection
__global__ void __launch_bounds__(256) k_random32(const V32* a, uint64_t n_lines, uint32_t* out) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (t >= n_lines) return;
  const uint64_t line = (t * 0x9E3779B97F4A7C15ull) & (n_lines - 1);
  const V32 v = a[line * 4 + (t & 3)];  // 128 B line = 4 x 32 B
  if ((v.t[0] ^ v.t[7]) == 0x12345678u) out[0] = v.t[1];
}
__global__ void __launch_bounds__(256) k_random16(const V16* a, uint64_t n_lines, uint32_t* out) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (t >= n_lines) return;
  const uint64_t line = (t * 0x9E3779B97F4A7C15ull) & (n_lines - 1);
  const V16 v = a[line * 8 + (t & 7)];
  if ((v.t[0] ^ v.t[3]) == 0x12345678u) out[0] = v.t[1];
}
// Both halves of every line, by the same lane: tells a 128-byte fill (the second load hits the line just filled: the counter
// stays where k_random16 leaves it) from 64-byte sector fills (the counter doubles).
__global__ void __launch_bounds__(256) k_random16_pair(const V16* a, uint64_t n_lines, uint32_t* out) {
  const uint64_t t = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
  if (t >= n_lines) return;
  const uint64_t line = (t * 0x9E3779B97F4A7C15ull) & (n_lines - 1);
  const V16 v = a[line * 8 + (t & 3)];
  const V16 u = a[line * 8 + 4 + (t & 3)];
  if ((v.t[0] ^ u.t[3]) == 0x12345678u) out[0] = v.t[1];
}
__global__ void __launch_bounds__(256) k_write8(unsigned long long* a, uint64_t n) {
  for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) a[i] = i * 0x9E3779B97F4A7C15ull;
}

int main() {
  const uint64_t bytes = 8ull << 30;
  void* a; uint32_t* out;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { fprintf(stderr, "alloc failed\n"); return 1; }
  hipMemset(a, 1, bytes);
  hipDeviceSynchronize();
  const uint64_t n_lines = bytes / 128;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_stream16, dim3(65536), dim3(256), 0, 0, (const V16*)a, bytes / 16, out);
    hipLaunchKernelGGL(k_stream8, dim3(65536), dim3(256), 0, 0, (const unsigned long long*)a, bytes / 8, out);
    hipLaunchKernelGGL(k_random32, dim3((unsigned)(n_lines / 256)), dim3(256), 0, 0, (const V32*)a, n_lines, out);
    hipLaunchKernelGGL(k_random16, dim3((unsigned)(n_lines / 256)), dim3(256), 0, 0, (const V16*)a, n_lines, out);
    hipLaunchKernelGGL(k_random16_pair, dim3((unsigned)(n_lines / 256)), dim3(256), 0, 0, (const V16*)a, n_lines, out);
    hipLaunchKernelGGL(k_write8, dim3(65536), dim3(256), 0, 0, (unsigned long long*)a, bytes / 8);
    hipDeviceSynchronize();
  }
  printf("{\"array_bytes\": %llu, \"lines\": %llu}\n", (unsigned long long)bytes, (unsigned long long)n_lines);
  return 0;
}
