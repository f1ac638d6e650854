// cu_mask.hip — do two HIP streams with disjoint compute-unit masks (hipExtStreamCreateWithCUMask) run their kernels side
// by side on one MI355X, and what does a kernel lose on half of the CUs?  Two kernels: VALU-bound (fma chain) and
// memory-bound (random 16-byte loads over 4 GiB, the probe's access pattern).
//   hipcc --offload-arch=gfx950 -O3 -o cu_mask cu_mask.hip && ./cu_mask
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void __launch_bounds__(256) k_valu(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-4f, c = 1.0001f;
  for (int i = 0; i < iters; i++) { a = a * c + b; b = b * c + a; a = a * c + b; b = b * c + a; }
  if (a + b == 12345.f) out[0] = a;
}

struct __attribute__((aligned(16))) Key { uint32_t t[4]; };
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
__global__ void __launch_bounds__(256) k_mem(const Key* a, uint64_t n, int chain, uint32_t* out) {
  uint32_t x = mix(blockIdx.x * 256 + threadIdx.x + 1);
  for (int c = 0; c < chain; c++) { const Key k = a[(uint64_t)x % n]; x = mix(x + k.t[0] + k.t[3]); }
  if (x == 0x12345678u) out[0] = x;
}

static hipStream_t masked(int n_cu, int mode, int group) {
  std::vector<uint32_t> m((n_cu + 31) / 32, 0u);
  for (int i = 0; i < n_cu; i++) {
    bool mine = mode == 1 ? ((i < n_cu / 2) == (group == 0)) : mode == 2 ? ((i % 8) / 4 == group) : ((i % 2) == group);
    if (mine) m[i / 32] |= 1u << (i % 32);
  }
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)m.size(), m.data()) != hipSuccess) { printf("mask stream failed\n"); return nullptr; }
  return s;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int n_cu = prop.multiProcessorCount;
  printf("%s, %d CUs\n", prop.gcnArchName, n_cu);
  float* d_out; CK(hipMalloc(&d_out, 64));
  const uint64_t n = (4ull << 30) / 16;
  Key* d_a; CK(hipMalloc(&d_a, n * 16)); CK(hipMemset(d_a, 1, n * 16));
  hipStream_t plain[2]; CK(hipStreamCreateWithFlags(&plain[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&plain[1], hipStreamNonBlocking));
  auto run = [&](const char* what, hipStream_t s0, hipStream_t s1, int k0, int k1) {  // kernel kinds: 0 valu, 1 mem, -1 none
    auto launch = [&](hipStream_t s, int kind) {
      if (kind == 0) hipLaunchKernelGGL(k_valu, dim3(n_cu * 32), dim3(256), 0, s, d_out, 20000);
      if (kind == 1) hipLaunchKernelGGL(k_mem, dim3(n_cu * 64), dim3(256), 0, s, d_a, n, 64, (uint32_t*)d_out);
    };
    for (int rep = 0; rep < 2; rep++) {
      hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      launch(s0, k0); if (s1) launch(s1, k1);
      hipDeviceSynchronize();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (rep == 1) printf("%-58s %8.3f ms\n", what, ms);
    }
  };
  run("valu, whole GPU", plain[0], nullptr, 0, -1);
  run("mem,  whole GPU", plain[0], nullptr, 1, -1);
  run("valu + valu, two plain streams", plain[0], plain[1], 0, 0);
  run("valu + mem,  two plain streams", plain[0], plain[1], 0, 1);
  for (int mode = 1; mode <= 3; mode++) {
    hipStream_t a = masked(n_cu, mode, 0), b = masked(n_cu, mode, 1);
    if (!a || !b) continue;
    char buf[128];
    const char* names[4] = {"", "lower/upper half of the bits", "bits by (i % 8) / 4", "even / odd bits"};
    snprintf(buf, sizeof buf, "[%s] valu, group 0 alone", names[mode]); run(buf, a, nullptr, 0, -1);
    snprintf(buf, sizeof buf, "[%s] valu, group 1 alone", names[mode]); run(buf, b, nullptr, 0, -1);
    snprintf(buf, sizeof buf, "[%s] mem,  group 0 alone", names[mode]); run(buf, a, nullptr, 1, -1);
    snprintf(buf, sizeof buf, "[%s] valu g0 + valu g1", names[mode]); run(buf, a, b, 0, 0);
    snprintf(buf, sizeof buf, "[%s] valu g0 + mem g1", names[mode]); run(buf, a, b, 0, 1);
    snprintf(buf, sizeof buf, "[%s] mem g0 + mem g1", names[mode]); run(buf, a, b, 1, 1);
    hipStreamDestroy(a); hipStreamDestroy(b);
  }
  return 0;
}
