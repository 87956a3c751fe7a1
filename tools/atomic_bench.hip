// atomic_bench.hip -- micro-benchmark that priced the accumulator update of the scatter-add on MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/atomic_bench.hip -o tools/atomic_bench
// Measures, over a float32[P*C] accumulator (P = 1M rows, C = 19), GB/s of read-modify-write traffic for
//   agent-scope float atomics (memory-side), workgroup-scope float atomics (XCD-L2-side), plain load+add+store,
// with rows visited in order or in a random permutation (76-byte granules), plus a pure streaming read.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int C = 19;

template <int MODE>
__device__ __forceinline__ void upd(float* p, float v) {
  if (MODE == 0) unsafeAtomicAdd(p, v);                                                            // agent scope
  else if (MODE == 1) __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // L2-side
  else *p = *p + v;                                                                                // plain RMW
}

template <int MODE>
__global__ void k_rows(float* acc, const uint32_t* perm, uint32_t P, float v) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (uint64_t)P * C) return;
  const uint32_t r = (uint32_t)(e / C), c = (uint32_t)(e - (uint64_t)r * C);
  const uint32_t row = perm ? perm[r] : r;
  upd<MODE>(&acc[(uint64_t)row * C + c], v);
}

__global__ void k_stream(const float4* src, uint64_t n4, float* sink) {
  float s = 0.f;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) *sink = s;
}

template <typename F>
float timeit(F f, int reps = 20) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const uint32_t P = 1000000;
  const uint64_t M = (uint64_t)P * C;
  float* acc; CK(hipMalloc(&acc, M * 4)); CK(hipMemset(acc, 0, M * 4));
  std::vector<uint32_t> perm(P); std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
  uint32_t* dperm; CK(hipMalloc(&dperm, P * 4)); CK(hipMemcpy(dperm, perm.data(), P * 4, hipMemcpyHostToDevice));
  // locally shuffled permutation: rows shuffled inside windows of 64 (what neighbouring pixels look like)
  std::vector<uint32_t> lperm(P); std::iota(lperm.begin(), lperm.end(), 0u);
  for (uint32_t b = 0; b + 64 <= P; b += 64) std::shuffle(lperm.begin() + b, lperm.begin() + b + 64, rng);
  uint32_t* dlperm; CK(hipMalloc(&dlperm, P * 4)); CK(hipMemcpy(dlperm, lperm.data(), P * 4, hipMemcpyHostToDevice));
  const uint64_t SRC = 2073600ull * C;  // one 1080p probs image
  float* src; CK(hipMalloc(&src, SRC * 4 * 4)); CK(hipMemset(src, 0, SRC * 4 * 4));
  float* sink; CK(hipMalloc(&sink, 4));
  const dim3 g((unsigned)((M + 255) / 256)), b(256);
  const double rmw_gb = 2.0 * M * 4 / 1e9;
  struct { const char* name; const uint32_t* p; } orders[3] = {{"in-order", nullptr}, {"local-shuffle64", dlperm}, {"random", dperm}};
  for (auto& o : orders) {
    float t0 = timeit([&] { hipLaunchKernelGGL(k_rows<0>, g, b, 0, 0, acc, o.p, P, 1.0f); });
    float t1 = timeit([&] { hipLaunchKernelGGL(k_rows<1>, g, b, 0, 0, acc, o.p, P, 1.0f); });
    float t2 = timeit([&] { hipLaunchKernelGGL(k_rows<2>, g, b, 0, 0, acc, o.p, P, 1.0f); });
    printf("%-16s agent-atomic %8.1f us (%6.0f GB/s rmw)  wg-atomic %8.1f us (%6.0f GB/s)  plain-rmw %8.1f us (%6.0f GB/s)\n",
           o.name, t0 * 1e3, rmw_gb / (t0 * 1e-3), t1 * 1e3, rmw_gb / (t1 * 1e-3), t2 * 1e3, rmw_gb / (t2 * 1e-3));
  }
  for (int rep = 1; rep <= 4; rep *= 4) {
    const uint64_t n4 = SRC * rep / 4;
    float ts = timeit([&] { hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, (const float4*)src, n4, sink); });
    printf("stream read %6.1f MB: %8.1f us (%6.0f GB/s)\n", n4 * 16 / 1e6, ts * 1e3, n4 * 16 / 1e9 / (ts * 1e-3));
  }
  return 0;
}
