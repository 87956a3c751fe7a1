// atomic_scaling.hip -- is the float-atomic rate a per-CU (issue path) or a chip-level (memory side) limit?
// Persistent grid of `blocks` workgroups x 256 threads replays the same total work; rows either in order or shuffled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int C = 19;

template <int MODE>  // 0 atomic, 1 store
__global__ void k(float* acc, const uint32_t* perm, uint32_t P, float v) {
  const uint64_t total = (uint64_t)P * C;
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t r = (uint32_t)(e / C), c = (uint32_t)(e - (uint64_t)r * C);
    float* p = &acc[(uint64_t)(perm ? perm[r] : r) * C + c];
    if (MODE == 0) unsafeAtomicAdd(p, v); else *p = v;
  }
}
template <typename F> float timeit(F f, int reps = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const uint32_t P = 1000000;
  float* acc; CK(hipMalloc(&acc, (size_t)P * C * 4)); CK(hipMemset(acc, 0, (size_t)P * C * 4));
  std::vector<uint32_t> perm(P); std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
  uint32_t* dperm; CK(hipMalloc(&dperm, P * 4)); CK(hipMemcpy(dperm, perm.data(), P * 4, hipMemcpyHostToDevice));
  for (int blocks : {32, 64, 128, 256, 512, 1024, 2048, 8192}) {
    float a0 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, acc, (const uint32_t*)nullptr, P, 1.f); });
    float a1 = timeit([&] { hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, acc, dperm, P, 1.f); });
    float s1 = timeit([&] { hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, acc, dperm, P, 1.f); });
    printf("blocks %5d: atomics in-order %8.1f us, shuffled rows %8.1f us | stores shuffled %8.1f us\n", blocks, a0 * 1e3, a1 * 1e3, s1 * 1e3);
  }
  return 0;
}
