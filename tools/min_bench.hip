// min_bench.hip -- prices the depth-test update of the rasteriser on MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 tools/min_bench.hip -o tools/min_bench
// N = 1080p pixels.  Each mode issues ONE update per pixel slot (M = 1.4 M updates, the fragment count of a cfg2 view):
//   u64 atomicMin / u32 atomicMin / u64 plain store / u32 plain store / u64 load+compare+conditional atomicMin,
// with lane -> address mappings: coalesced (lane i -> slot i), strided-2/-8/-16 (neighbouring lanes 2/8/16 slots apart, what one lane
// per triangle produces), random.  Also: the same updates when every update LOSES (value larger than what is stored).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename T, int MODE>  // MODE 0 atomicMin, 1 plain store, 2 load + compare + conditional atomicMin
__global__ void k(T* buf, const uint32_t* addr, uint32_t M, T val) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const uint32_t a = addr[i];
  if (MODE == 0) atomicMin(&buf[a], val);
  else if (MODE == 1) buf[a] = val;
  else { if (__builtin_nontemporal_load(&buf[a]) > val) atomicMin(&buf[a], val); }
}
template <typename T> __global__ void k_fill(T* buf, uint32_t N, T v) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) buf[i] = v;
}
template <typename P, typename F> float timeit(P pre, F f, int reps = 10) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float tot = 0;
  for (int i = 0; i < reps + 2; i++) {
    pre(); CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (i >= 2) tot += ms;
  }
  return tot / reps;
}
int main() {
  const uint32_t N = 1920 * 1080, M = 1400000;
  uint64_t* b64; uint32_t* b32; CK(hipMalloc(&b64, N * 8)); CK(hipMalloc(&b32, N * 4));
  std::mt19937 rng(1);
  struct Pat { const char* name; std::vector<uint32_t> a; };
  std::vector<Pat> pats;
  { Pat p{"coalesced", std::vector<uint32_t>(M)}; std::iota(p.a.begin(), p.a.end(), 0u); pats.push_back(p); }
  for (int s : {2, 8, 16, 64}) {
    // lane l of wave w -> slot ((w*64 + l) * s) mod N, offset by the wrap count so that every slot is hit at most once
    Pat p{s == 2 ? "stride-2" : s == 8 ? "stride-8" : s == 16 ? "stride-16" : "stride-64", std::vector<uint32_t>(M)};
    for (uint32_t i = 0; i < M; i++) { const uint64_t x = (uint64_t)i * s; p.a[i] = (uint32_t)((x % N + x / N) % N); }
    pats.push_back(p);
  }
  { Pat p{"random", std::vector<uint32_t>(N)}; std::iota(p.a.begin(), p.a.end(), 0u); std::shuffle(p.a.begin(), p.a.end(), rng); p.a.resize(M); pats.push_back(p); }
  uint32_t* da; CK(hipMalloc(&da, M * 4));
  const dim3 g((M + 255) / 256), b(256), gf((N + 255) / 256);
  for (auto& p : pats) {
    CK(hipMemcpy(da, p.a.data(), M * 4, hipMemcpyHostToDevice));
    auto arm64 = [&] { hipLaunchKernelGGL(k_fill<uint64_t>, gf, b, 0, 0, b64, N, ~0ull); };
    auto arm32 = [&] { hipLaunchKernelGGL(k_fill<uint32_t>, gf, b, 0, 0, b32, N, ~0u); };
    auto low64 = [&] { hipLaunchKernelGGL(k_fill<uint64_t>, gf, b, 0, 0, b64, N, 0ull); };
    auto low32 = [&] { hipLaunchKernelGGL(k_fill<uint32_t>, gf, b, 0, 0, b32, N, 0u); };
    float t[8];
    t[0] = timeit(arm64, [&] { hipLaunchKernelGGL((k<uint64_t, 0>), g, b, 0, 0, b64, da, M, (uint64_t)12345); });
    t[1] = timeit(arm32, [&] { hipLaunchKernelGGL((k<uint32_t, 0>), g, b, 0, 0, b32, da, M, 12345u); });
    t[2] = timeit(arm64, [&] { hipLaunchKernelGGL((k<uint64_t, 1>), g, b, 0, 0, b64, da, M, (uint64_t)12345); });
    t[3] = timeit(arm32, [&] { hipLaunchKernelGGL((k<uint32_t, 1>), g, b, 0, 0, b32, da, M, 12345u); });
    t[4] = timeit(arm64, [&] { hipLaunchKernelGGL((k<uint64_t, 2>), g, b, 0, 0, b64, da, M, (uint64_t)12345); });
    t[5] = timeit(low64, [&] { hipLaunchKernelGGL((k<uint64_t, 0>), g, b, 0, 0, b64, da, M, (uint64_t)12345); });
    t[6] = timeit(low64, [&] { hipLaunchKernelGGL((k<uint64_t, 2>), g, b, 0, 0, b64, da, M, (uint64_t)12345); });
    t[7] = timeit(low32, [&] { hipLaunchKernelGGL((k<uint32_t, 0>), g, b, 0, 0, b32, da, M, 12345u); });
    printf("%-10s min64 %6.1f  min32 %6.1f  st64 %6.1f  st32 %6.1f  ld+min64 %6.1f | losing: min64 %6.1f  ld+min64 %6.1f  min32 %6.1f  (us per 1.4 M updates)\n",
           p.name, t[0] * 1e3, t[1] * 1e3, t[2] * 1e3, t[3] * 1e3, t[4] * 1e3, t[5] * 1e3, t[6] * 1e3, t[7] * 1e3);
  }
  return 0;
}
