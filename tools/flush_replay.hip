// flush_replay.hip -- replays accumulator-row access lists (tools/make_flush_lists.py) to price flush
// strategies of the scatter-add on MI355X: all-atomic vs hybrid (plain RMW for tile-exclusive rows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int C = 19;
constexpr int SP = 32;  // padded row stride

// MODE 0: atomics for every row; MODE 1: rows without the high bit use plain load+add+store
template <int MODE, bool STREAM, int S = C>
__global__ void k_replay(float* acc, const uint32_t* list, uint32_t n, const float4* src, uint64_t n4, float* sink) {
  const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  if (STREAM) {
    // each thread streams its share of the probs image first (same total bytes as one view)
    const uint64_t total_threads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = e; i < n4; i += total_threads) { const float4 v = src[i]; s += v.x + v.y + v.z + v.w; }
  }
  if (e < (uint64_t)n * C) {
    const uint32_t r = (uint32_t)(e / C), c = (uint32_t)(e - (uint64_t)r * C);
    const uint32_t row = list[r];
    float* p = &acc[(uint64_t)(row & 0x7FFFFFFFu) * S + c];
    const float v = 1.0f + s * 1e-30f;
    if (MODE == 1 && !(row & 0x80000000u)) *p = *p + v; else unsafeAtomicAdd(p, v);
  }
  if (s == 123.456f) *sink = s;
}

__global__ void k_hist(uint32_t* cnt, const uint32_t* list, uint32_t n) {
  const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < n) atomicAdd(&cnt[list[e] & 0x7FFFFFFFu], 1u);
}

template <typename F> float timeit(F f, int reps = 20) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main(int argc, char** argv) {
  const uint32_t P = 1000000;
  float* acc; CK(hipMalloc(&acc, (size_t)P * SP * 4)); CK(hipMemset(acc, 0, (size_t)P * SP * 4));
  uint32_t* cnt; CK(hipMalloc(&cnt, P * 4)); CK(hipMemset(cnt, 0, P * 4));
  const uint64_t n4 = 2073600ull * C / 4;
  const int NSRC = 4;  // rotate through distinct source images so the stream is not cache resident
  float4* src; CK(hipMalloc(&src, n4 * 16 * NSRC)); CK(hipMemset(src, 0, n4 * 16 * NSRC));
  float* sink; CK(hipMalloc(&sink, 4));
  for (int a = 1; a < argc; a++) {
    FILE* f = fopen(argv[a], "rb"); if (!f) { printf("cannot open %s\n", argv[a]); continue; }
    fseek(f, 0, SEEK_END); const long bytes = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint32_t> h(bytes / 4); if (fread(h.data(), 1, bytes, f) != (size_t)bytes) return 1; fclose(f);
    const uint32_t n = (uint32_t)h.size();
    uint32_t natomic = 0; for (auto v : h) natomic += v >> 31;
    uint32_t* d; CK(hipMalloc(&d, bytes)); CK(hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice));
    const dim3 g((unsigned)(((uint64_t)n * C + 255) / 256)), b(256);
    int rot = 0;
    float t0 = timeit([&] { hipLaunchKernelGGL((k_replay<0, false>), g, b, 0, 0, acc, d, n, src, n4, sink); });
    float t1 = timeit([&] { hipLaunchKernelGGL((k_replay<1, false>), g, b, 0, 0, acc, d, n, src, n4, sink); });
    float t2 = timeit([&] { hipLaunchKernelGGL((k_replay<0, true>), g, b, 0, 0, acc, d, n, src + (rot++ % NSRC) * n4, n4, sink); });
    float t3 = timeit([&] { hipLaunchKernelGGL((k_replay<1, true>), g, b, 0, 0, acc, d, n, src + (rot++ % NSRC) * n4, n4, sink); });
    float t4 = timeit([&] { hipLaunchKernelGGL((k_replay<0, false, SP>), g, b, 0, 0, acc, d, n, src, n4, sink); });
    float t5 = timeit([&] { hipLaunchKernelGGL((k_replay<0, true, SP>), g, b, 0, 0, acc, d, n, src + (rot++ % NSRC) * n4, n4, sink); });
    float th = timeit([&] { hipLaunchKernelGGL(k_hist, dim3((n + 255) / 256), b, 0, 0, cnt, d, n); });
    printf("%-40s rows %8u (atomic-marked %7u): atomics %7.1f us | hybrid %7.1f us | +stream: atomics %7.1f us hybrid %7.1f us | padded32: atomics %7.1f us +stream %7.1f us | hist %6.1f us\n",
           argv[a], n, natomic, t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3, t5 * 1e3, th * 1e3);
    CK(hipFree(d));
  }
  return 0;
}
