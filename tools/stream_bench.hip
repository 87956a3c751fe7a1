// stream_bench.hip -- what the memory system of this MI355X delivers for the access patterns of the fusion kernels, as a function of
// the footprint (the 256 MB infinity cache holds cfg2's 76 MB accumulator, nothing of cfg5's 12 GB).
// Build: hipcc --offload-arch=gfx950 -O3 tools/stream_bench.hip -o tools/stream_bench
//   read   : float4 streaming read
//   rmw    : float4 read + add + write in place (accumulator pattern, dense)
//   rows   : read-modify-write of 600-byte rows (C = 150), one wave per row, only runs of 64 consecutive rows out of every 256 touched
//            (what a view does to cfg5's accumulator), 16-byte pieces at 4-byte alignment
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_read(const float4* __restrict__ src, uint64_t n4, float* sink) {
  float s = 0.f;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
    const float4 v = src[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) *sink = s;
}
__global__ void k_rmw(float4* __restrict__ buf, uint64_t n4) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (uint64_t)gridDim.x * blockDim.x) {
    float4 v = buf[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    buf[i] = v;
  }
}
typedef float fvec4 __attribute__((ext_vector_type(4)));
typedef fvec4 fvec4_a4 __attribute__((aligned(4)));
// one wave per row of C floats; rows [256 g, 256 g + 64) of every group g of 256 rows
__global__ __launch_bounds__(64) void k_rows(float* __restrict__ acc, uint64_t rows, int C) {
  const int l = threadIdx.x;
  for (uint64_t w = blockIdx.x; w < rows / 4; w += gridDim.x) {
    const uint64_t r = (w / 64) * 256 + (w % 64);
    float* row = acc + r * C;
    const int c = 4 * l;
    if (c + 4 <= C) {
      fvec4 v = *reinterpret_cast<const fvec4_a4*>(row + c);
      v += 1.f;
      *reinterpret_cast<fvec4_a4*>(row + c) = v;
    } else if (c < C) {
      for (int e = c; e < C; e++) row[e] += 1.f;
    }
  }
}
template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int main() {
  const uint64_t maxb = 12ull << 30;
  float* buf; CK(hipMalloc(&buf, maxb)); CK(hipMemset(buf, 0, maxb));
  float* sink; CK(hipMalloc(&sink, 4));
  for (uint64_t mb : {64ull, 128ull, 256ull, 512ull, 1024ull, 4096ull, 12288ull}) {
    const uint64_t bytes = mb << 20, n4 = bytes / 16;
    const int reps = mb <= 1024 ? 20 : 4;
    const float tr = timeit([&] { hipLaunchKernelGGL(k_read, dim3(256 * 16), dim3(256), 0, 0, (const float4*)buf, n4, sink); }, reps);
    const float tw = timeit([&] { hipLaunchKernelGGL(k_rmw, dim3(256 * 16), dim3(256), 0, 0, (float4*)buf, n4); }, reps);
    const uint64_t rows = bytes / 600;
    const float tq = timeit([&] { hipLaunchKernelGGL(k_rows, dim3(256 * 32), dim3(64), 0, 0, buf, rows, 150); }, reps);
    printf("%6llu MB: read %7.1f us (%5.2f TB/s)   rmw %7.1f us (%5.2f TB/s r+w)   600-B rows, 1/4 touched: %7.1f us (%5.2f TB/s r+w)\n",
           (unsigned long long)mb, tr * 1e3, bytes / 1e12 / (tr * 1e-3), tw * 1e3, 2.0 * bytes / 1e12 / (tw * 1e-3), tq * 1e3,
           2.0 * (rows / 4) * 600 / 1e12 / (tq * 1e-3));
  }
  return 0;
}
