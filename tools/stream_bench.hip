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
// 600-byte rows read from SCATTERED positions of a large buffer, `inflight` rows per wave at a time: what k_fuse_tri_wide does to the
// class-vector images at cfg5 (a visible triangle's pixel is one 600-byte row of a 5.3 GB image, eight images per launch).
//   stride = 0: row index = a hash of (wave, step) -- no locality at all
//   stride > 0: wave w reads rows (w * stride + step) mod rows: neighbouring waves read rows `stride` apart (an image column apart)
template <int INFLIGHT>
__global__ __launch_bounds__(64) void k_gather_rows(const float* __restrict__ buf, uint64_t rows, int C, uint32_t steps, uint64_t stride, float* sink,
                                                    int row_floats) {
  const int l = threadIdx.x;
  const int c = 4 * l;
  float s = 0.f;
  for (uint32_t st = 0; st < steps; st += INFLIGHT) {
    fvec4 v[INFLIGHT];
#pragma unroll
    for (int b = 0; b < INFLIGHT; b++) {
      uint64_t r;
      if (stride) r = ((uint64_t)blockIdx.x * stride + st + b) % rows;
      else {
        uint64_t h = ((uint64_t)blockIdx.x * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)(st + b) * 0xBF58476D1CE4E5B9ull);
        h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
        r = h % rows;
      }
      v[b] = fvec4{0.f, 0.f, 0.f, 0.f};
      if (c + 4 <= C) v[b] = *reinterpret_cast<const fvec4_a4*>(buf + r * (uint64_t)row_floats + c);
    }
#pragma unroll
    for (int b = 0; b < INFLIGHT; b++) s += v[b].x + v[b].y + v[b].z + v[b].w;
  }
  if (s == 123.456f) *sink = s;
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}
int gather_main() {
  // scattered 600-byte rows: footprint 12 GB and 42 GB (eight cfg5 images), 2 / 4 / 8 rows in flight per wave, 8 waves per SIMD worth of workgroups
  float* sink; CK(hipMalloc(&sink, 4));
  for (uint64_t gb : {12ull, 42ull}) {
    const uint64_t bytes = gb << 30;
    float* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    const uint64_t rows = bytes / 600;
    const uint32_t waves = 312500, steps = 96;            // ~ the waves of a cfg5 launch, ~ the rows a wave reads in it
    for (uint64_t stride : {0ull, 2160ull, 1ull}) {
      const double total = (double)waves * steps * 600.0;
      const float t2 = timeit([&] { hipLaunchKernelGGL(k_gather_rows<2>, dim3(waves), dim3(64), 0, 0, buf, rows, 150, steps, stride, sink, 150); }, 3);
      const float t4 = timeit([&] { hipLaunchKernelGGL(k_gather_rows<4>, dim3(waves), dim3(64), 0, 0, buf, rows, 150, steps, stride, sink, 150); }, 3);
      const float t8 = timeit([&] { hipLaunchKernelGGL(k_gather_rows<8>, dim3(waves), dim3(64), 0, 0, buf, rows, 150, steps, stride, sink, 150); }, 3);
      if (stride == 0) {   // the same 600 bytes read from rows padded to 640 bytes (five whole 128-byte lines, aligned): what padding the accumulator rows would buy
        const uint64_t prow = bytes / 640;
        const float tp = timeit([&] { hipLaunchKernelGGL(k_gather_rows<4>, dim3(waves), dim3(64), 0, 0, buf, prow, 150, steps, stride, sink, 160); }, 3);
        printf("   ... 600 of 640-byte rows at 128-byte alignment, 4 in flight: %6.2f TB/s useful (%6.2f TB/s of whole lines)\n",
               total / 1e12 / (tp * 1e-3), total * 640.0 / 600.0 / 1e12 / (tp * 1e-3));
      }
      printf("gather of 600-B rows, footprint %2llu GB, %s: 2 in flight %6.2f TB/s   4 in flight %6.2f TB/s   8 in flight %6.2f TB/s   (%.1f GB read)\n",
             (unsigned long long)gb, stride == 0 ? "hashed positions      " : stride == 1 ? "consecutive rows      " : "rows 2160 apart (x+1) ",
             total / 1e12 / (t2 * 1e-3), total / 1e12 / (t4 * 1e-3), total / 1e12 / (t8 * 1e-3), total / 1e9);
    }
    CK(hipFree(buf));
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'g') return gather_main();
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
