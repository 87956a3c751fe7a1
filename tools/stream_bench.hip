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
// Per-LANE rows of 160 bytes (C = 40: ten 16-byte loads of one lane) at hashed positions of a large buffer, read (the class vector of a
// visible pixel) or read-modify-written (the texel's accumulator row): what k_fuse_texel_multi does at cfg4 / cfg4t -- every visible
// pixel costs one such read and one such read-modify-write, 4.8 GB of accumulator at cfg4t.
template <bool RMW>
__global__ __launch_bounds__(256) void k_lane_rows(float* __restrict__ buf, uint64_t rows, uint32_t steps, float* sink) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float s = 0.f;
  for (uint32_t st = 0; st < steps; st++) {
    uint64_t h = (id * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)st * 0xBF58476D1CE4E5B9ull);
    h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
    float4* row = reinterpret_cast<float4*>(buf + (h % rows) * 40);
    float4 v[10];
#pragma unroll
    for (int k = 0; k < 10; k++) v[k] = row[k];
#pragma unroll
    for (int k = 0; k < 10; k++) {
      s += v[k].x + v[k].w;
      if (RMW) { v[k].x += 1.f; v[k].y += 1.f; v[k].z += 1.f; v[k].w += 1.f; row[k] = v[k]; }
    }
  }
  if (s == 123.456f) *sink = s;
}
// cfg4t's own pattern: the accumulator rows of the visible texels in ASCENDING order, one row in `stride` touched (672 102 visible
// pixels of a view among 64 M texel rows: one in 96; eight views of a launch: one in 12), beside one 160-byte row per lane gathered
// from hashed positions of a 200 MB class-vector image (`img`, or none).
__global__ __launch_bounds__(256) void k_lane_rows_sparse(float* __restrict__ buf, uint32_t stride, const float* __restrict__ img, uint32_t img_rows,
                                                          uint64_t lanes, float* sink) {
  const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= lanes) return;
  uint64_t h = id * 0x9E3779B97F4A7C15ull;
  h ^= h >> 29; h *= 0x94D049BB133111EBull; h ^= h >> 32;
  float4* row = reinterpret_cast<float4*>(buf + (id * stride + h % stride) * 40);
  float4 v[10], p[10];
#pragma unroll
  for (int k = 0; k < 10; k++) v[k] = row[k];
  if (img) {
    const float4* pr = reinterpret_cast<const float4*>(img + ((h >> 20) % img_rows) * 40);
#pragma unroll
    for (int k = 0; k < 10; k++) p[k] = pr[k];
  } else {
#pragma unroll
    for (int k = 0; k < 10; k++) p[k] = make_float4(1.f, 1.f, 1.f, 1.f);
  }
#pragma unroll
  for (int k = 0; k < 10; k++) { v[k].x += p[k].x; v[k].y += p[k].y; v[k].z += p[k].z; v[k].w += p[k].w; row[k] = v[k]; }
}
int lane_main() {
  float* sink; CK(hipMalloc(&sink, 4));
  {
    const uint64_t rows = 64268767, bytes = rows * 160;
    float *buf, *img; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    const uint32_t img_rows = 1296 * 968;
    CK(hipMalloc(&img, (size_t)img_rows * 160)); CK(hipMemset(img, 0, (size_t)img_rows * 160));
    for (uint32_t stride : {96u, 12u, 4u, 1u}) {
      const uint64_t lanes = rows / stride;
      for (int with_img = 0; with_img < 2; with_img++) {
        const float t = timeit([&] { hipLaunchKernelGGL(k_lane_rows_sparse, dim3((lanes + 255) / 256), dim3(256), 0, 0, buf, stride,
                                                        with_img ? img : (const float*)nullptr, img_rows, lanes, sink); }, 10);
        printf("ascending 160-B accumulator rows, one in %2u of 64 M read-modify-written%s: %6.1f us for %8llu rows = %5.2f G rows/s, %5.2f TB/s useful\n",
               stride, with_img ? " + a hashed 160-B row of a 200 MB image read" : "                                            ", t * 1e3,
               (unsigned long long)lanes, lanes / 1e9 / (t * 1e-3), lanes * (with_img ? 480.0 : 320.0) / 1e12 / (t * 1e-3));
      }
    }
    CK(hipFree(buf)); CK(hipFree(img));
  }
  for (uint64_t mb : {800ull, 4800ull, 20000ull}) {       // cfg4's accumulator, cfg4t's, four 5 GB images beside it
    const uint64_t bytes = mb << 20;
    float* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
    const uint64_t rows = bytes / 160;
    const uint32_t lanes = 5000000, steps = 4;
    const double total = (double)lanes * steps * 160.0;
    const float tr = timeit([&] { hipLaunchKernelGGL(k_lane_rows<false>, dim3(lanes / 256), dim3(256), 0, 0, buf, rows, steps, sink); }, 5);
    const float tw = timeit([&] { hipLaunchKernelGGL(k_lane_rows<true>, dim3(lanes / 256), dim3(256), 0, 0, buf, rows, steps, sink); }, 5);
    printf("per-lane 160-B rows at hashed positions, footprint %5llu MB: read %5.2f TB/s (%.1f G rows/s)   read-modify-write %5.2f TB/s r+w (%.1f G rows/s)\n",
           (unsigned long long)mb, total / 1e12 / (tr * 1e-3), lanes * (double)steps / 1e9 / (tr * 1e-3), 2.0 * total / 1e12 / (tw * 1e-3),
           lanes * (double)steps / 1e9 / (tw * 1e-3));
    CK(hipFree(buf));
  }
  return 0;
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
  if (argc > 1 && argv[1][0] == 'l') return lane_main();
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
