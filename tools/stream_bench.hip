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
             (unsigned long long)gb, stride == 0 ? "hashed positions      " : stride == 1 ? "consecutive rows [CACHE: every wave re-reads what its neighbours just fetched -- L2 / Infinity Cache rate, not HBM]" : "rows 2160 apart (x+1) ",
             total / 1e12 / (t2 * 1e-3), total / 1e12 / (t4 * 1e-3), total / 1e12 / (t8 * 1e-3), total / 1e9);
    }
    CK(hipFree(buf));
  }
  return 0;
}

// ---- replay of a REAL address stream (round 6): what k_fuse_tri_wide reads at cfg5, taken from renders of the scene
// itself (tools/dump_pixel_stream.py): per view the pixels of every triangle in triangle order -- wave w of the kernel owns triangles
// [64 w, 64 w + 64) and reads the 600-byte class vector of each of their visible pixels, view after view, INFLIGHT rows at a time; with
// ACC it also reads and writes back the 64 accumulator rows of its triangles once (those of triangles seen in any view).
//   file: uint32 magic 0x534D5253, nviews, pixels per image, waves; then per view: uint32 start[waves + 1], uint32 row[start[waves]]
template <int INFLIGHT, bool ACC>
__global__ __launch_bounds__(64) void k_replay(const float* __restrict__ img, uint64_t img_floats, const uint32_t* const* __restrict__ start,
                                               const uint32_t* const* __restrict__ row, int nviews, float* __restrict__ acc, float* sink) {
  const int l = threadIdx.x, c = 4 * l;
  const uint32_t w = blockIdx.x;
  float s = 0.f;
  bool any = false;
  for (int v = 0; v < nviews; v++) {
    const uint32_t a = start[v][w], b = start[v][w + 1];
    any = any || b > a;
    const float* __restrict__ im = img + (uint64_t)v * img_floats;
    for (uint32_t i = a; i < b; i += INFLIGHT) {
      fvec4 x[INFLIGHT];
#pragma unroll
      for (int k = 0; k < INFLIGHT; k++) {
        x[k] = fvec4{0.f, 0.f, 0.f, 0.f};
        if (i + k < b && c + 4 <= 150) x[k] = *reinterpret_cast<const fvec4_a4*>(im + (uint64_t)row[v][i + k] * 150u + c);
      }
#pragma unroll
      for (int k = 0; k < INFLIGHT; k++) s += x[k].x + x[k].y + x[k].z + x[k].w;
    }
  }
  if (ACC && any) {
    // the wave's 64 rows are one contiguous 38 400-byte block: 2 400 sixteen-byte pieces, 64 lanes
    fvec4_a4* blk = reinterpret_cast<fvec4_a4*>(acc + (uint64_t)w * 64u * 150u);
    for (int p = l; p < 2400; p += 64) { fvec4 v = blk[p]; v += s; blk[p] = v; }
  }
  if (s == 123.456f) *sink = s;
}

// The same rows in the KERNEL's order: triangle after triangle (a triangle's views in order), INFLIGHT class vectors at a time whatever
// triangle they belong to.  ROWS: 0 = class vectors only; 1 = + the wave's 64-row accumulator block once at the end (as k_replay);
// 2 = + each visible triangle's own 600-byte row read before its first pixel and written after its last (what the kernels do).
template <int INFLIGHT, int ROWS>
__global__ __launch_bounds__(64) void k_replay_merged(const float* __restrict__ img, uint64_t img_floats, const uint32_t* __restrict__ start,
                                                      const uint32_t* __restrict__ ent, const uint32_t* __restrict__ tri, float* __restrict__ acc, float* sink) {
  const int l = threadIdx.x, c = 4 * l;
  const uint32_t w = blockIdx.x;
  const uint32_t a = start[w], b = start[w + 1];
  float s = 0.f;
  uint32_t cur = 0xFFFFFFFFu;
  fvec4 row = fvec4{0.f, 0.f, 0.f, 0.f};
  for (uint32_t i = a; i < b; i += INFLIGHT) {
    fvec4 x[INFLIGHT];
    uint32_t t[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) {
      x[k] = fvec4{0.f, 0.f, 0.f, 0.f};
      t[k] = 0xFFFFFFFFu;
      if (i + k < b) {
        const uint32_t e = ent[i + k];
        t[k] = tri[i + k];
        if (c + 4 <= 150) x[k] = *reinterpret_cast<const fvec4_a4*>(img + (uint64_t)(e >> 29) * img_floats + (uint64_t)(e & 0x1FFFFFFFu) * 150u + c);
      }
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++) {
      if (ROWS == 2 && t[k] != 0xFFFFFFFFu && t[k] != cur) {
        if (cur != 0xFFFFFFFFu && c + 4 <= 150) *reinterpret_cast<fvec4_a4*>(acc + (uint64_t)cur * 150u + c) = row;
        cur = t[k];
        if (c + 4 <= 150) row = *reinterpret_cast<const fvec4_a4*>(acc + (uint64_t)cur * 150u + c);
      }
      row += x[k];
      s += x[k].x + x[k].y + x[k].z + x[k].w;
    }
  }
  if (ROWS == 2 && cur != 0xFFFFFFFFu && c + 4 <= 150) *reinterpret_cast<fvec4_a4*>(acc + (uint64_t)cur * 150u + c) = row;
  if (ROWS == 1 && b > a) {
    fvec4_a4* blk = reinterpret_cast<fvec4_a4*>(acc + (uint64_t)w * 64u * 150u);
    for (int p = l; p < 2400; p += 64) { fvec4 v = blk[p]; v += s; blk[p] = v; }
  }
  if (s == 123.456f) *sink = s;
}

int replay_main(const char* path) {
  FILE* f = fopen(path, "rb");
  if (!f) { printf("cannot open %s\n", path); return 1; }
  uint32_t head[4];
  if (fread(head, 4, 4, f) != 4 || head[0] != 0x534D5253u) { printf("bad stream file\n"); return 1; }
  const int nviews = (int)head[1];
  const uint64_t N = head[2];
  const uint32_t waves = head[3];
  float* img; CK(hipMalloc(&img, (uint64_t)nviews * N * 600)); CK(hipMemset(img, 0, (uint64_t)nviews * N * 600));
  float* acc; CK(hipMalloc(&acc, (uint64_t)waves * 64 * 600)); CK(hipMemset(acc, 0, (uint64_t)waves * 64 * 600));
  float* sink; CK(hipMalloc(&sink, 4));
  const uint32_t* h_start[8]; const uint32_t* h_row[8];
  uint64_t total_rows = 0, seen_waves = 0;
  uint32_t* any_seen = (uint32_t*)calloc(waves, 4);
  for (int v = 0; v < nviews; v++) {
    uint32_t* st = (uint32_t*)malloc((size_t)(waves + 1) * 4);
    if (fread(st, 4, waves + 1, f) != waves + 1) { printf("short file\n"); return 1; }
    const uint32_t n = st[waves];
    uint32_t* rw = (uint32_t*)malloc((size_t)(n ? n : 1) * 4);
    if (fread(rw, 4, n, f) != n) { printf("short file\n"); return 1; }
    for (uint32_t w = 0; w < waves; w++) if (st[w + 1] > st[w]) any_seen[w] = 1;
    uint32_t *d_st, *d_rw;
    CK(hipMalloc(&d_st, (size_t)(waves + 1) * 4)); CK(hipMalloc(&d_rw, (size_t)(n ? n : 1) * 4));
    CK(hipMemcpy(d_st, st, (size_t)(waves + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_rw, rw, (size_t)n * 4, hipMemcpyHostToDevice));
    h_start[v] = d_st; h_row[v] = d_rw; total_rows += n;
    free(st); free(rw);
  }
  // the merged (triangle-major) section, if the file has one
  uint32_t *d_mstart = nullptr, *d_ment = nullptr, *d_mtri = nullptr;
  uint64_t merged_rows = 0, merged_tris = 0;
  {
    uint32_t* st = (uint32_t*)malloc((size_t)(waves + 1) * 4);
    if (fread(st, 4, waves + 1, f) == waves + 1) {
      const uint32_t n = st[waves];
      uint32_t* en = (uint32_t*)malloc((size_t)(n ? n : 1) * 4);
      uint32_t* tr = (uint32_t*)malloc((size_t)(n ? n : 1) * 4);
      if (fread(en, 4, n, f) == n && fread(tr, 4, n, f) == n) {
        CK(hipMalloc(&d_mstart, (size_t)(waves + 1) * 4)); CK(hipMalloc(&d_ment, (size_t)(n ? n : 1) * 4)); CK(hipMalloc(&d_mtri, (size_t)(n ? n : 1) * 4));
        CK(hipMemcpy(d_mstart, st, (size_t)(waves + 1) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_ment, en, (size_t)n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_mtri, tr, (size_t)n * 4, hipMemcpyHostToDevice));
        merged_rows = n;
        for (uint32_t i = 0; i < n; i++) merged_tris += (i == 0 || tr[i] != tr[i - 1]) ? 1 : 0;
      }
      free(en); free(tr);
    }
    free(st);
  }
  fclose(f);
  for (uint32_t w = 0; w < waves; w++) seen_waves += any_seen[w];
  const uint32_t** d_start; const uint32_t** d_row;
  CK(hipMalloc(&d_start, 8 * sizeof(void*))); CK(hipMalloc(&d_row, 8 * sizeof(void*)));
  CK(hipMemcpy(d_start, h_start, nviews * sizeof(void*), hipMemcpyHostToDevice)); CK(hipMemcpy(d_row, h_row, nviews * sizeof(void*), hipMemcpyHostToDevice));
  const double gath = (double)total_rows * 600.0, accb = (double)seen_waves * 64 * 600.0 * 2.0;
  printf("replay of %s: %d views, %llu pixels per image, %u waves of 64 triangles (%llu with a visible triangle), %llu class-vector rows = %.2f GB; their accumulator blocks both ways %.2f GB\n",
         path, nviews, (unsigned long long)N, waves, (unsigned long long)seen_waves, (unsigned long long)total_rows, gath / 1e9, accb / 1e9);
#define REPLAY(B, A) timeit([&] { hipLaunchKernelGGL((k_replay<B, A>), dim3(waves), dim3(64), 0, 0, img, N * 150, d_start, d_row, nviews, acc, sink); }, 3)
  const float g2 = REPLAY(2, false), g4 = REPLAY(4, false), a2 = REPLAY(2, true), a4 = REPLAY(4, true);
  printf("  class vectors only        : 2 in flight %8.3f ms = %5.2f TB/s    4 in flight %8.3f ms = %5.2f TB/s\n", g2, gath / 1e12 / (g2 * 1e-3), g4, gath / 1e12 / (g4 * 1e-3));
  printf("  + accumulator blocks (r+w): 2 in flight %8.3f ms = %5.2f TB/s    4 in flight %8.3f ms = %5.2f TB/s\n", a2, (gath + accb) / 1e12 / (a2 * 1e-3), a4,
         (gath + accb) / 1e12 / (a4 * 1e-3));
  if (d_mstart) {
    const double rowb = (double)merged_tris * 600.0 * 2.0;
    printf("the same %llu rows triangle-major (the kernels' order: a triangle's views one after the other), %llu visible triangles (their rows both ways: %.2f GB)\n",
           (unsigned long long)merged_rows, (unsigned long long)merged_tris, rowb / 1e9);
#define REPLAYM(B, R) timeit([&] { hipLaunchKernelGGL((k_replay_merged<B, R>), dim3(waves), dim3(64), 0, 0, img, N * 150, d_mstart, d_ment, d_mtri, acc, sink); }, 3)
    const float m2 = REPLAYM(2, 0), m4 = REPLAYM(4, 0), m8 = REPLAYM(8, 0), b2 = REPLAYM(2, 1), b4 = REPLAYM(4, 1), r2 = REPLAYM(2, 2), r4 = REPLAYM(4, 2), r8 = REPLAYM(8, 2);
    printf("  class vectors only             : 2 in flight %8.3f ms = %5.2f TB/s    4 in flight %8.3f ms = %5.2f TB/s    8 in flight %8.3f ms = %5.2f TB/s\n",
           m2, gath / 1e12 / (m2 * 1e-3), m4, gath / 1e12 / (m4 * 1e-3), m8, gath / 1e12 / (m8 * 1e-3));
    printf("  + accumulator blocks (r+w)     : 2 in flight %8.3f ms = %5.2f TB/s    4 in flight %8.3f ms = %5.2f TB/s\n", b2, (gath + accb) / 1e12 / (b2 * 1e-3), b4,
           (gath + accb) / 1e12 / (b4 * 1e-3));
    printf("  + each triangle's own row (r+w): 2 in flight %8.3f ms = %5.2f TB/s    4 in flight %8.3f ms = %5.2f TB/s    8 in flight %8.3f ms = %5.2f TB/s\n",
           r2, (gath + rowb) / 1e12 / (r2 * 1e-3), r4, (gath + rowb) / 1e12 / (r4 * 1e-3), r8, (gath + rowb) / 1e12 / (r8 * 1e-3));
  }
  return 0;
}

int main(int argc, char** argv) {
  if (argc > 2 && argv[1][0] == 'p') return replay_main(argv[2]);      // (rePlay)
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
