// alloc_churn_repro.hip -- stand-alone reproducer (no libsmesh) of what tests/flake_hunt.py isolated in round 5:
// on a GPU shared by several PROCESSES, a kernel's writes to a buffer that hipMalloc has just handed out again after a hipFree of
// the same size are, now and then, missing for every workgroup that ran on ONE of the eight XCDs (the buffer keeps the zeros it
// came with).  One process alone never shows it; a buffer that is allocated once and reused never shows it.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/alloc_churn_repro tools/alloc_churn_repro.hip
//   for i in 0 1 2 3 4 5 6 7; do ./tools/alloc_churn_repro 4000 $i & done; wait          # churn: fresh allocation per iteration
//   for i in 0 1 2 3 4 5 6 7; do ./tools/alloc_churn_repro 4000 $i reuse & done; wait    # control: one allocation, reused
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void k_fill(uint32_t* out, size_t n, uint32_t tag) {      // 256 threads x 19 words each: the shape of the generator kernel
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 19 >= n) return;
  for (int c = 0; c < 19; c++) out[i * 19 + c] = tag ^ (uint32_t)(i * 19 + c) ^ 0x9E3779B9u;
}
__global__ void k_read(const uint32_t* in, size_t n, unsigned long long* sum) {   // every XCD reads the buffer (fills its TLBs) before it is freed
  unsigned long long s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += in[i];
  if (s == 0x123456789ull) *sum = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  const int id = argc > 2 ? atoi(argv[2]) : 0;
  const bool reuse = argc > 3 && !strcmp(argv[3], "reuse");
  const bool late = argc > 3 && !strcmp(argv[3], "late");      // fresh allocation per iteration, freed 64 iterations later (no address is handed out again at once)
  std::vector<uint32_t*> ring(64, nullptr);
  const int extra = argc > 4 ? atoi(argv[4]) : 0;      // further streams of this process that are kept busy (libsmesh has three, one of them with priority)
  const size_t pixels = 320 * 240, n = pixels * 19, bytes = n * 4;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long* d_sum;
  CK(hipMalloc(reinterpret_cast<void**>(&d_sum), 8));
  std::vector<hipStream_t> xs((size_t)extra);
  uint32_t* xbuf = nullptr;
  if (extra) CK(hipMalloc(reinterpret_cast<void**>(&xbuf), bytes));
  for (int i = 0; i < extra; i++) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (i == 0) CK(hipStreamCreateWithPriority(&xs[(size_t)i], hipStreamNonBlocking, greatest));
    else CK(hipStreamCreateWithFlags(&xs[(size_t)i], hipStreamNonBlocking));
  }
  std::vector<uint32_t> host(n);
  uint32_t* kept = nullptr;
  if (reuse) CK(hipMalloc(reinterpret_cast<void**>(&kept), bytes));
  int bad_iters = 0;
  for (int it = 0; it < iters; it++) {
    uint32_t* buf = kept;
    void* small[2] = {nullptr, nullptr};
    CK(hipMalloc(&small[0], 273600));                       // (an aggregator's accumulator and histogram are allocated in between)
    CK(hipMalloc(&small[1], 14400));
    CK(hipMemsetAsync(small[0], 0, 273600, st));
    if (!reuse) CK(hipMalloc(reinterpret_cast<void**>(&buf), bytes));
    else CK(hipMemsetAsync(buf, 0, bytes, st));
    for (int i = 0; i < extra; i++) hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, xs[(size_t)i], xbuf, n, d_sum);
    const uint32_t tag = (uint32_t)(it * 2654435761u) | 1u;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, st, buf, n, tag);
    hipLaunchKernelGGL(k_read, dim3(1024), dim3(256), 0, st, buf, n, d_sum);
    CK(hipMemcpyAsync(host.data(), buf, bytes, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
    size_t wrong = 0, zero = 0, first = 0;
    unsigned chunk_mask = 0;
    for (size_t i = 0; i < n; i++) {
      const uint32_t want = tag ^ (uint32_t)i ^ 0x9E3779B9u;
      if (host[i] != want) {
        if (!wrong) first = i;
        wrong++;
        zero += host[i] == 0;
        chunk_mask |= 1u << ((i / 19 / 256) % 8);
      }
    }
    if (wrong) {
      bad_iters++;
      printf("process %d iteration %d: %zu of %zu words wrong (%zu of them zero), first at word %zu; workgroup numbers mod 8 (bit mask) 0x%02x; buffer %p\n",
             id, it, wrong, n, zero, first, chunk_mask, (void*)buf);
      fflush(stdout);
    }
    CK(hipStreamSynchronize(st));
    for (int i = 0; i < extra; i++) CK(hipStreamSynchronize(xs[(size_t)i]));
    if (late) { if (ring[(size_t)it % 64]) CK(hipFree(ring[(size_t)it % 64])); ring[(size_t)it % 64] = buf; }
    else if (!reuse) CK(hipFree(buf));
    CK(hipFree(small[0]));
    CK(hipFree(small[1]));
  }
  printf("process %d (%s): %d iterations, %d with lost writes\n", id, reuse ? "one allocation reused" : late ? "fresh allocation per iteration, freed 64 iterations later" : "fresh allocation per iteration", iters, bad_iters);
  return bad_iters ? 1 : 0;
}
