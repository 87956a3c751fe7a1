// dpp_test.hip -- checks the DPP row_shl semantics assumed by fusion.hip: lane i reads lane i+D of its 16-lane row, 0 beyond.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int D> __device__ int row_down(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x100 + D, 0xF, 0xF, true); }
__global__ void k(int* out) {
  const int l = threadIdx.x;
  out[l] = row_down<1>(l + 100);
  out[64 + l] = row_down<4>(l + 100);
  out[128 + l] = row_down<8>(l + 100);
}
int main() {
  int* d; hipMalloc(&d, 192 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  const int D[3] = {1, 4, 8};
  for (int t = 0; t < 3; t++) for (int l = 0; l < 64; l++) {
    const int want = ((l & 15) + D[t] < 16) ? l + D[t] + 100 : 0;
    if (h[t * 64 + l] != want) { if (bad < 8) printf("D=%d lane %d got %d want %d\n", D[t], l, h[t * 64 + l], want); bad++; }
  }
  printf(bad ? "DPP TEST FAILED (%d)\n" : "dpp ok\n", bad);
  return bad != 0;
}
