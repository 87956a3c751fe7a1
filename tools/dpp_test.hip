// dpp_test.hip -- checks the DPP row_shl semantics assumed by fusion.hip: lane i reads lane i+D of its 16-lane row, 0 beyond.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int D> __device__ int row_down(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x100 + D, 0xF, 0xF, true); }
// wave64 inclusive scan used by raster.hip (row_shr 1,2,4,8 + row_bcast:15 / row_bcast:31)
__device__ int wave_scan(int v) {
  int s = v;
  s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xF, 0xF, true);
  s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xA, 0xF, false);
  s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xC, 0xF, false);
  return s;
}
__global__ void k_scan(int* out) {
  const int l = threadIdx.x;
  out[l] = wave_scan((l * 7 + 3) % 11);
  out[64 + l] = wave_scan((l % 5 == 0) ? l : 0);
}
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
  hipLaunchKernelGGL(k_scan, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 128 * 4, hipMemcpyDeviceToHost);
  int r0 = 0, r1 = 0;
  for (int l = 0; l < 64; l++) {
    r0 += (l * 7 + 3) % 11; r1 += (l % 5 == 0) ? l : 0;
    if (h[l] != r0 || h[64 + l] != r1) { if (bad < 16) printf("scan lane %d got %d/%d want %d/%d\n", l, h[l], h[64 + l], r0, r1); bad++; }
  }
  printf(bad ? "DPP TEST FAILED (%d)\n" : "dpp ok\n", bad);
  return bad != 0;
}
