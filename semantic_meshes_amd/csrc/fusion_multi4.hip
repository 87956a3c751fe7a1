// fusion_multi4.hip -- the four-views-per-launch instances of the triangle-order fusion kernel k_fuse_tri (class counts up to 40:
// exact instances 5 / 13 / 19 / 20 / 21 / 40, run-time-C instances sized 8 / 16 / 24 / 32 / 40).  A translation unit of its own so that the
// instance sets compile in parallel (see fusion_pair.hip).
#include <hip/hip_runtime.h>

#include "common.hpp"

#include <cmath>
#include <type_traits>

using namespace smesh;

namespace {

#include "fuse_tri.inc.hpp"

}  // namespace

void smesh_launch_fuse_tri_4(int kind, int tri_ct, dim3 grid, hipStream_t st, const TriFuseArgs& t, const TriViews<8>& tv) {
  const dim3 block(kWave);
  TriViews<4> vn;
  for (int v = 0; v < 4; v++) vn.v[v] = tv.v[v];
#define SMESH_FTN(K)                                                                                     \
  switch (tri_ct) {                                                                                      \
    case 5:  hipLaunchKernelGGL((k_fuse_tri<5, K, true, 4>), grid, block, 0, st, t, vn); break;          \
    case 13: hipLaunchKernelGGL((k_fuse_tri<13, K, true, 4>), grid, block, 0, st, t, vn); break;         \
    case 19: hipLaunchKernelGGL((k_fuse_tri<19, K, true, 4>), grid, block, 0, st, t, vn); break;         \
    case 20: hipLaunchKernelGGL((k_fuse_tri<20, K, true, 4>), grid, block, 0, st, t, vn); break;         \
    case 21: hipLaunchKernelGGL((k_fuse_tri<21, K, true, 4>), grid, block, 0, st, t, vn); break;         \
    case 40: hipLaunchKernelGGL((k_fuse_tri<40, K, true, 4>), grid, block, 0, st, t, vn); break;         \
    case 32: hipLaunchKernelGGL((k_fuse_tri<32, K, false, 4>), grid, block, 0, st, t, vn); break;        \
    case 41: hipLaunchKernelGGL((k_fuse_tri<40, K, false, 4>), grid, block, 0, st, t, vn); break;        \
    case 8:  hipLaunchKernelGGL((k_fuse_tri<8, K, false, 4>), grid, block, 0, st, t, vn); break;         \
    case 16: hipLaunchKernelGGL((k_fuse_tri<16, K, false, 4>), grid, block, 0, st, t, vn); break;        \
    default: hipLaunchKernelGGL((k_fuse_tri<24, K, false, 4>), grid, block, 0, st, t, vn); break;        \
  }
  switch (kind) {
    case SMESH_AGG_SUM: SMESH_FTN(SMESH_AGG_SUM); break;
    case SMESH_AGG_SUMMAX: SMESH_FTN(SMESH_AGG_SUMMAX); break;
    default: SMESH_FTN(SMESH_AGG_MUL); break;
  }
#undef SMESH_FTN
}
