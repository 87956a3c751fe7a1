// fusion_pair.hip -- the two-views-per-launch instances of the triangle-order fusion kernel (smesh_fuse_views; see fusion.hip and
// DESIGN.md 3.0).  A translation unit of its own: the 36 instances take as long to compile as the rest of fusion.hip.
#include <hip/hip_runtime.h>

#include "common.hpp"

#include <cmath>
#include <type_traits>

using namespace smesh;

namespace {

#include "fuse_tri.inc.hpp"

}  // namespace

// `tri_ct`: the class-count slot smesh_aggregator_fuse_triangles chose (exact instances 5 / 13 / 19 / 20 / 21 / 40; run-time-C
// instances sized 8, 16, 24, 32, 40 [slot 41], 48).  One wave per workgroup; `grid` as for the one-view kernel.
void smesh_launch_fuse_tri_pair(int kind, int tri_ct, dim3 grid, hipStream_t st, const TriFuseArgs& t, const TriFuseArgs& tb) {
  const dim3 block(kWave);
#define SMESH_FT2(K)                                                                                  \
  switch (tri_ct) {                                                                                   \
    case 5:  hipLaunchKernelGGL((k_fuse_tri<5, K, true, 2>), grid, block, 0, st, t, tb); break;      \
    case 13: hipLaunchKernelGGL((k_fuse_tri<13, K, true, 2>), grid, block, 0, st, t, tb); break;     \
    case 19: hipLaunchKernelGGL((k_fuse_tri<19, K, true, 2>), grid, block, 0, st, t, tb); break;     \
    case 20: hipLaunchKernelGGL((k_fuse_tri<20, K, true, 2>), grid, block, 0, st, t, tb); break;     \
    case 21: hipLaunchKernelGGL((k_fuse_tri<21, K, true, 2>), grid, block, 0, st, t, tb); break;     \
    case 40: hipLaunchKernelGGL((k_fuse_tri<40, K, true, 2>), grid, block, 0, st, t, tb); break;     \
    case 8:  hipLaunchKernelGGL((k_fuse_tri<8, K, false, 2>), grid, block, 0, st, t, tb); break;     \
    case 16: hipLaunchKernelGGL((k_fuse_tri<16, K, false, 2>), grid, block, 0, st, t, tb); break;    \
    case 24: hipLaunchKernelGGL((k_fuse_tri<24, K, false, 2>), grid, block, 0, st, t, tb); break;    \
    case 32: hipLaunchKernelGGL((k_fuse_tri<32, K, false, 2>), grid, block, 0, st, t, tb); break;    \
    case 41: hipLaunchKernelGGL((k_fuse_tri<40, K, false, 2>), grid, block, 0, st, t, tb); break;    \
    default: hipLaunchKernelGGL((k_fuse_tri<48, K, false, 2>), grid, block, 0, st, t, tb); break;    \
  }
  switch (kind) {
    case SMESH_AGG_SUM: SMESH_FT2(SMESH_AGG_SUM); break;
    case SMESH_AGG_SUMMAX: SMESH_FT2(SMESH_AGG_SUMMAX); break;
    default: SMESH_FT2(SMESH_AGG_MUL); break;
  }
#undef SMESH_FT2
}
