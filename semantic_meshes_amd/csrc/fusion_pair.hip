// fusion_pair.hip -- the two-views-per-launch instances of the triangle-order fusion kernel (smesh_fuse_views; see fusion.hip and
// DESIGN.md 3.2).  A translation unit of its own: the 36 instances take as long to compile as the rest of fusion.hip.
// (fusion_multi4.hip / fusion_multi8.hip: the four- and eight-view instances for class counts up to 24.)
#include <hip/hip_runtime.h>

#include "common.hpp"

#include <cmath>
#include <type_traits>

using namespace smesh;

namespace {

#include "fuse_tri.inc.hpp"

}  // namespace

void smesh_launch_fuse_tri_4(int kind, int tri_ct, dim3 grid, hipStream_t st, const TriFuseArgs& t, const TriViews<8>& tv);
void smesh_launch_fuse_tri_8(int kind, int tri_ct, dim3 grid, hipStream_t st, const TriFuseArgs& t, const TriViews<8>& tv);

// `tri_ct`: the class-count slot smesh_aggregator_fuse_triangles chose (exact instances 5 / 13 / 19 / 20 / 21 / 40; run-time-C
// instances sized 8, 16, 24, 32, 40 [slot 41], 48).  One wave per workgroup; `grid` as for the one-view kernel.  `nviews`: 2, 4 or 8
// (4 and 8: class counts up to 24 only), the first `nviews` entries of `tv`.
void smesh_launch_fuse_tri_multi(int kind, int tri_ct, int nviews, dim3 grid, hipStream_t st, const TriFuseArgs& t, const TriViews<8>& tv) {
  if (nviews == 8) { smesh_launch_fuse_tri_8(kind, tri_ct, grid, st, t, tv); return; }
  if (nviews == 4) { smesh_launch_fuse_tri_4(kind, tri_ct, grid, st, t, tv); return; }
  const dim3 block(kWave);
  TriViews<2> v2;
  v2.v[0] = tv.v[0]; v2.v[1] = tv.v[1];
#define SMESH_FT2(K)                                                                                  \
  switch (tri_ct) {                                                                                   \
    case 5:  hipLaunchKernelGGL((k_fuse_tri<5, K, true, 2>), grid, block, 0, st, t, v2); break;      \
    case 13: hipLaunchKernelGGL((k_fuse_tri<13, K, true, 2>), grid, block, 0, st, t, v2); break;     \
    case 19: hipLaunchKernelGGL((k_fuse_tri<19, K, true, 2>), grid, block, 0, st, t, v2); break;     \
    case 20: hipLaunchKernelGGL((k_fuse_tri<20, K, true, 2>), grid, block, 0, st, t, v2); break;     \
    case 21: hipLaunchKernelGGL((k_fuse_tri<21, K, true, 2>), grid, block, 0, st, t, v2); break;     \
    case 40: hipLaunchKernelGGL((k_fuse_tri<40, K, true, 2>), grid, block, 0, st, t, v2); break;     \
    case 8:  hipLaunchKernelGGL((k_fuse_tri<8, K, false, 2>), grid, block, 0, st, t, v2); break;     \
    case 16: hipLaunchKernelGGL((k_fuse_tri<16, K, false, 2>), grid, block, 0, st, t, v2); break;    \
    case 24: hipLaunchKernelGGL((k_fuse_tri<24, K, false, 2>), grid, block, 0, st, t, v2); break;    \
    case 32: hipLaunchKernelGGL((k_fuse_tri<32, K, false, 2>), grid, block, 0, st, t, v2); break;    \
    case 41: hipLaunchKernelGGL((k_fuse_tri<40, K, false, 2>), grid, block, 0, st, t, v2); break;    \
    default: hipLaunchKernelGGL((k_fuse_tri<48, K, false, 2>), grid, block, 0, st, t, v2); break;    \
  }
  switch (kind) {
    case SMESH_AGG_SUM: SMESH_FT2(SMESH_AGG_SUM); break;
    case SMESH_AGG_SUMMAX: SMESH_FT2(SMESH_AGG_SUMMAX); break;
    default: SMESH_FT2(SMESH_AGG_MUL); break;
  }
#undef SMESH_FT2
}
