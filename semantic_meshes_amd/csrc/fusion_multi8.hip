// fusion_multi8.hip -- the eight-views-per-launch instances of the triangle-order fusion kernel k_fuse_tri (class counts up to 40:
// exact instances 5 / 13 / 19 / 20 / 21 / 40, run-time-C instances sized 8 / 16 / 24 / 32 / 40).  A translation unit of its own so that the
// instance sets compile in parallel (see fusion_pair.hip).
#include <hip/hip_runtime.h>

#include "common.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>

using namespace smesh;

namespace {

#include "fuse_tri.inc.hpp"

}  // namespace

void smesh_launch_fuse_tri_8(int kind, int tri_ct, dim3 grid, hipStream_t st, const TriFuseArgs& t, const TriViews<8>& tv) {
  const dim3 block(kWave);
  // Group pipeline (the default since round 5; SMESH_GROUP_PIPELINE=0 / smesh_set_option turn it off; raster.hip): the rasteriser of the next group runs beside this launch, and the one-wave
  // workgroups of this kernel would take every wave slot that frees up.  An LDS pad caps them at five per CU -- alone the kernel
  // loses 2 % that way (it is bound by the memory system, not by occupancy) -- and leaves the rest of the register file to the
  // rasteriser's waves (NOTES/round2.md).  SMESH_FUSE_LDS_PAD overrides (bytes).
  static const int pad_env = getenv("SMESH_FUSE_LDS_PAD") ? atoi(getenv("SMESH_FUSE_LDS_PAD")) : -1;
  const unsigned pad = pad_env >= 0 ? (unsigned)pad_env : t.lds_pad;      // (decided per launch: smesh_aggregator_fuse_triangles)
  TriViews<8> vn;
  for (int v = 0; v < 8; v++) vn.v[v] = tv.v[v];
#define SMESH_FTN(K)                                                                                     \
  switch (tri_ct) {                                                                                      \
    case 5:  hipLaunchKernelGGL((k_fuse_tri<5, K, true, 8>), grid, block, pad, st, t, vn); break;          \
    case 13: hipLaunchKernelGGL((k_fuse_tri<13, K, true, 8>), grid, block, pad, st, t, vn); break;         \
    case 19: hipLaunchKernelGGL((k_fuse_tri<19, K, true, 8>), grid, block, pad, st, t, vn); break;         \
    case 20: hipLaunchKernelGGL((k_fuse_tri<20, K, true, 8>), grid, block, pad, st, t, vn); break;         \
    case 21: hipLaunchKernelGGL((k_fuse_tri<21, K, true, 8>), grid, block, pad, st, t, vn); break;         \
    case 40: hipLaunchKernelGGL((k_fuse_tri<40, K, true, 8>), grid, block, pad, st, t, vn); break;         \
    case 32: hipLaunchKernelGGL((k_fuse_tri<32, K, false, 8>), grid, block, pad, st, t, vn); break;        \
    case 41: hipLaunchKernelGGL((k_fuse_tri<40, K, false, 8>), grid, block, pad, st, t, vn); break;        \
    case 8:  hipLaunchKernelGGL((k_fuse_tri<8, K, false, 8>), grid, block, pad, st, t, vn); break;         \
    case 16: hipLaunchKernelGGL((k_fuse_tri<16, K, false, 8>), grid, block, pad, st, t, vn); break;        \
    default: hipLaunchKernelGGL((k_fuse_tri<24, K, false, 8>), grid, block, pad, st, t, vn); break;        \
  }
  switch (kind) {
    case SMESH_AGG_SUM: SMESH_FTN(SMESH_AGG_SUM); break;
    case SMESH_AGG_SUMMAX: SMESH_FTN(SMESH_AGG_SUMMAX); break;
    default: SMESH_FTN(SMESH_AGG_MUL); break;
  }
#undef SMESH_FTN
}
