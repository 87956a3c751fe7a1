#!/bin/bash
# round 6: the checks behind the per-view index-plane decision (RasterArgs::idx_optional == 2) -- the seeds that caught the earlier attempts,
# fresh fuse_views soups (class counts 2 .. 60: k_fuse_tri, its 48-slot instance, k_fuse_tri_any), the same with queues of 48 slots, cfg2 A/B
# against one decision per launch (SMESH_PLANE_LEVEL=1).  usage: bash tools/r6_verify_planes.sh [seed offset]
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6v
off=${1:-0}
timeout 600 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_deferred.py -x -q -m gpu 2>&1 | tail -2
echo "== fuse_views soups, 8000 seeds"
SMESH_SWEEP_ONLY=fuse_views timeout 900 python tools/soup_sweep.py $((1000000 + off)) 8000 2>&1 | grep -v amdgpu | tail -3 | cut -c1-250
echo "== fuse_views soups with queues of 48 slots (SMESH_FRAG_CAP=48), 3000 seeds"
SMESH_FRAG_CAP=48 SMESH_SWEEP_ONLY=fuse_views timeout 900 python tools/soup_sweep.py $((1100000 + off)) 3000 2>&1 | grep -v amdgpu | tail -3 | cut -c1-250
echo "== the two seeds of the first attempt"
SMESH_SWEEP_ONLY=fuse_views timeout 300 python tools/soup_sweep.py 553071 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
SMESH_FRAG_CAP=48 SMESH_SWEEP_ONLY=fuse_views timeout 300 python tools/soup_sweep.py 702926 1 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200
bash tools/r6_ab.sh r6v/ab cfg2 "SMESH_PLANE_LEVEL=1" "SMESH_X=1" 2>&1 | grep -E "^==|raster_frag_group|resolve_group|k_fuse_tri<19, 0, true, 8>|pipelined|views/s"
