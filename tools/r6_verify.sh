#!/bin/bash
# round 6: the verification batch behind a rasteriser / fusion change -- the whole GPU suite, the cfg2 kernels, fresh differential sweeps
# (also with fragment queues of 48 slots, which overflow in most views).  usage: bash tools/r6_verify.sh [seed offset]
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6v
off=${1:-0}
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6v/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6v/pytest.log; tail -3 gpurun_out/r6v/pytest.log
bash tools/r6_ab.sh r6v/ab cfg2 "SMESH_X=1" 2>&1 | grep -E "raster_frag_group|resolve_group|project_vertices_group|k_fuse_tri<19, 0, true, 8>|pipelined"
bash tools/r6_sweeps.sh 100 1500 $((800000 + off))
echo "== fuse_views soups, 6000 seeds"
SMESH_SWEEP_ONLY=fuse_views timeout 900 python tools/soup_sweep.py $((900000 + off)) 6000 2>&1 | grep -v amdgpu | tail -3 | cut -c1-250
echo "== all generators with queues of 48 slots (SMESH_FRAG_CAP=48), 1200 seeds"
SMESH_FRAG_CAP=48 timeout 900 python tools/soup_sweep.py $((950000 + off)) 1200 2>&1 | grep -v amdgpu | tail -3 | cut -c1-250
