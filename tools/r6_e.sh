#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6e
R6_PMC_MEMPIPE=1 bash tools/r6_pmc_workload.sh cfg2 r6e/pmc_cfg2 > gpurun_out/r6e/pmc_cfg2.txt 2>&1
grep -A1 "raster_frag_group\|tile_resolve_group\|project_vertices_group\|k_fuse_tri<19, 0, true, 8>" gpurun_out/r6e/pmc_cfg2.txt
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6e/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6e/pytest.log
tail -8 gpurun_out/r6e/pytest.log
