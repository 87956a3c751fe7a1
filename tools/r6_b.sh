#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6b
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py -x -q -m gpu > gpurun_out/r6b/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6b/pytest.log
tail -5 gpurun_out/r6b/pytest.log
bash tools/r6_ab.sh r6b/ab cfg2 "SMESH_RASTER_XCD=0 SMESH_RASTER_SKIP_PLANE=0" "SMESH_RASTER_XCD=1 SMESH_RASTER_SKIP_PLANE=0" "SMESH_RASTER_XCD=1 SMESH_RASTER_SKIP_PLANE=1" 2>&1 | tee gpurun_out/r6b/ab.txt
