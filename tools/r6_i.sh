#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6i
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6i/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6i/pytest.log
tail -8 gpurun_out/r6i/pytest.log
bash tools/r6_ab.sh r6i/ab cfg2 "SMESH_RASTER_FUSE_PROJECT=0" "SMESH_RASTER_FUSE_PROJECT=1" 2>&1 | tee gpurun_out/r6i/ab.txt
bash tools/r6_ab.sh r6i/ab4 cfg4 "SMESH_RASTER_FUSE_PROJECT=0" "SMESH_RASTER_FUSE_PROJECT=1" 2>&1 | tee gpurun_out/r6i/ab4.txt
SMESH_RASTER_FUSE_PROJECT=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r6i/pytest_fused.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6i/pytest_fused.log
tail -5 gpurun_out/r6i/pytest_fused.log
