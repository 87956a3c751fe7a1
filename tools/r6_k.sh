#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6k
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6k/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6k/pytest.log
tail -5 gpurun_out/r6k/pytest.log
( timeout 900 python tools/dump_pixel_stream.py cfg5 8 /tmp/cfg5_stream.bin && timeout 600 tools/stream_bench p /tmp/cfg5_stream.bin ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r6k/replay_cfg5.txt
rm -f /tmp/cfg5_stream.bin
bash tools/pmc_sq_medium.sh r6k/m10k 100 50 2>&1 | tee gpurun_out/r6k/sq_m10k.txt
bash tools/pmc_sq_medium.sh r6k/m900 30 15 2>&1 | tee gpurun_out/r6k/sq_m900.txt
