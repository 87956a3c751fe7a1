#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6g
bash tools/r6_f.sh 2>&1 | tail -60
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6g/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6g/pytest.log
tail -8 gpurun_out/r6g/pytest.log
