#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6q
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6q/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6q/pytest.log
tail -5 gpurun_out/r6q/pytest.log
