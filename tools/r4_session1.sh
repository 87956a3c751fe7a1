#!/bin/bash
# Round-4 GPU session 1: whole -m gpu suite, then the measurement batch (tools/r4_bench_all.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests -q -m gpu --maxfail=12 --durations=8 -o faulthandler_timeout=300 2>&1 | tail -150 > gpurun_out/suite/suite.log
grep -E "^FAILED|^ERROR|passed|failed|^E  |Timeout" gpurun_out/suite/suite.log | head -40
timeout 1200 bash tools/r4_bench_all.sh 2>&1 | tail -60
