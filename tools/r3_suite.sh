# GPU session: the whole -m gpu suite (a Python traceback is dumped if a test sits still for five minutes)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 2700 python -m pytest tests -q -m gpu --maxfail=12 --durations=8 -o faulthandler_timeout=300 2>&1 | tail -120 > gpurun_out/suite/suite.log
grep -E "^FAILED|^ERROR|passed|failed|^E  |Timeout|File \"" gpurun_out/suite/suite.log | head -60
grep -n "slowest" -A 9 gpurun_out/suite/suite.log
