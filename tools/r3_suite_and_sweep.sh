# GPU session: class-count sweep of add() on foreign images (records from moments / scatter-add), then the whole -m gpu suite
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sweep
( echo "records (moments):"; SMESH_ADD_RECORDS_MIN_C=0 python tools/generic_add_sweep.py 2 3 5 8 13 19 27 32 40 48 64 100 150 2>&1 | tail -1
  echo "scatter-add:"; SMESH_ADD_RECORDS=0 python tools/generic_add_sweep.py 2 3 5 8 13 19 27 32 2>&1 | tail -1 ) > gpurun_out/sweep/class_sweep.txt
cat gpurun_out/sweep/class_sweep.txt
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/sweep/suite.log
cat gpurun_out/sweep/suite.log
