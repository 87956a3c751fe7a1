#!/bin/bash
# round 6, first GPU call: the new launcher test, the rasteriser's counters at cfg2 (baseline), a serialised kernel trace
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "without_a_launcher or two_rank_run" > gpurun_out/r6a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6a/pytest.log
bash tools/r6_pmc_workload.sh cfg2 r6a/pmc_cfg2 > gpurun_out/r6a/pmc_cfg2.txt 2>&1
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r6a/trace -o t -- python $root/bench.py --steps 64 --warmup 8 --repeats 1 --no-cpu-baseline --no-host-path --no-pmc --no-group-pipeline > $root/gpurun_out/r6a/trace.log 2>&1 )
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/r6a/trace/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-60s calls %6s avg %10.1f ns  total %5.1f%%" % (r['Name'].replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['AverageNs']), float(r['Percentage'])))
PY
tail -3 gpurun_out/r6a/pytest.log
