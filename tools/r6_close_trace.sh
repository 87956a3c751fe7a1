#!/bin/bash
# round 6: where an inside-the-scene view's time goes -- kernel trace of tools/close_view_bench.py for one grid size.  usage: bash tools/r6_close_trace.sh 300x150 [ENV=..]
g=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/r6close_$g; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python $root/tools/close_view_bench.py $g > $out/run.log 2>&1 )
grep triangles $out/run.log | cut -c1-220
OUT=$out python - <<'PY'
import csv, glob, os
for f in glob.glob('%s/trace/**/*kernel_stats.csv' % os.environ['OUT'], recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:16]:
        print("   %-60s calls %5s avg %9.1f us  total %8.1f us (%4.1f%%)" % (r['Name'].replace('(anonymous namespace)::', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
