#!/bin/bash
# kernel trace of a python tool: bash tools/r6_trace_py.sh <tag> <script and args...>  -> per-kernel calls / average / share
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/$tag; mkdir -p $out
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python "$@" > $out/run.log 2>&1 )
OUT=$out python - <<'PY'
import csv, glob, os
for f in glob.glob(os.environ['OUT'] + '/trace/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r['TotalDurationNs']) for r in rows)
    for r in rows[:12]:
        print("   %-58s calls %5s avg %9.1f us  total %8.1f us (%4.1f%%)" % (r['Name'].replace('(anonymous namespace)::', '')[:58], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
