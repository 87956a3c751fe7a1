# usage: bash tools/kernel_stats.sh <tag> [test]   -- bench.py under rocprofv3 --kernel-trace --stats, per-kernel averages; "test" runs the GPU test suite first
tag=$1; mkdir -p gpurun_out/$tag
if [ "$2" = "test" ]; then timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; tail -3 gpurun_out/$tag/pytest.log; fi
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt -o bench -- python bench.py --no-cpu-baseline --steps 100 > gpurun_out/$tag/bench_kt.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/$tag/bench_kt.log | head -1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kt/bench_kernel_stats.csv")):
    n=r["Name"]
    if "synth" in n or "rocclr" in n: continue
    print("  %-60s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:60],r["Calls"],float(r["AverageNs"])/1e3))
PY
