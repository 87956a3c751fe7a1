tag=$1; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; tail -3 gpurun_out/$tag/pytest.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt5 -o c5 -- python tools/big_configs.py cfg5 > gpurun_out/$tag/cfg5.log 2>&1; grep -v rocprof gpurun_out/$tag/cfg5.log | tail -6
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kt5/c5_kernel_stats.csv")):
    n=r["Name"]
    print("  %-60s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:60],r["Calls"],float(r["AverageNs"])/1e3))
PY
