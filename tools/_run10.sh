tag=$1; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt4 -o c4 -- python tools/big_configs.py cfg4 > gpurun_out/$tag/cfg4.log 2>&1; grep -v "rocprof\|Opened" gpurun_out/$tag/cfg4.log | tail -7
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kt4/c4_kernel_stats.csv")):
    n=r["Name"]
    print("  %-60s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:60],r["Calls"],float(r["AverageNs"])/1e3))
PY
