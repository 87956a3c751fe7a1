tag=$1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "render or fuzz or soup or overlap or overflow or watertight or cfg2" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/$tag
for w in cfg2 cfg4; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt_$w -o bench -- python bench.py --workload $w --no-cpu-baseline --no-host-path --steps 100 > gpurun_out/$tag/bench_$w.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/$tag/bench_$w.log | head -1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kt_$w/bench_kernel_stats.csv")):
    n=r["Name"]
    if "group" in n or "fuse_t" in n: print("  %-60s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:60],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
python bench.py --no-cpu-baseline --no-host-path 2>&1 | grep -o '"value": [0-9.]*'
