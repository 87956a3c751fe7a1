tag=$1; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; tail -3 gpurun_out/$tag/pytest.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in 1 0; do
SMESH_CULL=$c timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt$c -o bench -- python bench.py --no-cpu-baseline --steps 100 > gpurun_out/$tag/bench_kt$c.log 2>&1
echo "cull=$c"; grep -o '"value": [0-9.]*' gpurun_out/$tag/bench_kt$c.log | head -1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kt$c/bench_kernel_stats.csv")):
    n=r["Name"]
    if "synth" in n or "rocclr" in n or "finalize" in n or "fill" in n: continue
    print("  %-60s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:60],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
