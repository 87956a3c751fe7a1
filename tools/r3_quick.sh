cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
timeout 1200 python -m pytest tests/test_gpu_image_records.py tests/test_gpu_sharded.py -q -m gpu --maxfail=5 --durations=5 2>&1 | tail -12
echo "--- foreign images, cfg2 geometry"; python tools/generic_add_bench.py cfg2 16 2>&1 | grep add
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/quick/kt -o gab -- python tools/generic_add_bench.py cfg2 16 > gpurun_out/quick/gab.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/quick/kt/gab_kernel_stats.csv")):
    n=r["Name"]
    if "k_rec" in n or "k_fuse" in n or "sparse" in n:
        print("  %-70s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
