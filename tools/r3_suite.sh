# GPU session: the whole -m gpu suite, then the foreign-image add() timing
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests -q -m gpu --maxfail=12 --durations=8 2>&1 | tail -80 > gpurun_out/suite/suite.log
grep -E "^FAILED|^ERROR|passed|failed|^E  " gpurun_out/suite/suite.log | head -60
echo "--- foreign images, cfg2 geometry"; python tools/generic_add_bench.py cfg2 16 2>&1 | grep add
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/suite/kt -o gab -- python tools/generic_add_bench.py cfg2 16 > gpurun_out/suite/gab.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/suite/kt/gab_kernel_stats.csv")):
    n=r["Name"]
    if "k_rec" in n or "k_fuse" in n or "sparse" in n:
        print("  %-70s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
