# usage (on the GPU box): bash tools/run_generic_add_bench.sh [cfg] -- add() on foreign images: image records vs scatter, kernel breakdown
cfg=${1:-cfg2}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gab
python tools/generic_add_bench.py $cfg 16 2>&1 | grep add
SMESH_ADD_RECORDS=0 python tools/generic_add_bench.py $cfg 16 2>&1 | grep add
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gab/kt -o gab -- python tools/generic_add_bench.py $cfg 16 > gpurun_out/gab/gab.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/gab/kt/gab_kernel_stats.csv")):
    n=r["Name"]
    if "synth" in n: continue
    print("  %-70s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
