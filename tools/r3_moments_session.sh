# GPU session: image records from moments -- tests, then timing at cfg2 geometry against the old passes and the scatter-add
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mom
timeout 1500 python -m pytest tests/test_gpu_image_records.py tests/test_gpu_parity.py -x -q -m gpu --durations=8 2>&1 | tail -25 > gpurun_out/mom/tests.log
cat gpurun_out/mom/tests.log
export SMESH_ADD_RECORDS_MIN_C=0
echo "--- moments"; python tools/generic_add_bench.py cfg2 16 2>&1 | grep add
for d in 1 2 4 7; do echo "--- moments, SMESH_REC_DBG=$d (timing only)"; SMESH_REC_DBG=$d python tools/generic_add_bench.py cfg2 16 2>&1 | grep add; done
echo "--- passes A/B"; SMESH_REC_MOMENTS=0 python tools/generic_add_bench.py cfg2 16 2>&1 | grep add
echo "--- scatter"; SMESH_ADD_RECORDS=0 python tools/generic_add_bench.py cfg2 16 2>&1 | grep add
for d in 0 1 2 4; do
SMESH_REC_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mom/kt$d -o gab -- python tools/generic_add_bench.py cfg2 16 > gpurun_out/mom/gab$d.log 2>&1
echo "--- kernels, SMESH_REC_DBG=$d"
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/mom/kt$d/gab_kernel_stats.csv")):
    n=r["Name"]
    if "k_rec" in n or "k_fuse" in n or "sparse" in n:
        print("  %-70s calls %4s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
