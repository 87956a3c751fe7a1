tag=$1; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 8 24; do
SMESH_RDBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt$d -o bench -- python bench.py --no-cpu-baseline --steps 60 > gpurun_out/$tag/b$d.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kt$d/bench_kernel_stats.csv")):
    if "k_raster_frag" in r["Name"]: print("RDBG=$d  k_raster_frag avg %.1f us" % (float(r["AverageNs"])/1e3))
PY
done
