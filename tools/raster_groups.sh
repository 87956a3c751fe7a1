# k_raster_frag_group with 1 / 2 / 4 / 8 groups of 64 triangles per wave (vertex indices of the next group prefetched); cfg2 and cfg4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/rg
for g in 1 2 4 8; do
for w in cfg2 cfg4; do
SMESH_RASTER_GROUPS=$g timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rg/kt_${g}_$w -o bench -- python bench.py --workload $w --no-cpu-baseline --no-host-path --steps 100 > gpurun_out/rg/bench_${g}_$w.log 2>&1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/rg/kt_${g}_$w/bench_kernel_stats.csv")):
    n=r["Name"]
    if "raster_frag_group" in n or "resolve_group" in n: print("groups $g $w  %-40s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:40],float(r["AverageNs"])/1e3))
PY
done
done
