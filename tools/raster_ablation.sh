# NEEDS an ablation build of the library: make -C semantic_meshes_amd/csrc clean && make -C semantic_meshes_amd/csrc -j8 ABLATION=1
# k_raster_frag_group under the SMESH_RDBG ablation bits (development; results are wrong by design when a bit is set).
# usage: bash tools/raster_ablation.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 2 4 24 8 1; do
  SMESH_RDBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ab$d -o b -- python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-host-path > $out/ab$d.log 2>&1
  python - <<PY
import csv
for r in csv.DictReader(open("$out/ab$d/b_kernel_stats.csv")):
    if "raster_frag_group" in r["Name"] or "tile_resolve_group" in r["Name"]: print("rdbg $d  %-40s avg %8.1f us"%(r["Name"][:40], float(r["AverageNs"])/1e3))
PY
done
