tag=$1; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp semantic_meshes_amd/csrc/raster.hip /tmp/raster_orig.hip
for w in 2 4 16; do
sed "s/constexpr int kQSub = 8; /constexpr int kQSub = $w; /" /tmp/raster_orig.hip > semantic_meshes_amd/csrc/raster.hip
make -C semantic_meshes_amd/csrc > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/kt$w -o bench -- python bench.py --no-cpu-baseline --steps 100 > gpurun_out/$tag/b$w.log 2>&1
echo -n "kQSub=$w "; grep -o '"value": [0-9.]*' gpurun_out/$tag/b$w.log | head -1
python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/$tag/kt$w/bench_kernel_stats.csv")):
    if "k_raster_frag" in r["Name"] or "k_tile_resolve" in r["Name"]: print("   %s avg %.1f us" % (r["Name"][23:40], float(r["AverageNs"])/1e3))
PY
done
cp /tmp/raster_orig.hip semantic_meshes_amd/csrc/raster.hip
