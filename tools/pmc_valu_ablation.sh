# NEEDS an ablation build of the library: make -C semantic_meshes_amd/csrc clean && make -C semantic_meshes_amd/csrc -j8 ABLATION=1
# VALU instructions per wave of k_raster_frag_group under the SMESH_RDBG ablation bits (instruction counts add up; times do not).
tag=$1; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 2 4 24 8 1; do
SMESH_RDBG=$d timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/v$d -o b -- python bench.py --steps 16 --warmup 8 --no-cpu-baseline --no-host-path > $out/v$d.log 2>&1
D=$d OUT=$out python - <<'PY'
import csv, collections, os
acc = collections.defaultdict(float)
for r in csv.DictReader(open(os.environ["OUT"] + "/v%s/b_counter_collection.csv" % os.environ["D"])):
    if "k_raster_frag_group" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
w = acc["SQ_WAVES"] or 1
print("rdbg %2s: per wave VALU %6.1f SALU %6.1f VMEM_RD %5.1f VMEM_WR %5.1f" % (os.environ["D"], acc["SQ_INSTS_VALU"] / w, acc["SQ_INSTS_SALU"] / w, acc["SQ_INSTS_VMEM_RD"] / w, acc["SQ_INSTS_VMEM_WR"] / w))
PY
done
