# usage: bash tools/pipeline_timeline.sh <tag>  -- kernel start/end times of a SMESH_FUSE_PIPELINE=1 bench run (do the two streams overlap?)
tag=$1; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SMESH_FUSE_PIPELINE=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag/kt -o bench -- python bench.py --no-cpu-baseline --steps 50 > gpurun_out/$tag/bench_kt.log 2>&1
python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/$tag/kt/bench_kernel_trace.csv")) if "synth" not in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=int(rows[-60]["Start_Timestamp"])
for r in rows[-60:-30]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","")[:22]
    print("%-22s q=%s start %8.1f end %8.1f dur %6.1f"%(n,r.get("Queue_Id"),(int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
PY
