# kernel trace of the group pipeline (raster of group g+1 beside the fusion of group g): do the kernels overlap, and what does it do to their durations?
tag=$1; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
SMESH_GROUP_PIPELINE=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/kt -o b -- python bench.py --steps 80 --warmup 16 --no-cpu-baseline --no-host-path > $out/b.log 2>&1
tail -1 $out/b.log | cut -c1-160
OUT=$out python - <<'PY'
import csv, os
rows = list(csv.DictReader(open(os.environ["OUT"] + "/kt/b_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_fuse_tri<19, 0, true, 8>" in n]
i0 = idx[len(idx) // 2]
t0 = int(rows[i0 - 6]["Start_Timestamp"])
for r in rows[i0 - 6:i0 + 10]:
    s = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    print("%-36s start %8.1f end %8.1f dur %7.1f us  queue %s" % (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:36], (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get("Queue_Id", "?")))
PY
