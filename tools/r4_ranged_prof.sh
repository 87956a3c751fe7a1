#!/bin/bash
out=gpurun_out/r4g; mkdir -p $out
cd /root/repo
for parts in 1 2 3 4; do
  SMESH_BENCH_EXCHANGE_PARTS=$parts python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$parts bench.py --gpus 1 --steps 20 --warmup 5 --no-host-path > $out/parts$parts.json 2> $out/parts$parts.err
done
( cd /tmp && export TMPDIR=/tmp && RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 SMESH_BENCH_EXCHANGE_PARTS=4 rocprofv3 --kernel-trace --output-format csv -d /root/repo/$out/prof -o tr -- python /root/repo/bench.py --gpus 1 --steps 20 --warmup 5 --no-host-path > /root/repo/$out/traced.json 2> /root/repo/$out/traced.err )
find $out/prof -name "*kernel_trace.csv" -exec cp {} $out/trace.csv \;
rm -rf $out/prof
python - <<'PY'
import csv,json,glob
for f in sorted(glob.glob("gpurun_out/r4g/parts*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); c=d["config"]; print(f, d["value"], {k:c[k] for k in ("compute_ms","exchange_ms","exchange_exposed_ms","timed_region_ms","exchange_parts","held_views")})
rows=list(csv.DictReader(open("gpurun_out/r4g/trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last 60 kernels
t0=int(rows[-70]["Start_Timestamp"])
for r in rows[-70:]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:40]
    print("%9.1f %8.1f  %s" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, n))
PY
