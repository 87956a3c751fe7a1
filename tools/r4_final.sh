#!/bin/bash
# Round-4 measurement batch at the round's kernels (one gpurun call): driver-shaped lines, kernel trace, cfg4 / cfg4t / cfg5 with counters,
# the N > 1 code path with one rank, the foreign-image trace, the density sweep.  Copies of what matters go to profiles/r04_*.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4final; mkdir -p $out
R=$GRAFT_REPO_ROOT
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench20.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-host-path --no-pmc > $R/$out/bench_under_rocprof.json 2> $R/$out/rocprof.err )
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \; ; rm -rf $out/prof
timeout 1500 python bench.py --workload cfg4 --cpu-baseline --pmc --no-host-path > $out/bench_cfg4.json 2> $out/cfg4.err
timeout 900 python bench.py --workload cfg4t --pmc --no-cpu-baseline --no-host-path > $out/bench_cfg4t.json 2> $out/cfg4t.err
timeout 1200 python bench.py --workload cfg5 --pmc --no-cpu-baseline --no-host-path > $out/bench_cfg5.json 2> $out/cfg5.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/prof4 -o bench -- python $R/bench.py --workload cfg4 --no-cpu-baseline --no-host-path --no-pmc > /dev/null 2> $R/$out/rocprof4.err )
find $out/prof4 -name "*kernel_stats.csv" -exec cp {} $out/bench_cfg4_kernel_stats.csv \; ; rm -rf $out/prof4
for parts in 4 1; do
  SMESH_BENCH_EXCHANGE_PARTS=$parts python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2961$parts bench.py --gpus 1 --steps 20 --warmup 5 --no-host-path > $out/launched_world1_parts$parts.json 2> $out/launched$parts.err
done
SMESH_BENCH_EXCHANGE_PARTS=4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 1 --steps 200 --warmup 10 --no-host-path > $out/launched_world1_parts4_steps200.json 2> $out/launched4b.err
python tools/mesh_density_sweep.py > $out/density.txt 2>&1
bash tools/r4_foreign_trace.sh > $out/foreign_trace.txt 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4final/*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d["roofline"]; c=d["config"]
            print("%-40s %9.1f views/s %.4f ms kernel %.1f us/view frac %s traffic_frac %s" % (f.split("/")[-1], d["value"], d["ms_per_step"], r.get("us_per_view") or 0, r.get("frac"), r.get("frac_traffic")))
            if "exchange_parts" in c: print("    ", {k:c[k] for k in ("compute_ms","exchange_ms","exchange_exposed_ms","timed_region_ms","exchange_parts","held_views") if k in c})
            if d.get("foreign_images"): print("    foreign", d["foreign_images"].get("ms_per_view"), d["foreign_images"].get("frac"))
            if d.get("cpu_baseline"): print("    cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("sample"))
PY
grep triangles $out/density.txt
