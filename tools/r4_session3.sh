#!/bin/bash
# Round-4 GPU session 3: whole suite at the merged-launch kernels, then driver-shaped bench lines + kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite gpurun_out/r4s3
timeout 2400 python -m pytest tests -q -m gpu --maxfail=12 --durations=8 -o faulthandler_timeout=300 2>&1 | tail -150 > gpurun_out/suite/suite.log
grep -E "^FAILED|^ERROR|passed|failed|^E  |Timeout" gpurun_out/suite/suite.log | head -40
out=gpurun_out/r4s3
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench20.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-path --no-pmc > $GRAFT_REPO_ROOT/$out/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$out/rocprof.err )
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
rm -rf $out/prof
python tools/mesh_density_sweep.py > $out/density.txt 2>&1
python - <<'PY'
import json
for f in ("bench","bench_steps20_warmup5","bench_under_rocprof"):
    for l in open("gpurun_out/r4s3/%s.json"%f):
        if l.startswith('{"metric"'):
            d=json.loads(l); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"], d["roofline"].get("frac"), d["roofline"].get("frac_traffic"), d["config"].get("foreign_images"))
PY
head -9 $out/bench_kernel_stats.csv | cut -c1-150
grep -i "triangles" $out/density.txt
