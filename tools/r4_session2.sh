#!/bin/bash
# Round-4 GPU session 2: the tests that touch medium / big / huge triangles, then the driver-shaped bench lines and the density sweep
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4s2; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_multi.py tests/test_gpu_sharded.py -q -m gpu -x -k "medium or mixed or big or huge or room or interleaved or soup or fuse_views or ranged or sharded or cfg3 or overlap or shuffled" -o faulthandler_timeout=300 2>&1 | tail -15
python bench.py --no-cpu-baseline --no-host-path --no-pmc > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-pmc > $out/bench20.json 2> $out/bench20.err
python tools/mesh_density_sweep.py > $out/density.txt 2>&1
python - <<'PY'
import json
for f in ("bench","bench20"):
    for l in open("gpurun_out/r4s2/%s.json"%f):
        if l.startswith('{"metric"'):
            d=json.loads(l); print(f, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_us"])
PY
grep -i "triangles" $out/density.txt
