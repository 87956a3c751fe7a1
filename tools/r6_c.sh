#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6c
timeout 900 python -m pytest tests/test_gpu_deferred.py -x -q -m gpu > gpurun_out/r6c/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6c/pytest.log
tail -15 gpurun_out/r6c/pytest.log
bash tools/r6_ab.sh r6c/ab cfg2 "SMESH_RASTER_XCD=0" "SMESH_RASTER_XCD=2" "SMESH_RASTER_XCD=8" "SMESH_RASTER_XCD=32" "SMESH_RASTER_XCD=128" 2>&1 | tee gpurun_out/r6c/ab.txt
timeout 600 python tools/two_thread_harness.py 64 2>&1 | tee gpurun_out/r6c/two_thread.txt
for x in 0 1; do
  echo "== cfg5 SMESH_WIDE_XCD=$x"
  SMESH_WIDE_XCD=$x timeout 900 python bench.py --workload cfg5 --no-pmc --no-host-path --repeats 3 > gpurun_out/r6c/cfg5_$x.json 2> gpurun_out/r6c/cfg5_$x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6c/cfg5_$x.json") if l.startswith("{")][-1])
    print("   %.1f views/s; kernel %s us/view %.1f frac %.3f" % (d["value"], d["roofline"]["kernel"][:20], d["roofline"]["us_per_view"], d["roofline"]["frac"]))
except Exception as e:
    print("   failed", e, open("gpurun_out/r6c/cfg5_$x.err").read()[-600:])
PY
done 2>&1 | tee gpurun_out/r6c/cfg5.txt
