#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6l
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu -k "class_count or cfg5 or wide or any_class or 150 or 130 or 300" > gpurun_out/r6l/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6l/pytest.log
tail -5 gpurun_out/r6l/pytest.log
for x in 0 1; do
  echo "== cfg5 SMESH_WIDE_LIST=$x"
  SMESH_WIDE_LIST=$x timeout 900 python bench.py --workload cfg5 --no-pmc --no-host-path --repeats 3 > gpurun_out/r6l/cfg5_$x.json 2> gpurun_out/r6l/cfg5_$x.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6l/cfg5_$x.json") if l.startswith("{")][-1])
    print("   %.1f views/s; kernel %s us/view %.1f frac %.3f" % (d["value"], d["roofline"]["kernel"][:20], d["roofline"]["us_per_view"], d["roofline"]["frac"]))
except Exception as e:
    print("   failed", e, open("gpurun_out/r6l/cfg5_$x.err").read()[-600:])
PY
done 2>&1 | tee gpurun_out/r6l/cfg5.txt
