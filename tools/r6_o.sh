#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6o
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_fuzz.py -x -q -m gpu -k "class_count or cfg5 or wide or any_class or 150 or 130 or 300" > gpurun_out/r6o/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6o/pytest.log
tail -4 gpurun_out/r6o/pytest.log
run() {  # label, env...
  label=$1; shift
  env "$@" timeout 900 python bench.py --workload cfg5 --no-pmc --no-host-path --repeats 3 > gpurun_out/r6o/cfg5_$label.json 2> gpurun_out/r6o/cfg5_$label.err
  L=$label python - <<'PY'
import json, os
l = os.environ["L"]
try:
    d = json.loads([x for x in open("gpurun_out/r6o/cfg5_%s.json" % l) if x.startswith("{")][-1])
    print("%-28s %.1f views/s; kernel us/view %.1f frac %.3f" % (l, d["value"], d["roofline"]["us_per_view"], d["roofline"]["frac"]))
except Exception as e:
    print(l, "failed", e, open("gpurun_out/r6o/cfg5_%s.err" % l).read()[-600:])
PY
}
run ring_8waves SMESH_WIDE_LIST=1
run ring_7waves SMESH_WIDE_LIST=1 SMESH_LIB_PATH=$root/semantic_meshes_amd/csrc_v7/libsmesh_hip.so
run ring_6waves SMESH_WIDE_LIST=1 SMESH_LIB_PATH=$root/semantic_meshes_amd/csrc_v6/libsmesh_hip.so
