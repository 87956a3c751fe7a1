#!/bin/bash
# cfg5 ablation ladder of the wide-row kernels (development build; SMESH_FDBG: 1 = stop after the records pass, 2 = no row stores, 4 = no row loads, 8 = no class-vector loads)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6n
export SMESH_LIB_PATH=$root/semantic_meshes_amd/csrc_abl/libsmesh_hip.so
for lst in 0 1; do for d in 0 1 14 8 6; do
  SMESH_WIDE_LIST=$lst SMESH_FDBG=$d timeout 600 python bench.py --workload cfg5 --no-pmc --no-host-path --repeats 1 --steps 16 --warmup 8 > gpurun_out/r6n/c_${lst}_$d.json 2> gpurun_out/r6n/c_${lst}_$d.err
  python - <<PY
import json
try:
    d = json.loads([l for l in open("gpurun_out/r6n/c_${lst}_$d.json") if l.startswith("{")][-1])
    print("SMESH_WIDE_LIST=$lst SMESH_FDBG=%-2s  kernel us per view %8.1f" % ("$d", d["roofline"]["us_per_view"]))
except Exception as e:
    print("failed $lst $d", e, open("gpurun_out/r6n/c_${lst}_$d.err").read()[-300:])
PY
done; done 2>&1 | tee gpurun_out/r6n/ladder.txt
