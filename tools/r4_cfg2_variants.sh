#!/bin/bash
# cfg2 bench with library variants under tools/_variants (same box, back to back)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4c2v; mkdir -p $out
cp semantic_meshes_amd/csrc/libsmesh_hip.so /tmp/keep.so
for rep in 1 2; do
for v in "$@"; do
  cp tools/_variants/$v.so semantic_meshes_amd/csrc/libsmesh_hip.so
  python bench.py --no-cpu-baseline --no-host-path --no-pmc > $out/cfg2_${v}_$rep.json 2>> $out/err.txt
done; done
cp /tmp/keep.so semantic_meshes_amd/csrc/libsmesh_hip.so
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c2v/*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d["roofline"]; print("%-24s %8.1f views/s  %.4f ms  fuse %.1f us/view frac %.3f" % (f.split("/")[-1], d["value"], d["ms_per_step"], r.get("us_per_view"), r.get("frac")))
PY
