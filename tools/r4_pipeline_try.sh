#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4gp; mkdir -p $out
for i in 1 2; do
for gp in 0 1; do
  SMESH_GROUP_PIPELINE=$gp python bench.py --no-cpu-baseline --no-host-path --no-pmc > $out/b200_gp${gp}_$i.json 2>> $out/err.txt
  SMESH_GROUP_PIPELINE=$gp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-pmc > $out/b20_gp${gp}_$i.json 2>> $out/err.txt
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4gp/*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d["roofline"]; print(f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_us"])
PY
