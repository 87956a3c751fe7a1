#!/bin/bash
# cfg4 / cfg4t bench lines (no CPU baseline, no PMC): quick comparison of kernel variants
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4c4; mkdir -p $out
tag=${1:-x}
python bench.py --workload cfg4 --no-cpu-baseline --no-host-path --no-pmc > $out/cfg4_$tag.json 2>> $out/err.txt
python bench.py --workload cfg4t --no-cpu-baseline --no-host-path --no-pmc > $out/cfg4t_$tag.json 2>> $out/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4c4/*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d["roofline"]; print(f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_us"], r.get("us_per_view"), r.get("frac"))
PY
tail -2 $out/err.txt
