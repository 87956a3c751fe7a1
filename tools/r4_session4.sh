#!/bin/bash
# Round-4 GPU session 4: timing regions that ride on the launches (hipExtLaunchKernelGGL) against event records
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4s4; mkdir -p $out
timeout 900 python -m pytest tests -q -m gpu -x -k "profil or prof or bench or timing" -o faulthandler_timeout=300 2>&1 | tail -5
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-pmc > $out/b20_attach_$i.json 2>> $out/err.txt
  SMESH_PROF_ATTACH=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-pmc > $out/b20_record_$i.json 2>> $out/err.txt
done
python bench.py --no-cpu-baseline --no-host-path --no-pmc > $out/b200_attach.json 2>> $out/err.txt
SMESH_PROF_ATTACH=0 python bench.py --no-cpu-baseline --no-host-path --no-pmc > $out/b200_record.json 2>> $out/err.txt
SMESH_BENCH_PROFILE_EVERY=1 python bench.py --no-cpu-baseline --no-host-path --no-pmc > $out/b200_attach_every1.json 2>> $out/err.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4s4/*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d["roofline"]; print(f.split("/")[-1], d["value"], d["ms_per_step"], r["avg_launch_us"], r["launches_timed"], r.get("frac"))
PY
tail -3 $out/err.txt
