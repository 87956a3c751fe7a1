#!/bin/bash
# Round-4 GPU session 5: whole suite, then cfg2 / cfg4 / cfg4t / cfg5 lines
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite gpurun_out/r4s5
timeout 2400 python -m pytest tests -q -m gpu --maxfail=12 --durations=5 -o faulthandler_timeout=300 2>&1 | tail -150 > gpurun_out/suite/suite.log
grep -E "^FAILED|^ERROR|passed|failed|^E  |Timeout" gpurun_out/suite/suite.log | head -40
out=gpurun_out/r4s5
python bench.py --no-cpu-baseline --no-host-path --no-pmc > $out/cfg2.json 2>> $out/err.txt
for w in cfg4 cfg4t cfg5; do python bench.py --workload $w --no-cpu-baseline --no-host-path --no-pmc > $out/$w.json 2>> $out/err.txt; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4s5/*.json")):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); r=d["roofline"]; print("%-14s %8.1f views/s  %.4f ms  kernel %.1f us/view frac %.3f" % (f.split("/")[-1], d["value"], d["ms_per_step"], r.get("us_per_view"), r.get("frac")))
PY
