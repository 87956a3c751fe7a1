#!/bin/bash
# round 6: pipelined value only, per variant environment.  usage: bash tools/r6_pipe.sh <tag> <workload> "ENV=.." "ENV=.." ...
tag=$1; w=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/$tag; mkdir -p $out
i=0
for envs in "$@"; do
  i=$((i+1))
  ( cd $root && env $envs timeout 600 python bench.py --workload $w --steps 200 --warmup 10 --repeats 5 --no-cpu-baseline --no-host-path --no-pmc > $out/bench$i.json 2> $out/bench$i.err )
  OUT=$out I=$i ENVS="$envs" python - <<'PY'
import os, json
out, i = os.environ['OUT'], os.environ['I']
try:
    d = json.loads([l for l in open('%s/bench%s.json' % (out, i)) if l.startswith('{')][-1])
    print("%-60s pipelined: %.0f views/s (min %.0f max %.0f)  kernel us/view %.1f" % (os.environ['ENVS'], d['value'], d['config']['value_min'], d['config']['value_max'], d['roofline']['us_per_view']))
except Exception as e:
    print(os.environ['ENVS'], "bench failed:", e, open('%s/bench%s.err' % (out, i)).read()[-400:])
PY
done
