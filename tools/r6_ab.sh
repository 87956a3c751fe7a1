#!/bin/bash
# round 6: A/B of rasteriser knobs at a workload -- per-kernel averages (serialised, rocprofv3 kernel trace) and the pipelined value.
# usage: bash tools/r6_ab.sh <tag> <workload> "ENV1=a ENV2=b" "ENV1=c" ...      (each quoted string = one variant's environment)
tag=$1; w=$2; shift 2
root=${GRAFT_REPO_ROOT:-/root/repo}; out=$root/gpurun_out/$tag; mkdir -p $out
i=0
for envs in "$@"; do
  i=$((i+1))
  ( cd /tmp && export TMPDIR=/tmp && env $envs timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace$i -o t -- python $root/bench.py --workload $w --steps 64 --warmup 8 --repeats 1 --no-cpu-baseline --no-host-path --no-pmc --no-group-pipeline > $out/trace$i.log 2>&1 )
  ( cd $root && env $envs timeout 600 python bench.py --workload $w --steps 200 --warmup 10 --repeats 5 --no-cpu-baseline --no-host-path --no-pmc > $out/bench$i.json 2> $out/bench$i.err )
  echo "== variant $i: $envs"
  OUT=$out I=$i python - <<'PY'
import csv, glob, os, json
out, i = os.environ['OUT'], os.environ['I']
for f in glob.glob('%s/trace%s/**/*kernel_stats.csv' % (out, i), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('(anonymous namespace)::', '')
        if any(k in n for k in ('raster', 'resolve', 'project', 'fuse_t', 'k_fuse')):
            print("   %-56s calls %5s avg %9.1f us" % (n[:56], r['Calls'], float(r['AverageNs']) / 1e3))
try:
    d = json.loads([l for l in open('%s/bench%s.json' % (out, i)) if l.startswith('{')][-1])
    print("   pipelined: %.0f views/s (min %.0f max %.0f)  ms/step %.4f" % (d['value'], d['config']['value_min'], d['config']['value_max'], d['ms_per_step']))
except Exception as e:
    print("   bench failed:", e, open('%s/bench%s.err' % (out, i)).read()[-400:])
PY
done
