#!/bin/bash
# round 6: the driver-shaped line (--steps 20 --warmup 5) under different ways of cutting twenty views into groups.  usage: bash tools/r6_steps20.sh "ENV=.. --flag" ...
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
for v in "$@"; do
  envs=$(echo "$v" | tr ' ' '\n' | grep = | grep -v '^--' | tr '\n' ' '); flags=$(echo "$v" | tr ' ' '\n' | grep '^--' | tr '\n' ' ')
  env $envs python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path --no-pmc $flags 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); c = d['config']
        print('%-60s %8.0f views/s (min %6.0f max %6.0f)  region_ms %s' % (sys.argv[1], d['value'], c['value_min'], c['value_max'], c.get('region_ms')))
" "$v"
done
