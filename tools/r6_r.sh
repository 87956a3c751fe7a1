#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6r
timeout 1500 python bench.py --workload cfg5 --no-host-path > gpurun_out/r6r/cfg5.json 2> gpurun_out/r6r/cfg5.err
python -c "
import json; d=json.loads([l for l in open('gpurun_out/r6r/cfg5.json') if l.startswith('{')][-1]); print(d['value'], d['roofline']['traffic'], d['roofline']['frac_traffic'], d['roofline']['traffic_source'])"
