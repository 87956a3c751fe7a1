#!/bin/bash
# Round-5 session 1: the restructured bench (repeats / median, group pipeline by default, serialised roofline leg) on cfg2, and the
# stream-flattened k_fuse_tri_wide at cfg5 with 2 / 4 / 8 rows in flight per wave.
out=gpurun_out/r5s1; mkdir -p $out
cd /root/repo
python bench.py --no-cpu-baseline > $out/bench_cfg2.json 2> $out/bench_cfg2.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_cfg2_s20.json 2> $out/bench_cfg2_s20.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-host-path --no-group-pipeline > $out/bench_cfg2_s20_serial.json 2> $out/bench_cfg2_s20_serial.err
for b in 4 2 8; do
  SMESH_WIDE_B=$b python bench.py --workload cfg5 --no-pmc --no-host-path --repeats 3 > $out/bench_cfg5_B$b.json 2> $out/bench_cfg5_B$b.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5s1/*.json')):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); c=d['config']; r=d['roofline']
            print(f.split('/')[-1], d['value'], 'min/max', c.get('value_min'), c.get('value_max'), 'spread', c.get('value_spread'), 'gp', c.get('group_pipeline'),
                  'frac', r['frac'], 'frac_needed', r['frac_needed'], 'frac_traffic', r['frac_traffic'], 'us/view', r['us_per_view'], 'launches', r['launches_by_views'], 'traffic', r['traffic'], r.get('traffic_by_views_per_launch'))
            if 'foreign_images' in d: print('   foreign', d['foreign_images'].get('ms_per_view'), d['foreign_images'].get('frac'))
PY
tail -3 $out/*.err
