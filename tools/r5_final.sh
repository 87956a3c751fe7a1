#!/bin/bash
# Round-5 measurement batch (one gpurun call): the driver-shaped lines, kernel traces (default = group pipeline; and serialised),
# the other BASELINE configs with their counter traffic measured in the run, the mesh-density sweep, the N > 1 code path with one
# rank, the contended hunt on the final build.  Summaries are copied into profiles/r05_* afterwards.
out=gpurun_out/r5f; mkdir -p $out
cd /root/repo
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench20.err
for mode in "" "--no-group-pipeline"; do
  tag=$([ -z "$mode" ] && echo pipelined || echo serial)
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$tag -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-host-path --no-pmc --repeats 3 $mode > /root/repo/$out/bench_under_rocprof_$tag.json 2> /root/repo/$out/rocprof_$tag.err )
  find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats_$tag.csv \;
  rm -rf $out/prof_$tag
done
python bench.py --workload cfg4 > $out/bench_cfg4.json 2> $out/cfg4.err
python bench.py --workload cfg4t > $out/bench_cfg4t.json 2> $out/cfg4t.err
python bench.py --workload cfg5 > $out/bench_cfg5.json 2> $out/cfg5.err
python tools/mesh_density_sweep.py 2>&1 | grep -v amdgpu.ids > $out/mesh_density_sweep.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-host-path > $out/launched_world1.json 2> $out/launched.err
(timeout 500 python tests/flake_hunt.py fuse --workers 8 --iters 2500 --hog 1 --timeout 480 2>&1 | grep -v amdgpu.ids | grep "FLAKE\|mismatches\|flake_hunt" > $out/fuse_hunt_final.log)
(timeout 600 python tests/flake_hunt.py exchange --repeats 25 --hog 1 --timeout 550 2>&1 | grep "FLAKE\|MISMATCH\|flake_hunt" > $out/exchange_hunt_final.log)
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5f/*.json')):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); c=d['config']; r=d['roofline']
            print(f.split('/')[-1], d['value'], 'spread', c.get('value_spread'), 'gp', c.get('group_pipeline'), 'frac', r['frac'], 'needed', r['frac_needed'], 'traffic', r['frac_traffic'], 'us/view', r['us_per_view'], 'get_ms', c.get('get_ms'))
            if 'foreign_images' in d:
                fi=d['foreign_images']; print('   foreign seq', fi.get('ms_per_view'), fi.get('frac'), 'batched', (fi.get('batched') or {}).get('ms_per_view'), (fi.get('batched') or {}).get('frac'))
            if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'), (d['cpu_baseline'].get('optimised_cpu') or {}).get('value'))
PY
head -8 $out/bench_kernel_stats_serial.csv | cut -c1-160; cat $out/mesh_density_sweep.txt; cat $out/fuse_hunt_final.log | tail -3; cat $out/exchange_hunt_final.log
for f in $out/*.err; do if [ -s $f ]; then echo == $f; grep -v amdgpu.ids $f | tail -n 3; fi; done
