#!/bin/bash
# Round-6 measurement batch (one gpurun call): the driver-shaped lines, kernel traces (default = group pipeline; and serialised), the other
# BASELINE configs with their counter traffic measured IN the run and their serialised kernel traces, the PMC counters of the cfg2 view
# pipeline at the final kernels, the mesh-density sweep, views from inside, the two-thread harness, the N > 1 code path with one rank and
# with two ranks started by the script itself.  Summaries are copied into profiles/r06_* afterwards.
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
out=gpurun_out/r6f_final; mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench20.err
for mode in "" "--no-group-pipeline"; do
  tag=$([ -z "$mode" ] && echo pipelined || echo serial)
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$tag -o bench -- python $root/bench.py --no-cpu-baseline --no-host-path --no-pmc --repeats 3 $mode > $root/$out/bench_under_rocprof_$tag.json 2> $root/$out/rocprof_$tag.err )
  find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats_$tag.csv \;
  rm -rf $out/prof_$tag
done
bash tools/r6_pmc_workload.sh cfg2 r6f_final/pmc_cfg2 > $out/pmc_raster_cfg2.txt 2>&1
for w in cfg4 cfg4t cfg5; do
  timeout 1500 python bench.py --workload $w > $out/bench_$w.json 2> $out/$w.err
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$w -o bench -- python $root/bench.py --workload $w --no-cpu-baseline --no-host-path --no-pmc --repeats 3 --no-group-pipeline > $root/$out/bench_${w}_serial.json 2> $root/$out/${w}_serial.err )
  find $out/prof_$w -name "*kernel_stats.csv" -exec cp {} $out/bench_${w}_kernel_stats.csv \;
  rm -rf $out/prof_$w
done
python tools/mesh_density_sweep.py 2>&1 | grep -v amdgpu.ids > $out/mesh_density_sweep.txt
python tools/close_view_bench.py 2>&1 | grep -v amdgpu.ids > $out/close_views.txt
python tools/two_thread_harness.py 64 2>&1 | grep -v amdgpu.ids > $out/two_thread_harness.txt
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-host-path > $out/launched_world1.json 2> $out/launched.err
SMESH_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 > $out/self_launched_world2_gloo.json 2> $out/self_launched.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6f_final/*.json')):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); c=d['config']; r=d['roofline']
            print(f.split('/')[-1], d['value'], 'n_gpus', d['n_gpus'], 'spread', c.get('value_spread'), 'gp', c.get('group_pipeline'), 'frac', r['frac'], 'needed', r['frac_needed'], 'traffic', r['frac_traffic'], 'us/view', r['us_per_view'])
            if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'), (d['cpu_baseline'].get('optimised_cpu') or {}).get('value'))
PY
head -8 $out/bench_kernel_stats_serial.csv | cut -c1-160; cat $out/mesh_density_sweep.txt; cat $out/close_views.txt; cat $out/two_thread_harness.txt
for f in $out/*.err; do if [ -s $f ]; then echo == $f; grep -v amdgpu.ids $f | tail -n 3; fi; done
