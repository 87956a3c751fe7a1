#!/bin/bash
# round 6: the cfg2 / cfg5 measurements again at the final build (the index-plane decision per raster launch) -- same commands as tools/r6_final.sh
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
timeout 1500 python bench.py --workload cfg5 > $out/bench_cfg5.json 2> $out/cfg5.err
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_cfg5 -o bench -- python $root/bench.py --workload cfg5 --no-cpu-baseline --no-host-path --no-pmc --repeats 3 --no-group-pipeline > $root/$out/bench_cfg5_serial.json 2> $root/$out/cfg5_serial.err )
find $out/prof_cfg5 -name "*kernel_stats.csv" -exec cp {} $out/bench_cfg5_kernel_stats.csv \; ; rm -rf $out/prof_cfg5
python tools/two_thread_harness.py 64 2>&1 | grep -v amdgpu.ids > $out/two_thread_harness.txt
python tools/mesh_density_sweep.py 2>&1 | grep -v amdgpu.ids > $out/mesh_density_sweep.txt
python tools/close_view_bench.py 2>&1 | grep -v amdgpu.ids > $out/close_views.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6f_final/bench*.json')):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); c=d['config']; r=d['roofline']
            print(f.split('/')[-1], d['value'], c.get('value_min'), c.get('value_max'), 'frac', r['frac'], 'needed', r['frac_needed'], 'traffic', r['frac_traffic'], 'us/view', r['us_per_view'])
            if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'), (d['cpu_baseline'].get('optimised_cpu') or {}).get('value'))
PY
head -7 $out/bench_kernel_stats_serial.csv | cut -d, -f1-4 | cut -c1-150; grep -E "raster|resolve|project" $out/pmc_raster_cfg2.txt | cut -c1-60,330-420; cat $out/two_thread_harness.txt | head -7; cat $out/mesh_density_sweep.txt $out/close_views.txt
