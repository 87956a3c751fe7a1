#!/bin/bash
# round 6: the round's last build (index planes decided per view where every fusion launch is k_fuse_tri) -- the whole GPU suite, smoke(), then the cfg2
# measurements of tools/r6_final.sh again (cfg4 / cfg4t / cfg5 run kernels this build did not change: their round-6 lines stand).
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root
out=gpurun_out/r6f_final3; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log; tail -3 $out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench20.err
for mode in "" "--no-group-pipeline"; do
  tag=$([ -z "$mode" ] && echo pipelined || echo serial)
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/prof_$tag -o bench -- python $root/bench.py --no-cpu-baseline --no-host-path --no-pmc --repeats 3 $mode > $root/$out/bench_under_rocprof_$tag.json 2> $root/$out/rocprof_$tag.err )
  find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats_$tag.csv \;
  rm -rf $out/prof_$tag
done
bash tools/r6_pmc_workload.sh cfg2 r6f_final3/pmc_cfg2 > $out/pmc_raster_cfg2.txt 2>&1
python tools/two_thread_harness.py 64 2>&1 | grep -v amdgpu.ids > $out/two_thread_harness.txt
python tools/mesh_density_sweep.py 2>&1 | grep -v amdgpu.ids > $out/mesh_density_sweep.txt
python tools/close_view_bench.py 2>&1 | grep -v amdgpu.ids > $out/close_views.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6f_final3/bench*.json')):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); c=d['config']; r=d['roofline']
            print(f.split('/')[-1], d['value'], c.get('value_min'), c.get('value_max'), 'frac', r['frac'], 'needed', r['frac_needed'], 'traffic', r['frac_traffic'], 'us/view', r['us_per_view'])
            if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline'].get('value'), d['cpu_baseline'].get('cores'), (d['cpu_baseline'].get('optimised_cpu') or {}).get('value'))
PY
head -7 $out/bench_kernel_stats_serial.csv | cut -d, -f1-4 | cut -c1-150; grep -E "raster|resolve|project" $out/pmc_raster_cfg2.txt | cut -c1-60,330-420; cat $out/two_thread_harness.txt | head -7; cat $out/mesh_density_sweep.txt $out/close_views.txt
for f in $out/*.err; do if [ -s $f ]; then echo == $f; grep -v amdgpu.ids $f | tail -n 3; fi; done
