#!/bin/bash
# Round-5 session 2: lane compaction in the rasteriser (bit-exact tests, cfg2 / cfg4 timings with it off and on), add_many, the
# group pipeline as the default.
out=gpurun_out/r5s2; mkdir -p $out
cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_image_records.py tests/test_gpu_multi.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -8 > $out/tests.log
cat $out/tests.log
for c in 0 1; do
  SMESH_RASTER_COMPACT=$c python bench.py --no-cpu-baseline --no-pmc --no-host-path --no-group-pipeline --repeats 3 > $out/cfg2_serial_compact$c.json 2> $out/cfg2_serial_compact$c.err
  SMESH_BENCH_PROFILE_ALL=1 SMESH_RASTER_COMPACT=$c python bench.py --no-cpu-baseline --no-pmc --no-host-path --no-group-pipeline --repeats 3 --steps 40 > $out/cfg2_profall_compact$c.json 2> $out/cfg2_profall_compact$c.err
  SMESH_RASTER_COMPACT=$c python bench.py --no-cpu-baseline --no-pmc --no-host-path --repeats 5 > $out/cfg2_pipe_compact$c.json 2> $out/cfg2_pipe_compact$c.err
  SMESH_RASTER_COMPACT=$c python bench.py --workload cfg4 --no-pmc --no-host-path --repeats 3 > $out/cfg4_compact$c.json 2> $out/cfg4_compact$c.err
done
SMESH_RASTER_COMPACT_GROUPS=8 python bench.py --no-cpu-baseline --no-pmc --no-host-path --repeats 5 > $out/cfg2_pipe_compact1_g8.json 2> $out/cfg2_pipe_compact1_g8.err
SMESH_RASTER_COMPACT_GROUPS=2 python bench.py --no-cpu-baseline --no-pmc --no-host-path --repeats 5 > $out/cfg2_pipe_compact1_g2.json 2> $out/cfg2_pipe_compact1_g2.err
python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 > $out/cfg2_s20_default.json 2> $out/cfg2_s20_default.err
./tools/stream_bench gather > $out/stream_gather.txt 2>&1; cat $out/stream_gather.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5s2/*.json')):
    for l in open(f):
        if l.startswith('{"metric"'):
            d=json.loads(l); c=d['config']; r=d['roofline']
            print(f.split('/')[-1], d['value'], 'spread', c.get('value_spread'), 'gp', c.get('group_pipeline'), 'us/view', r['us_per_view'], 'other', r.get('other_kernels_us_per_view'))
            if 'foreign_images' in d:
                fi=d['foreign_images']; print('   foreign seq', fi.get('ms_per_view'), fi.get('frac'), 'batched', fi.get('batched'))
PY
for f in $out/*.err; do if [ -s $f ]; then echo == $f; grep -v amdgpu.ids $f | tail -n 3; fi; done
