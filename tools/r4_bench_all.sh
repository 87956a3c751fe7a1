#!/bin/bash
# Round-4 measurement batch (one gpurun call): the driver-shaped lines, the kernel trace, the N > 1 code path with one rank.
set -x
out=gpurun_out/r4f; mkdir -p $out
cd /root/repo
python bench.py > $out/bench.json 2> $out/bench.err
python bench.py --steps 20 --warmup 5 > $out/bench_steps20_warmup5.json 2> $out/bench20.err
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof -o bench -- python /root/repo/bench.py --no-cpu-baseline --no-host-path --no-pmc > /root/repo/$out/bench_under_rocprof.json 2> /root/repo/$out/rocprof.err )
find $out/prof -name "*kernel_stats.csv" -exec cp {} $out/bench_kernel_stats.csv \;
rm -rf $out/prof
for parts in 4 1; do
  SMESH_BENCH_EXCHANGE_PARTS=$parts python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --no-host-path > $out/launched_world1_parts$parts.json 2> $out/launched$parts.err
done
SMESH_BENCH_EXCHANGE_PARTS=4 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 200 --warmup 10 --no-host-path > $out/launched_world1_parts4_steps200.json 2> $out/launched4b.err
tail -c 600 $out/bench.json; echo; tail -c 300 $out/bench_steps20_warmup5.json; echo; head -12 $out/bench_kernel_stats.csv
for f in $out/launched_*.json; do python - "$f" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); c=d["config"]; print(sys.argv[1], d["value"], {k:c[k] for k in ("compute_ms","exchange_ms","exchange_exposed_ms","timed_region_ms","exchange_parts","held_views")})
PY
done
