#!/bin/bash
# rocprofv3 --kernel-trace --stats of the other BASELINE configs at the round's final kernels (serialised: every kernel on its own)
out=gpurun_out/r5t; mkdir -p $out; cd /root/repo
for w in cfg4 cfg4t cfg5; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$w -o bench -- python /root/repo/bench.py --workload $w --no-cpu-baseline --no-host-path --no-pmc --repeats 3 --no-group-pipeline > /root/repo/$out/bench_$w.json 2> /root/repo/$out/$w.err )
  find $out/prof_$w -name "*kernel_stats.csv" -exec cp {} $out/bench_${w}_kernel_stats.csv \;
  rm -rf $out/prof_$w
  echo "== $w"; head -7 $out/bench_${w}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150
done
