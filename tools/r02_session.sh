# Round-2 GPU session: tests, bench lines at the driver's arguments and the default ones, the other single-GPU configs,
# rocprofv3 kernel stats of the same commands.  usage: bash tools/r02_session.sh <tag> [steps...]
tag=$1; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
what=${2:-all}
if [ "$what" = all ] || [ "$what" = test ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; tail -5 $out/pytest.log
fi
if [ "$what" = all ] || [ "$what" = bench ]; then
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $out/bench_20.log 2>&1; tail -1 $out/bench_20.log | cut -c1-1800
  python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log | cut -c1-2600
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt20 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $out/bench_kt20.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o bench -- python bench.py --no-cpu-baseline --no-host-path > $out/bench_kt.log 2>&1
  for d in kt20 kt; do python - <<PY
import csv
print("== $d")
for r in csv.DictReader(open("$out/$d/bench_kernel_stats.csv")):
    n=r["Name"]
    if "synth" in n or "rocclr" in n: continue
    print("  %-70s calls %5s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
  done
  grep -h -o '"frac": [0-9.]*\|"avg_launch_us": [0-9.]*\|"views_per_launch": [0-9.]*' $out/bench_kt20.log $out/bench_kt.log
fi
if [ "$what" = all ] || [ "$what" = big ]; then
  for w in cfg4 cfg5; do
    timeout 900 python bench.py --workload $w > $out/bench_$w.log 2>&1; tail -1 $out/bench_$w.log | cut -c1-2200
    timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -o bench -- python bench.py --workload $w --no-host-path > $out/bench_kt_$w.log 2>&1
    python - <<PY
import csv
print("== $w")
for r in csv.DictReader(open("$out/kt_$w/bench_kernel_stats.csv")):
    n=r["Name"]
    if "synth" in n or "rocclr" in n: continue
    print("  %-70s calls %5s avg %8.1f us"%(n.replace("(anonymous namespace)::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
  done
fi
