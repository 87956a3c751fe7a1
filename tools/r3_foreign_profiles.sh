# GPU session: bench lines at the new default of add() on foreign images, and kernel stats of that path at cfg2 / cfg5 geometry
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/fp; mkdir -p $out
last() { grep '^{"metric' $1 | tail -1; }
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench20.log 2>&1; last $out/bench20.log > $out/r03_bench_steps20_warmup5.json
timeout 600 python bench.py > $out/bench.log 2>&1; last $out/bench.log > $out/r03_bench.json
python - <<PY
import json
for f in ("r03_bench_steps20_warmup5.json", "r03_bench.json"):
    d = json.load(open("$out/" + f))
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], json.dumps(d.get("foreign_images"))[:600])
PY
{
for cfg in cfg2 cfg5; do
  python tools/generic_add_bench.py $cfg 8 2>&1 | grep add
  SMESH_REC_MOMENTS=0 python tools/generic_add_bench.py $cfg 8 2>&1 | grep add | sed 's/^/  (round 2 passes A-B) /'
  SMESH_ADD_RECORDS=0 python tools/generic_add_bench.py $cfg 8 2>&1 | grep add | sed 's/^/  (scatter-add) /'
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$cfg -o gab -- python tools/generic_add_bench.py $cfg 8 > $out/gab_$cfg.log 2>&1
  cp $out/kt_$cfg/gab_kernel_stats.csv $out/r03_foreign_images_${cfg}_kernel_stats.csv
  python - <<PY
import csv
for r in csv.DictReader(open("$out/kt_$cfg/gab_kernel_stats.csv")):
    n=r["Name"]
    if "k_rec" in n or "k_fuse" in n or "sparse" in n:
        print("    %-70s calls %4s avg %9.1f us"%(n.replace("(anonymous namespace)::","")[:70],r["Calls"],float(r["AverageNs"])/1e3))
PY
done
} > $out/r03_foreign_images.txt 2>&1
cat $out/r03_foreign_images.txt
