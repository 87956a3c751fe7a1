# usage: bash tools/pmc_cfg.sh <workload> "<counters>"  -- one rocprofv3 --pmc pass over a short bench run of a workload, per-kernel averages
w=$1; tag=pmc_$w; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/$tag/pmc -o bench -- python bench.py --workload $w --no-cpu-baseline --no-host-path --steps 16 --warmup 8 > gpurun_out/$tag/pmc.log 2>&1
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open("gpurun_out/$tag/pmc/bench_counter_collection.csv")):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","")[:44]
    acc[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(n,r["Counter_Name"])]+=1
for n,d in acc.items():
    if "synth" in n or "rocclr" in n: continue
    print(n); print("   "+"  ".join("%s=%.4g"%(k,v/cnt[(n,k)]) for k,v in sorted(d.items())))
PY
