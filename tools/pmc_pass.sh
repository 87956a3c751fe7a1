# usage: bash tools/pmc_pass.sh <tag> "<counters>"  -- one rocprofv3 --pmc pass over a short bench run, per-kernel averages
tag=$1; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --pmc $2 --output-format csv -d gpurun_out/$tag/pmc -o bench -- python bench.py --no-cpu-baseline --steps 20 --warmup 2 > gpurun_out/$tag/pmc.log 2>&1
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open("gpurun_out/$tag/pmc/bench_counter_collection.csv")):
    n=r["Kernel_Name"].replace("(anonymous namespace)::","")[:40]
    acc[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(n,r["Counter_Name"])]+=1
for n,d in acc.items():
    if "synth" in n: continue
    print(n); print("   "+"  ".join("%s=%.4g"%(k,v/cnt[(n,k)]) for k,v in sorted(d.items())))
PY
