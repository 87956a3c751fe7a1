# Collects what profiles/ holds for a round: bench line, rocprofv3 kernel stats of the same command, HBM counters
# (separate --pmc passes).  usage: bash tools/profile_round.sh <tag>   (run on the GPU box through gpurun)
tag=$1; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python bench.py > $out/bench.log 2>&1; tail -1 $out/bench.log | cut -c1-400
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o bench -- python bench.py --no-cpu-baseline > $out/bench_kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o bench -- python bench.py --no-cpu-baseline --steps 20 --warmup 2 > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o bench -- python bench.py --no-cpu-baseline --steps 20 --warmup 2 > $out/pmc_write.log 2>&1
SMESH_FUSE=strip python bench.py --no-cpu-baseline > $out/bench_strip.log 2>&1; tail -1 $out/bench_strip.log | cut -c1-200
python - <<PY
import csv, collections, json
out="$out"
stats={}
for r in csv.DictReader(open(out+"/kt/bench_kernel_stats.csv")):
    stats[r["Name"].replace("(anonymous namespace)::","")]={"calls":int(r["Calls"]),"avg_us":float(r["AverageNs"])/1e3}
pm={}
for kind in ("fetch","write"):
    acc=collections.defaultdict(float); cnt=collections.Counter()
    for r in csv.DictReader(open(out+"/pmc_%s/bench_counter_collection.csv"%kind)):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","")
        acc[n]+=float(r["Counter_Value"]); cnt[n]+=1
    for n in acc: pm.setdefault(n,{})[kind.upper()+"_SIZE_KiB_avg"]=acc[n]/cnt[n]
json.dump({"kernel_stats":stats,"pmc":pm},open(out+"/summary.json","w"),indent=1)
for n,d in stats.items():
    if "synth" in n: continue
    print("%-62s %5d x %8.1f us   %s"%(n[:62],d["calls"],d["avg_us"],pm.get(n,"")))
PY
