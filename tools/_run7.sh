tag=$1; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -ioE "(TCP_UTCL1|UTCL2)[A-Za-z0-9_]*" | sort -u | tr '\n' ' '; echo
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
d=gpurun_out/$tag/p_$(echo $set | cut -c1-12 | tr ' ' '_')
timeout 600 rocprofv3 --pmc $set --output-format csv -d $d -o c5 -- python tools/big_configs.py cfg5 > $d.log 2>&1
python - <<PY
import csv, collections, glob
f=glob.glob("$d/*counter_collection.csv")
if not f: print("no output for $set")
else:
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","")[:28]
        acc[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(n,r["Counter_Name"])]+=1
    for n,d in acc.items():
        if "fuse" in n or "raster_frag" in n: print(n, "  ".join("%s=%.4g"%(k,v/cnt[(n,k)]) for k,v in sorted(d.items())))
PY
done
