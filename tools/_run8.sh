tag=$1; mkdir -p gpurun_out/$tag; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_32B_sum"; do
i=$((i+1)); d=gpurun_out/$tag/p$i
timeout 600 rocprofv3 --pmc $set --output-format csv -d $d -o c5 -- python tools/big_configs.py cfg5 > $d.log 2>&1
python - <<PY
import csv, collections, glob
f=glob.glob("$d/*counter_collection.csv")
if not f: print("no output: $set")
else:
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        n=r["Kernel_Name"].replace("(anonymous namespace)::","")[:28]
        acc[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(n,r["Counter_Name"])]+=1
    for n,d in acc.items():
        if "fuse" in n: print(n, "  ".join("%s=%.4g"%(k,v/cnt[(n,k)]) for k,v in sorted(d.items())))
PY
done
