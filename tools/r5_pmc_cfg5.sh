#!/bin/bash
# Round 5: where do k_fuse_tri_wide's cycles go at cfg5?  SQ occupancy / stall counters, L2 hit rate, TLB
# counters if the tool lists them, and the HBM traffic re-measured this round.  One counter set per rocprofv3 pass.
out=gpurun_out/r5pmc5; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > /root/repo/$out/counters_list.txt 2>&1
cd /root/repo
run() {  # name, counters
  ( cd /tmp && timeout 900 rocprofv3 --pmc $2 --output-format csv -d /root/repo/$out/$1 -o b -- python /root/repo/bench.py --workload cfg5 --steps 16 --warmup 8 --repeats 1 --no-cpu-baseline --no-host-path --no-pmc --no-group-pipeline > /root/repo/$out/$1.log 2>&1 )
}
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
run tcp "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"
python - <<'PY'
import csv, collections, glob, os
out='gpurun_out/r5pmc5'
res=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('sq','sq2','tcc','fetch','write','tcp'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv'%(out,d), recursive=True):
        for r in csv.DictReader(open(f)):
            n=r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0]
            res[n][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for n,d in res.items():
        if 'synth' in n or 'rocclr' in n or '__amd' in n: continue
        line=n[:70]+'\n   '+'  '.join('%s=%.5g (x%d)'%(k,sum(v)/len(v),len(v)) for k,v in sorted(d.items()))
        print(line); fo.write(line+'\n')
PY
grep -i "utcl\|tlb" $out/counters_list.txt | head -20
