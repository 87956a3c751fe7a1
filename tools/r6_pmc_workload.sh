#!/bin/bash
# Round 6: PMC counters of every kernel of a workload's view pipeline (the rasteriser's FETCH / WRITE and SQ wait
# breakdown at cfg2).  One counter set per rocprofv3 pass over `bench.py --workload $1 --steps 16 --warmup 8 --repeats 1 --no-group-pipeline`.
# usage: bash tools/r6_pmc_workload.sh <workload> <tag>     -> gpurun_out/<tag>/summary.txt
w=${1:-cfg2}; tag=${2:-r6pmc_$w}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters
  ( cd /tmp && timeout 900 rocprofv3 --pmc $2 --output-format csv -d $out/$1 -o b -- python $root/bench.py --workload $w --steps 16 --warmup 8 --repeats 1 --no-cpu-baseline --no-host-path --no-pmc --no-group-pipeline > $out/$1.log 2>&1 )
}
run sq "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"
run sq2 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
run tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
cd $root
OUT=$out python - <<'PY'
import csv, collections, glob, os
out = os.environ['OUT']
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('sq', 'sq2', 'tcc', 'fetch', 'write'):
    for f in glob.glob('%s/%s/**/*counter_collection.csv' % (out, d), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
            res[n][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out + '/summary.txt', 'w') as fo:
    for n, d in res.items():
        if 'synth' in n or 'rocclr' in n or '__amd' in n: continue
        m = {k: sum(v) / len(v) for k, v in d.items()}
        wc = m.get('SQ_WAVE_CYCLES') or 1.0
        wv = m.get('SQ_WAVES') or 1.0
        line = ("%-40s x%-3d waves %8.0f  quad-cycles/wave %7.0f  parked on memory %4.1f%%  issue-stalled %4.1f%%  issuing %4.1f%% (VALU %4.1f%%)  "
                "per wave: VALU %6.0f SALU %6.0f LDS %5.0f VMEM rd %5.0f wr %4.0f  L2 hit rate %4.1f%% (%.3g requests)  "
                "FETCH %.1f MB x2 + WRITE %.1f MB = traffic %.1f MB per launch" % (
                    n[:40], len(d.get('SQ_WAVES', [0])), wv, wc / wv, 100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc,
                    100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_VALU', 0) / wc,
                    m.get('SQ_INSTS_VALU', 0) / wv, m.get('SQ_INSTS_SALU', 0) / wv, m.get('SQ_INSTS_LDS', 0) / wv,
                    m.get('SQ_INSTS_VMEM_RD', 0) / wv, m.get('SQ_INSTS_VMEM_WR', 0) / wv,
                    100 * m.get('TCC_HIT_sum', 0) / max(m.get('TCC_REQ_sum', 1), 1), m.get('TCC_REQ_sum', 0),
                    m.get('FETCH_SIZE', 0) * 1024 / 1e6, m.get('WRITE_SIZE', 0) * 1024 / 1e6,
                    (2 * m.get('FETCH_SIZE', 0) + m.get('WRITE_SIZE', 0)) * 1024 / 1e6))
        print(line); fo.write(line + '\n')
PY
