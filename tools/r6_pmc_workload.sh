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
if [ -n "$R6_PMC_MEMPIPE" ]; then   # the vector-memory pipeline of a CU: address (TA), cache (TCP) and data (TD) units
run ta "TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum GRBM_GUI_ACTIVE"
run tcp "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"
run tcp2 "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCR_TCP_STALL_CYCLES_sum"
run td "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_STORE_WAVEFRONT_sum TD_ATOMIC_WAVEFRONT_sum TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum"
fi
cd $root
OUT=$out python - <<'PY'
import csv, collections, glob, os
out = os.environ['OUT']
res = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('sq', 'sq2', 'tcc', 'fetch', 'write', 'ta', 'tcp', 'tcp2', 'td'):
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
        extra = {k: v for k, v in m.items() if k.startswith(('TA_', 'TCP_', 'TD_', 'GRBM'))}
        if extra:
            line = '      ' + '  '.join('%s=%.4g' % kv for kv in sorted(extra.items()))
            print(line); fo.write(line + '\n')
PY
