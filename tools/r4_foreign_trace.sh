#!/bin/bash
# Round-4: where an add() on a foreign image goes -- kernel trace (start offsets and durations of the last calls) at cfg2 geometry
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4ft; mkdir -p $out
python tools/generic_add_bench.py cfg2 16 2>&1 | grep add > $out/foreign.txt
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/kt -o gab -- python $GRAFT_REPO_ROOT/tools/generic_add_bench.py cfg2 16 > $GRAFT_REPO_ROOT/$out/gab.log 2>&1 )
find $out/kt -name "*kernel_trace.csv" -exec cp {} $out/trace.csv \;
find $out/kt -name "*kernel_stats.csv" -exec cp {} $out/stats.csv \;
rm -rf $out/kt
cat $out/foreign.txt
python - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r4ft/trace.csv")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
sel=rows[-32:]
t0=int(sel[0]["Start_Timestamp"]); prev_end=None
for r in sel:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][:44]
    print("%9.1f dur %7.1f gap %6.1f  %s" % ((s-t0)/1e3,(e-s)/1e3, 0 if prev_end is None else (s-prev_end)/1e3, n))
    prev_end=e
PY
