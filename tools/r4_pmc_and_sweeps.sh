#!/bin/bash
# Round-4: per-kernel HBM traffic (counters, separate passes) for cfg2 / cfg4, then larger differential sweeps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
bash tools/pmc_hbm.sh cfg2 --no-pmc > gpurun_out/pmc_cfg2.txt 2>&1
bash tools/pmc_hbm.sh cfg4 --no-pmc > gpurun_out/pmc_cfg4.txt 2>&1
out=gpurun_out/r4sw2; mkdir -p $out
timeout 2400 python tools/soup_sweep.py 30000 2000 > $out/soup_sweep.txt 2>&1
timeout 900 python tools/image_records_sweep.py 51000 3000 > $out/image_records_sweep.txt 2>&1
timeout 900 python tools/mul_sweep.py 61000 1000 > $out/mul_sweep.txt 2>&1
tail -2 gpurun_out/pmc_cfg2.txt | cut -c1-200; tail -2 $out/soup_sweep.txt; tail -2 $out/image_records_sweep.txt; tail -2 $out/mul_sweep.txt
