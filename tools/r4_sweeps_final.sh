#!/bin/bash
# Round-4, final HEAD: differential sweeps over further seeds (after the get() staging fix)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r4sw3; mkdir -p $out
timeout 2400 python tools/soup_sweep.py 40000 3000 > $out/soup_sweep.txt 2>&1
SMESH_RESULT_POOL_MB=0 timeout 900 python tools/soup_sweep.py 30900 700 > $out/soup_sweep_pageable_results.txt 2>&1
grep -v amdgpu.ids $out/soup_sweep.txt | tail -4 | cut -c1-300; grep -v amdgpu.ids $out/soup_sweep_pageable_results.txt | tail -3 | cut -c1-300
