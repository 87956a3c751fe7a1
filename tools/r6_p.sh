#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6p
for x in 0 1; do echo "== SMESH_WIDE_LIST=$x"; SMESH_WIDE_LIST=$x timeout 900 python tools/wide_rows_sweep.py 136,160,176,192,208,224,240 2>&1 | grep -v amdgpu.ids | grep "sum  "; done | tee gpurun_out/r6p/wide_rows_sweep2.txt
