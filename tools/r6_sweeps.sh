#!/bin/bash
# Round-6 final kernels: the randomised differential tests over seeds beyond the committed ones (one gpurun call)
# usage: bash tools/r6_sweeps.sh [dense seeds] [soup seeds] [seed offset]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r6sw; mkdir -p $out
SMESH_SWEEP_ONLY=dense timeout 1800 python tools/soup_sweep.py $((50000 + ${3:-0})) ${1:-300} > $out/dense_sweep.txt 2>&1
timeout 1500 python tools/soup_sweep.py $((52000 + ${3:-0})) ${2:-2000} > $out/soup_sweep.txt 2>&1
timeout 600 python tools/image_records_sweep.py $((71000 + ${3:-0})) 2000 > $out/image_records_sweep.txt 2>&1
timeout 600 python tools/mul_sweep.py $((81000 + ${3:-0})) 800 > $out/mul_sweep.txt 2>&1
for f in dense_sweep soup_sweep image_records_sweep mul_sweep; do echo "== $f"; grep -v amdgpu.ids $out/$f.txt | tail -6 | cut -c1-400; done
