#!/bin/bash
# Round-4: the randomised differential tests over seeds beyond the committed ones, at the round's kernels (medium triangles inside the
# k_fuse_tri launch, launches left out by proof)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4sw; mkdir -p $out
timeout 1500 python tools/soup_sweep.py 20000 400 > $out/soup_sweep.txt 2>&1
timeout 600 python tools/image_records_sweep.py 31000 600 > $out/image_records_sweep.txt 2>&1
timeout 600 python tools/mul_sweep.py 41000 200 > $out/mul_sweep.txt 2>&1
tail -3 $out/soup_sweep.txt; tail -3 $out/image_records_sweep.txt; tail -3 $out/mul_sweep.txt
