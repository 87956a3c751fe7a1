#!/bin/bash
# round 6: the ablation ladder of k_raster_frag_group at cfg2 (development build in csrc_abl/, SMESH_RDBG: 2 = load + set-up + records,
# 4 = + coverage, 8 = + slot reservations, 1 = + depths and keys but no stores, 0 = everything; 16 = grouping without the atomics)
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6h
export SMESH_LIB_PATH=$root/semantic_meshes_amd/csrc_abl/libsmesh_hip.so
for d in 0 2 4 8 1 16; do
  ( cd /tmp && export TMPDIR=/tmp && SMESH_RDBG=$d timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/r6h/t$d -o t -- python $root/bench.py --steps 64 --warmup 8 --repeats 1 --no-cpu-baseline --no-host-path --no-pmc --no-group-pipeline > $root/gpurun_out/r6h/t$d.log 2>&1 )
  echo "SMESH_RDBG=$d: $(grep -h 'k_raster_frag_group' $root/gpurun_out/r6h/t$d/*/*kernel_stats.csv $root/gpurun_out/r6h/t$d/*kernel_stats.csv 2>/dev/null | head -1 | cut -d, -f1-4 | cut -c1-120)"
done 2>&1 | tee gpurun_out/r6h/ladder.txt
unset SMESH_LIB_PATH
bash tools/r6_trace_py.sh r6h/m10k $root/tools/medium_mesh_profile.py 100 50 2>&1 | tee gpurun_out/r6h/m10k.txt
bash tools/r6_trace_py.sh r6h/m900 $root/tools/medium_mesh_profile.py 30 15 2>&1 | tee gpurun_out/r6h/m900.txt
bash tools/r6_trace_py.sh r6h/close $root/tools/close_view_bench.py 300x150 2>&1 | tee gpurun_out/r6h/close.txt
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6h/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6h/pytest.log
tail -8 gpurun_out/r6h/pytest.log
