# Round-3 evidence for profiles/ (run on the GPU box through gpurun): bash tools/profile_round3.sh <tag>; results land in gpurun_out/<tag>/r03_*.
tag=$1; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
last() { grep '^{"metric' $1 | tail -1; }
# bench lines: defaults, the driver's arguments, with in-run PMC traffic
timeout 600 python bench.py > $out/bench.log 2>&1; last $out/bench.log > $out/r03_bench.json
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench20.log 2>&1; last $out/bench20.log > $out/r03_bench_steps20_warmup5.json
timeout 900 python bench.py --pmc --no-cpu-baseline --no-host-path > $out/bench_pmc.log 2>&1; last $out/bench_pmc.log > $out/r03_bench_pmc.json
# rocprofv3 kernel stats of the same commands
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o bench -- python bench.py --no-cpu-baseline --no-host-path > $out/bench_kt.log 2>&1
last $out/bench_kt.log > $out/r03_bench_under_rocprof.json; cp $out/kt/bench_kernel_stats.csv $out/r03_bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt20 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $out/bench_kt20.log 2>&1
cp $out/kt20/bench_kernel_stats.csv $out/r03_bench_steps20_kernel_stats.csv
# HBM counters, separate passes, per workload
for w in cfg2 cfg4 cfg5; do
  bash tools/pmc_hbm.sh $w > $out/pmc_$w.txt 2>&1; cp gpurun_out/pmc_hbm_$w/summary.json $out/r03_pmc_hbm_$w.json
done
# the other single-GPU configs
for w in cfg4 cfg4t cfg5; do
  timeout 900 python bench.py --workload $w > $out/bench_$w.log 2>&1; last $out/bench_$w.log > $out/r03_bench_$w.json
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -o bench -- python bench.py --workload $w --no-host-path > $out/bench_kt_$w.log 2>&1
  cp $out/kt_$w/bench_kernel_stats.csv $out/r03_bench_${w}_kernel_stats.csv
done
# generic scatter path forced, foreign images
SMESH_FUSE=strip timeout 600 python bench.py --no-cpu-baseline --no-host-path > $out/bench_strip.log 2>&1; last $out/bench_strip.log > $out/r03_bench_generic_scatter_path.json
# permuted class vectors, medium meshes
{ python tools/permuted_probs_bench.py 2>&1 | grep "render + add"; echo "# SMESH_STRIDED_PROBS=0 (gathered copy, round 2):"; SMESH_STRIDED_PROBS=0 python tools/permuted_probs_bench.py 2>&1 | grep "render + add"; } > $out/r03_permuted_probs.txt
{ python tools/mesh_density_sweep.py 2>&1 | grep triangles; echo "# SMESH_FUSE_MID=0 (one wave per queued triangle, round 2):"; SMESH_FUSE_MID=0 python tools/mesh_density_sweep.py 2>&1 | grep triangles; } > $out/r03_mesh_density_sweep.txt
ls $out | grep r03_
