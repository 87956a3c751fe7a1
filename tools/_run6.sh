tag=$1; mkdir -p gpurun_out/$tag
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 200 > gpurun_out/$tag/b_prof.log 2>&1; echo -n "events on:  "; grep -o '"value": [0-9.]*' gpurun_out/$tag/b_prof.log
SMESH_BENCH_NO_PROFILE=1 python bench.py --no-cpu-baseline --steps 200 > gpurun_out/$tag/b_noprof.log 2>&1; echo -n "events off: "; grep -o '"value": [0-9.]*' gpurun_out/$tag/b_noprof.log
done
