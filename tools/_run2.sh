tag=$1; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; tail -3 gpurun_out/$tag/pytest.log
for p in 0 1; do SMESH_FUSE_PIPELINE=$p python bench.py --no-cpu-baseline --steps 200 > gpurun_out/$tag/bench_p$p.log 2>&1; echo "pipeline=$p"; grep -o '"value": [0-9.]*\|"avg_launch_us": [0-9.]*\|"frac": [0-9.]*' gpurun_out/$tag/bench_p$p.log | tr '\n' ' '; echo; done
