tag=$1; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; tail -3 gpurun_out/$tag/pytest.log
timeout 600 python tools/big_configs.py cfg4 2>&1 | grep "pass\|coverage"
