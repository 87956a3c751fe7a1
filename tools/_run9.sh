tag=$1; mkdir -p gpurun_out/$tag
timeout 1500 python -m pytest tests -m gpu -x -q -k "triangle_order or any_class or threshold" > gpurun_out/$tag/pytest.log 2>&1; tail -2 gpurun_out/$tag/pytest.log
for d in 0 14; do echo -n "FDBG=$d: "; SMESH_FDBG=$d timeout 600 python tools/big_configs.py cfg5 2>&1 | grep "pass 1"; done
