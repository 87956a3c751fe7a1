tag=$1; mkdir -p gpurun_out/$tag
for d in 0 1 2 4 8 6 14; do echo -n "FDBG=$d: "; SMESH_FDBG=$d timeout 600 python tools/big_configs.py cfg5 2>&1 | grep "pass 1"; done
