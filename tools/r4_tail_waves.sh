#!/bin/bash
# how many (empty, at cfg2) tail workgroups does the k_fuse_tri launch carry?  SMESH_BIG_WAVES x SMESH_MID_WAVES per CU
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in "16 32" "8 16" "4 8" "2 4" "1 1"; do set -- $cfg
  v=$(SMESH_BIG_WAVES=$1 SMESH_MID_WAVES=$2 python bench.py --no-cpu-baseline --no-host-path --no-pmc 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print(d['value'], d['roofline']['avg_launch_us'])")
  echo "big $1 mid $2: $v"
done; done
