#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in 8 16 32 64; do echo "SMESH_MID_WAVES=$w"; SMESH_MID_WAVES=$w python tools/mesh_density_sweep.py 2>&1 | grep triangles | cut -c1-100; done
