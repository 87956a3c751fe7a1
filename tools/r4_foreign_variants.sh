#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp semantic_meshes_amd/csrc/libsmesh_hip.so /tmp/keep.so
for rep in 1 2; do
for v in "$@"; do
  cp tools/_variants/$v.so semantic_meshes_amd/csrc/libsmesh_hip.so
  echo "$v: $(python tools/generic_add_bench.py cfg2 16 2>&1 | grep add | cut -c1-120)"
  echo "$v: $(python tools/mesh_density_sweep.py 2>&1 | grep '1000000 triangles')"
done; done
cp /tmp/keep.so semantic_meshes_amd/csrc/libsmesh_hip.so
