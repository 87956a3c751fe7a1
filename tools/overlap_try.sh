tag=$1; shift; out=gpurun_out/$tag; mkdir -p $out; cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 80 --warmup 16 --no-cpu-baseline --no-host-path | grep -o '"value": [0-9.]*'; }
for pad in 20000 24576 28000; do
  for prio in none high low; do
    echo "pad $pad prio $prio: pipeline $(SMESH_RASTER_PRIO=$prio SMESH_GROUP_PIPELINE=1 SMESH_FUSE_LDS_PAD=$pad run)"
  done
done
