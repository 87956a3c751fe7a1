#!/bin/bash
# Round 5: which mix of rasteriser and fusion waves per CU serves the group pipeline best?  LDS pads cap the workgroups per CU of
# k_raster_frag_group (256 lanes) and of the eight-view k_fuse_tri (64 lanes).  cfg2, 200 steps, median of 5 regions.
out=gpurun_out/r5mix; mkdir -p $out; cd /root/repo
for rp in 0 40960 49152; do for fp in 0 8192 16384 24576 40960; do
  SMESH_RASTER_LDS_PAD=$rp SMESH_FUSE_LDS_PAD=$fp python bench.py --no-cpu-baseline --no-pmc --no-host-path --repeats 5 > $out/r${rp}_f${fp}.json 2>/dev/null
  python - $out/r${rp}_f${fp}.json $rp $fp <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l); print("raster pad %6s fusion pad %6s: %8.1f views/s (min %8.1f max %8.1f)" % (sys.argv[2], sys.argv[3], d["value"], d["config"]["value_min"], d["config"]["value_max"]))
PY
done; done
python bench.py --no-cpu-baseline --no-pmc --no-host-path --repeats 5 --views-per-call 16 > $out/vpc16.json 2>/dev/null; python -c "
import json
for l in open('$out/vpc16.json'):
    if l.startswith('{\"metric\"'): d=json.loads(l); print('views-per-call 16:', d['value'])"
