#!/bin/bash
root=${GRAFT_REPO_ROOT:-/root/repo}; cd $root; mkdir -p gpurun_out/r6j
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r6j/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6j/pytest.log
tail -5 gpurun_out/r6j/pytest.log
for x in 0 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && export TMPDIR=/tmp && SMESH_RASTER_XCD=$x timeout 300 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/r6j/x${x}_$c -o b -- python $root/bench.py --steps 16 --warmup 8 --repeats 1 --no-cpu-baseline --no-host-path --no-pmc --no-group-pipeline > $root/gpurun_out/r6j/x${x}_$c.log 2>&1 )
  done
done
python - <<'PY'
import csv, glob, collections
for x in (0, 8):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("gpurun_out/r6j/x%d_%s/**/*counter_collection.csv" % (x, c), recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                acc[n][c].append(float(r["Counter_Value"]))
    for n, d in acc.items():
        if "raster" in n or "resolve" in n or "project" in n:
            f_ = sum(d["FETCH_SIZE"]) / max(len(d["FETCH_SIZE"]), 1) * 1024 / 1e6
            w_ = sum(d["WRITE_SIZE"]) / max(len(d["WRITE_SIZE"]), 1) * 1024 / 1e6
            print("SMESH_RASTER_XCD=%d  %-36s FETCH %7.1f MB x2 + WRITE %7.1f MB = %7.1f MB per launch" % (x, n[:36], f_, w_, 2 * f_ + w_))
PY
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r6j/bench_driver.json 2> gpurun_out/r6j/bench_driver.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r6j/bench_driver.json') if l.startswith('{')][-1]); print('driver-shaped:', d['value'], d['config']['value_min'], d['config']['value_max'], d['roofline']['frac'], d['roofline']['frac_traffic'], d['roofline']['us_per_view'])"
