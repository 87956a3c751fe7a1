# Round-2 evidence for profiles/: bench lines (driver's arguments and defaults), rocprofv3 kernel stats of the same commands,
# HBM counters of the dominant kernel (separate --pmc passes), the other single-GPU configs, the generic scatter path.
# usage (on the GPU box, through gpurun): bash tools/profile_round2.sh <tag>; results land in gpurun_out/<tag>/r02_*.
tag=$1; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
last() { grep '^{"metric' $1 | tail -1; }
python bench.py > $out/bench.log 2>&1; last $out/bench.log > $out/r02_bench.json
python bench.py --steps 20 --warmup 5 > $out/bench20.log 2>&1; last $out/bench20.log > $out/r02_bench_steps20_warmup5.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o bench -- python bench.py --no-cpu-baseline --no-host-path > $out/bench_kt.log 2>&1
last $out/bench_kt.log > $out/r02_bench_under_rocprof.json; cp $out/kt/bench_kernel_stats.csv $out/r02_bench_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt20 -o bench -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $out/bench_kt20.log 2>&1
last $out/bench_kt20.log > $out/r02_bench_steps20_under_rocprof.json; cp $out/kt20/bench_kernel_stats.csv $out/r02_bench_steps20_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -o bench -- python bench.py --no-cpu-baseline --no-host-path --steps 40 --warmup 8 > $out/pmc_$c.log 2>&1
done
SMESH_FUSE=strip python bench.py --no-cpu-baseline --no-host-path > $out/bench_strip.log 2>&1; last $out/bench_strip.log > $out/r02_bench_generic_scatter_path.json
SMESH_FUSE=strip timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_strip -o bench -- python bench.py --no-cpu-baseline --no-host-path > /dev/null 2>&1
cp $out/kt_strip/bench_kernel_stats.csv $out/r02_bench_kernel_stats_generic_scatter_path.csv
for w in cfg4 cfg5; do
  timeout 900 python bench.py --workload $w > $out/bench_$w.log 2>&1; last $out/bench_$w.log > $out/r02_bench_$w.json
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$w -o bench -- python bench.py --workload $w --no-host-path > $out/bench_kt_$w.log 2>&1
  cp $out/kt_$w/bench_kernel_stats.csv $out/r02_bench_${w}_kernel_stats.csv
done
# add() on index images the library did not render (image_records.hip): records + triangle-order fusion against the scatter-add
{
  echo "# python tools/generic_add_sweep.py (cfg2 geometry, device images and probs, ms per view) -- SMESH_ADD_RECORDS_MIN_C=0, then SMESH_ADD_RECORDS=0"
  SMESH_ADD_RECORDS_MIN_C=0 python tools/generic_add_sweep.py 2>&1 | grep ms/view
  SMESH_ADD_RECORDS=0 python tools/generic_add_sweep.py 2>&1 | grep ms/view
  echo "# python tools/generic_add_bench.py cfg5 8 -- default (records from 32 classes), then SMESH_ADD_RECORDS=0"
  python tools/generic_add_bench.py cfg5 8 2>&1 | grep "add()"
  SMESH_ADD_RECORDS=0 python tools/generic_add_bench.py cfg5 8 2>&1 | grep "add()"
} > $out/r02_foreign_images.txt
SMESH_ADD_RECORDS_MIN_C=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_fi -o gab -- python tools/generic_add_bench.py cfg2 16 > $out/gab_kt.log 2>&1
cp $out/kt_fi/gab_kernel_stats.csv $out/r02_foreign_images_cfg2_records_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_fi5 -o gab -- python tools/generic_add_bench.py cfg5 8 > $out/gab_kt5.log 2>&1
cp $out/kt_fi5/gab_kernel_stats.csv $out/r02_foreign_images_cfg5_kernel_stats.csv
# class-count sweep at cfg2's geometry through fuse_views (every triangle-order kernel), groups of eight views and one launch per view
{
  echo "# python tools/class_sweep_views.py <C ...>: fuse_views at cfg2 geometry (1 M triangles, 1920x1080), rasteriser included, groups of eight views"
  python tools/class_sweep_views.py 5 13 19 21 27 32 40 41 48 49 64 100 127 128 150 256 512 2>&1 | grep "C ="
  echo "# the same with one fusion launch per view (SMESH_FUSE_VIEWS=1)"
  SMESH_FUSE_VIEWS=1 python tools/class_sweep_views.py 5 19 40 48 64 100 127 150 256 2>&1 | grep "C ="
} > $out/r02_class_count_sweep.txt
# the group pipeline (opt-in): bench line, kernel stats and a trace excerpt showing the rasteriser of group g+1 beside the fusion of group g
python bench.py --group-pipeline --no-cpu-baseline --no-host-path > $out/bench_gp.log 2>&1; last $out/bench_gp.log > $out/r02_bench_group_pipeline.json
python bench.py --group-pipeline --steps 20 --warmup 5 --no-cpu-baseline --no-host-path > $out/bench_gp20.log 2>&1; last $out/bench_gp20.log > $out/r02_bench_group_pipeline_steps20.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_gp -o bench -- python bench.py --group-pipeline --steps 80 --warmup 16 --no-cpu-baseline --no-host-path > $out/bench_kt_gp.log 2>&1
cp $out/kt_gp/bench_kernel_stats.csv $out/r02_bench_group_pipeline_kernel_stats.csv
OUT=$out python - <<'PY' > $out/r02_group_pipeline_trace_excerpt.txt
import csv, os
rows = list(csv.DictReader(open(os.environ["OUT"] + "/kt_gp/bench_kernel_trace.csv")))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_fuse_tri<19, 0, true, 8>" in r["Kernel_Name"]]
i0 = idx[len(idx) // 2]
t0 = int(rows[i0 - 6]["Start_Timestamp"])
print("# rocprofv3 --kernel-trace of `python bench.py --group-pipeline --steps 80 --warmup 16`: consecutive dispatches in the middle of the timed loop (us)")
for r in rows[i0 - 6:i0 + 11]:
    b = int(r["Start_Timestamp"]); e = int(r["End_Timestamp"])
    print("%-40s start %8.1f  end %8.1f  duration %7.1f  queue %s" % (r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40], (b - t0) / 1e3, (e - t0) / 1e3, (e - b) / 1e3, r.get("Queue_Id", "?")))
PY
OUT=$out python - <<'PY'
import csv, collections, json, os
out = os.environ["OUT"]
pm = {}
for kind in ("FETCH_SIZE", "WRITE_SIZE"):
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(out + "/pmc_%s/bench_counter_collection.csv" % kind)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[n] += float(r["Counter_Value"]); cnt[n] += 1
    for n in acc:
        if "synth" in n or "rocclr" in n: continue
        pm.setdefault(n, {})[kind + "_KiB_avg_per_launch"] = round(acc[n] / cnt[n], 1)
for n, d in pm.items():
    if "FETCH_SIZE_KiB_avg_per_launch" in d and "WRITE_SIZE_KiB_avg_per_launch" in d:
        # MI355X_MICROARCH.md, HBM: FETCH_SIZE reports half of the bytes of wide reads on gfx950 -> doubled; WRITE_SIZE as reported
        d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE_KiB_avg_per_launch"] + d["WRITE_SIZE_KiB_avg_per_launch"]) * 1024)
json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --steps 40 --warmup 8`, "
           "averages per launch; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 correction of the guide)", "kernels": pm},
          open(out + "/r02_pmc_hbm_summary.json", "w"), indent=1)
b = json.load(open(out + "/r02_bench.json"))
print(b["value"], b["roofline"]["frac"], b["roofline"]["avg_launch_us"], b.get("cpu_baseline", {}).get("value"))
for n, d in pm.items():
    print(n[:60], d)
PY
