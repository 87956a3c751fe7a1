# usage: bash tools/pmc_hbm.sh <workload> [bench args ...]  -- HBM bytes per launch of every kernel of a short bench run:
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (the TCC block has 4 counter slots, FETCH_SIZE takes 3 and
# WRITE_SIZE 2: MI355X_MICROARCH.md "rocprofv3 PMC slots"; asking for both in one pass aborts inside rocprofiler and leaves the
# process hanging).  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (the guide's gfx950 correction: FETCH_SIZE
# counts 64 B per 128-byte request).  Writes gpurun_out/pmc_hbm_<workload>/summary.json.
w=$1; shift; out=gpurun_out/pmc_hbm_$w; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 420 rocprofv3 --pmc $c --output-format csv -d $out/$c -o bench -- python bench.py --workload $w --no-cpu-baseline --no-host-path --steps 16 --warmup 8 "$@" > $out/$c.log 2>&1 || echo "pass $c failed (see $out/$c.log)"
done
OUT=$out W=$w python - <<'PY'
import csv, collections, json, os
out = os.environ["OUT"]
pm = {}
for kind in ("FETCH_SIZE", "WRITE_SIZE"):
    path = out + "/%s/bench_counter_collection.csv" % kind
    if not os.path.exists(path):
        continue
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        acc[n] += float(r["Counter_Value"]); cnt[n] += 1
    for n in acc:
        if "synth" in n or "rocclr" in n: continue
        pm.setdefault(n, {})[kind + "_KiB_avg_per_launch"] = round(acc[n] / cnt[n], 1)
        pm[n]["launches"] = cnt[n]
for n, d in pm.items():
    if "FETCH_SIZE_KiB_avg_per_launch" in d and "WRITE_SIZE_KiB_avg_per_launch" in d:
        d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE_KiB_avg_per_launch"] + d["WRITE_SIZE_KiB_avg_per_launch"]) * 1024)
json.dump({"_note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `python bench.py --workload %s --steps 16 --warmup 8`, "
           "averages per launch; hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 (gfx950 correction of MI355X_MICROARCH.md)" % os.environ["W"],
           "kernels": pm}, open(out + "/summary.json", "w"), indent=1)
for n, d in sorted(pm.items(), key=lambda kv: -kv[1].get("hbm_bytes_per_launch", 0)):
    print("%-64s %s" % (n[:64], {k: v for k, v in d.items()}))
PY
