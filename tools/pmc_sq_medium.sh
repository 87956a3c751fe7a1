# SQ counters of the rasteriser on one view at a time of a medium mesh (tools/medium_mesh_profile.py a b single), and of cfg2 for comparison
# usage: bash tools/pmc_sq_medium.sh <tag> [a b]
tag=$1; a=${2:-200}; b=${3:-100}; out=gpurun_out/$tag; mkdir -p $out; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for mode in single group; do
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/sq_$mode -o b -- python tools/medium_mesh_profile.py $a $b $mode > $out/sq_$mode.log 2>&1
echo "== $a x $b quads, $mode"
OUT=$out/sq_$mode python - <<'PY'
import csv, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(os.environ["OUT"] + "/b_counter_collection.csv")):
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    acc[n][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": cnt[n] += 1
for n, d in acc.items():
    if "synth" in n or "rocclr" in n: continue
    k = cnt[n] or 1
    wc = d["SQ_WAVE_CYCLES"] or 1
    print("%-44s launches %3d waves %9.0f  per wave: quad-cycles %7.0f  parked %4.1f%%  issue-stall %4.1f%%  issuing %4.1f%% (VALU %4.1f%%)  VALU insts %6.0f  busy %8.0f" % (
        n[:44], k, d["SQ_WAVES"] / k, wc / d["SQ_WAVES"], 100 * d["SQ_WAIT_ANY"] / wc, 100 * d["SQ_WAIT_INST_ANY"] / wc,
        100 * d["SQ_ACTIVE_INST_ANY"] / wc, 100 * d["SQ_ACTIVE_INST_VALU"] / wc, d["SQ_INSTS_VALU"] / d["SQ_WAVES"], d["SQ_BUSY_CYCLES"] / k))
PY
rm -rf $out/sq_$mode
done
