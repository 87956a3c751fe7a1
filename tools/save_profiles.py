"""Copies the summaries of a `tools/profile_round.sh <tag>` run from gpurun_out/<tag>/ into profiles/ (tracked)."""
import json, shutil, sys
tag = sys.argv[1]
out = "gpurun_out/" + tag
last = lambda path: [l for l in open(path) if l.startswith('{"metric')][-1]
open("profiles/r01_bench.json", "w").write(last(out + "/bench.log"))
open("profiles/r01_bench_under_rocprof.json", "w").write(last(out + "/bench_kt.log"))
open("profiles/r01_bench_generic_scatter_path.json", "w").write(last(out + "/bench_strip.log"))
shutil.copy(out + "/kt/bench_kernel_stats.csv", "profiles/r01_bench_kernel_stats.csv")
pm = json.load(open(out + "/summary.json"))["pmc"]
old = json.load(open("profiles/r01_pmc_hbm_summary.json"))
old["kernels"] = {k: v for k, v in pm.items() if "synth" not in k and "rocclr" not in k}
json.dump(old, open("profiles/r01_pmc_hbm_summary.json", "w"), indent=1)
ft = json.load(open("profiles/fusion_traffic.json"))
for suffix, key in ((", 1>", "k_fuse_tri"), (", 2>", "k_fuse_tri_pair")):   # one view per launch / two (fuse_views)
    ks = [v for n, v in pm.items() if n.startswith("void k_fuse_tri<19, 0") and n.split("(")[0].endswith(suffix)]
    if not ks:
        continue
    k = ks[0]
    ft.setdefault(key, {}).update({"hbm_bytes_per_launch": int((2 * k["FETCH_SIZE_KiB_avg"] + k["WRITE_SIZE_KiB_avg"]) * 1024),
                                   "FETCH_SIZE_KiB": round(k["FETCH_SIZE_KiB_avg"], 1), "WRITE_SIZE_KiB": round(k["WRITE_SIZE_KiB_avg"], 1)})
json.dump(ft, open("profiles/fusion_traffic.json", "w"), indent=1)
b = json.loads(last(out + "/bench.log"))
print(b["value"], b["roofline"]["frac"], b["roofline"]["avg_launch_us"], b["cpu_baseline"]["value"], b["cpu_baseline"]["optimised_cpu"]["value"])
