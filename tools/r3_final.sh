# GPU session: whole -m gpu suite, then the bench lines for profiles/
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
timeout 2700 python -m pytest tests -q -m gpu --maxfail=12 --durations=6 -o faulthandler_timeout=300 2>&1 | tail -60 > $out/suite.log
grep -E "^FAILED|^ERROR|passed|failed|^E  " $out/suite.log | head -40
last() { grep '^{"metric' $1 | tail -1; }
timeout 600 python bench.py --steps 20 --warmup 5 > $out/bench20.log 2>&1; last $out/bench20.log > $out/r03_bench_steps20_warmup5.json
timeout 600 python bench.py > $out/bench.log 2>&1; last $out/bench.log > $out/r03_bench.json
python - <<PY
import json
for f in ("r03_bench_steps20_warmup5.json", "r03_bench.json"):
    d = json.load(open("$out/" + f))
    fi = d.get("foreign_images", {})
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"], fi.get("ms_per_view"), fi.get("frac"), fi.get("path"))
PY
