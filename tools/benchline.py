import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["config"]["workload"][:6], d["value"], d["roofline"]["us_per_view"], d["roofline"]["frac"], d["ms_per_step"])
