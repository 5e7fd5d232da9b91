"""usage: python tools/print_bench.py <bench log | bench_detail.json>  -- the headline numbers of a bench.py run"""
import json
import os
import sys

d = None
if sys.argv[1].endswith(".json"):          # a bench_detail.json (the full result)
    d = json.load(open(sys.argv[1]))
else:
    for l in open(sys.argv[1]):
        if l.startswith("{"):
            d = json.loads(l)
    if d and d.get("detail") and os.path.exists(d["detail"]):   # the line is a bounded extract: the full result is in the file it names
        d = json.load(open(d["detail"]))
print("value", d["value"], d["unit"])
for k in ("end_to_end", "end_to_end_hard"):
    if k in d:
        e = d[k]
        print(k, e["pairs_per_s"], e["stage_ms"], "side by side", e.get("side_by_side", {}).get("pairs_per_s"), "evaluate_pairs", e.get("evaluate_pairs_loop", {}).get("pairs_per_s"))
f = d.get("f1_selection", {})
for k in ("plain", "hard"):
    if k in f:
        print("f1", k, f[k]["stage_ms"])
