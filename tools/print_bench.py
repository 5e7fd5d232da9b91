"""usage: python tools/print_bench.py <bench log>  -- the headline numbers of a bench.py JSON line"""
import json
import sys

d = None
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
print("value", d["value"], d["unit"])
for k in ("end_to_end", "end_to_end_hard"):
    if k in d:
        e = d[k]
        print(k, e["pairs_per_s"], e["stage_ms"], "side by side", e.get("side_by_side", {}).get("pairs_per_s"), "evaluate_pairs", e.get("evaluate_pairs_loop", {}).get("pairs_per_s"))
f = d.get("f1_selection", {})
for k in ("plain", "hard"):
    if k in f:
        print("f1", k, f[k]["stage_ms"])
