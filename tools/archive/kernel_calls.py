"""Per-call durations of the kernels whose name contains a substring, from a rocprofv3 --kernel-trace CSV: grid / workgroup size and
duration of every call in launch order (first N), and the distribution by grid size.
usage: python tools/kernel_calls.py <run_kernel_trace.csv> <substring> [n_first=40]"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_first = int(sys.argv[3]) if len(sys.argv) > 3 else 40
by = collections.defaultdict(list)
for i, r in enumerate(rows):
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = (r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Workgroup_Size_X", "?"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "?")))
    by[g].append(d)
    if i < n_first:
        print(f"{i:4d} grid {g[0]:>8s} x {g[1]:>3s} wg {g[2]:>5s} lds {g[3]:>6s}  {d:9.1f} us")
print()
for g, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print(f"grid {g[0]:>8s} x {g[1]:>3s} wg {g[2]:>5s}: {len(v):4d} calls  min {v[0]:8.1f}  median {v[len(v) // 2]:8.1f}  max {v[-1]:8.1f}  total {sum(v) / 1e3:8.2f} ms")
