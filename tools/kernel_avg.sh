#!/bin/bash
# usage (GPU box): tools/kernel_avg.sh <script> [args...] -- rocprofv3 --kernel-trace --stats of a script, average duration per kernel (top 14)
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ka_out
(cd $W && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ka_out -o run -- python "$@" > /tmp/ka.log 2>&1)
f=$(ls /tmp/ka_out/*/run_kernel_stats.csv /tmp/ka_out/run_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print(f'{float(r["AverageNs"]) / 1e3:9.1f} us avg  x{int(r["Calls"]):5d}  {float(r["Percentage"]):5.1f} %  {r["Name"][:90]}')
PY
