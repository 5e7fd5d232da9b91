#!/bin/bash
# usage (GPU box): tools/f1_v2_stats.sh <reps> <plain|hard|rot> <variant> <config>  -- rocprofv3 kernel stats of tools/exp_f1_v2.py (one variant)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f1v2 -o run -- python $R/tools/exp_f1_v2.py ${1:-3} ${2:-plain} ${3:-def} ${4:-KT} > $R/gpurun_out/f1v2.log 2>&1
f=$(ls $R/gpurun_out/f1v2/*/run_kernel_stats.csv $R/gpurun_out/f1v2/run_kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:${5:-16}]:
    print(r["Name"][:64], r["Calls"], "avg %.3f min %.3f max %.3f ms" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6), r["Percentage"])
PY
rm -rf $R/gpurun_out/f1v2
grep "^plain\|^hard\|^rot" $R/gpurun_out/f1v2.log | cut -c1-120
