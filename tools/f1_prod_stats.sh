#!/bin/bash
# usage (GPU box): tools/f1_prod_stats.sh [reps] [plain|hard|rot] -- rocprofv3 kernel stats of the production f1 path alone
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f1prod -o run -- python $R/tools/exp_f1_prod.py ${1:-5} ${2:-plain} > $R/gpurun_out/f1prod.log 2>&1
f=$(ls $R/gpurun_out/f1prod/*/run_kernel_stats.csv $R/gpurun_out/f1prod/run_kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<EOF
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(r["Name"][:64], r["Calls"], "avg %.3f min %.3f max %.3f ms" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6), r["Percentage"])
EOF
rm -rf $R/gpurun_out/f1prod
grep "^plain\|^hard\|^rot" $R/gpurun_out/f1prod.log
