#!/bin/bash
# usage (GPU box): tools/f1_stats.sh [reps]  -- rocprofv3 kernel stats of tools/exp_f1_lattice.py + its own summary lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/f1lat -o run -- python $R/tools/exp_f1_lattice.py ${1:-3} > $R/gpurun_out/f1lat.log 2>&1
f=$(ls $R/gpurun_out/f1lat/*/run_kernel_stats.csv $R/gpurun_out/f1lat/run_kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<EOF
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:10]:
    print(r["Name"][:64], r["Calls"], "avg %.3f min %.3f max %.3f ms" % (float(r["AverageNs"])/1e6, float(r["MinNs"])/1e6, float(r["MaxNs"])/1e6), r["Percentage"])
EOF
rm -rf $R/gpurun_out/f1lat
grep "^plain\|^hard" $R/gpurun_out/f1lat.log | cut -c1-700
