#!/bin/bash
# usage (GPU box): tools/grid_build_stats.sh <N>   -- build-kernel durations for one cloud size
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gb -o run -- python $R/tools/exp_grid_build.py $1 > $R/gpurun_out/gb.log 2>&1
f=$(ls $R/gpurun_out/gb/*/run_kernel_stats.csv $R/gpurun_out/gb/run_kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<PY
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1]))):
    if any(k in r["Name"] for k in ("grid_", "pack_points", "kp_order")):
        print(r["Name"][:48], r["Calls"], "avg %.1f min %.1f max %.1f us" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
rm -rf $R/gpurun_out/gb
