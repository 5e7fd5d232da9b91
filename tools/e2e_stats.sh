#!/bin/bash
# usage (GPU box): tools/e2e_stats.sh [extra bench args] -- rocprofv3 kernel stats of bench.py's end-to-end leg alone (plain KT pairs)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/e2eprof -o run -- python $R/bench.py --steps 1 --warmup 1 --pairs-per-step 4 --no-cpu-baseline --e2e-hard-pairs 0 --detail $R/gpurun_out/e2eprof_detail.json "$@" > $R/gpurun_out/e2eprof.log 2>&1
f=$(ls $R/gpurun_out/e2eprof/*/run_kernel_stats.csv $R/gpurun_out/e2eprof/run_kernel_stats.csv 2>/dev/null | head -1)
python - "$f" <<EOF
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.1f ms, %d launches" % (tot/1e6, sum(int(r["Calls"]) for r in rows)))
for r in rows[:45]:
    print(r["Name"][:80], r["Calls"], "tot %.2f ms avg %.3f ms" % (float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6), r["Percentage"])
EOF
rm -rf $R/gpurun_out/e2eprof
python - $R/gpurun_out/e2eprof_detail.json <<PY
import json,sys
d=json.load(open(sys.argv[1])); e=d['end_to_end']; print(e['pairs'], e['pairs_per_s'], e['stage_ms'])
PY
