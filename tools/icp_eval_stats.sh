#!/bin/bash
# usage (GPU box): tools/icp_eval_stats.sh -- durations of the ICP evaluation launches of 24 end-to-end pairs: those that work and those that only find the stop flag set
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/icps -o run -- python $R/tools/bench_altlib.py --no-cpu-baseline --steps 1 --warmup 1 --e2e-pairs 24 --e2e-hard-pairs 0 --e2e-side-by-side 0 > $R/gpurun_out/icps.log 2>&1
f=$(ls $R/gpurun_out/icps/*/run_kernel_trace.csv $R/gpurun_out/icps/run_kernel_trace.csv 2>/dev/null | head -1)
python - "$f" <<PY
import csv,sys
rows=sorted((int(r["Start_Timestamp"]),(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in csv.DictReader(open(sys.argv[1])) if "icp_eval" in r["Kernel_Name"])
d=[x for _,x in rows]
print("in launch order (us):", [round(x) for x in d[-48:]])
d.sort()
import statistics
idle=[x for x in d if x<12]; act=[x for x in d if x>=12]
print("deciles (us):", [round(d[int(q*(len(d)-1)/10)],1) for q in range(11)])
print("icp_eval calls", len(d), "idle", len(idle), "median %.1f us" % (statistics.median(idle) if idle else 0), "active", len(act), "median %.1f us" % (statistics.median(act) if act else 0), "p90 %.1f" % (act[int(0.9*len(act))] if act else 0))
PY
rm -rf $R/gpurun_out/icps
