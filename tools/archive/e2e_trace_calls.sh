#!/bin/bash
# usage (GPU box): tools/e2e_trace_calls.sh <kernel substring> [bench args]  -- per-call durations of one kernel in bench.py's end-to-end leg
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tc -o run -- python $R/bench.py --steps 1 --warmup 1 --pairs-per-step 4 --no-cpu-baseline --e2e-hard-pairs 0 --pool 4 --resident-steps 0 --e2e-side-by-side 0 "$@" > $R/gpurun_out/tc.log 2>&1
f=$(ls $R/gpurun_out/tc/*/run_kernel_trace.csv $R/gpurun_out/tc/run_kernel_trace.csv 2>/dev/null | head -1)
for k in $K; do echo "==== $k"; python $R/tools/kernel_calls.py "$f" $k 60; done
rm -rf $R/gpurun_out/tc
