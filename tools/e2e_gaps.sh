#!/bin/bash
# usage (GPU box): tools/e2e_gaps.sh -- idle gaps between kernels in the pair-by-pair end-to-end leg (48 pairs): GPU busy / span per pair, the largest gaps by (kernel before, kernel after)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gp -o run -- python $R/bench.py --no-cpu-baseline --steps 1 --warmup 1 --e2e-pairs 48 --e2e-hard-pairs 0 --e2e-side-by-side 0 --pool 4 --resident-steps 0 $@ > $R/gpurun_out/gp.log 2>&1
f=$(ls $R/gpurun_out/gp/*/run_kernel_trace.csv $R/gpurun_out/gp/run_kernel_trace.csv 2>/dev/null | head -1)
python $R/tools/trace_gaps.py "$f" | sed 's/ume_coarse_h_kernel.*/coarse/' | head -${GAP_LINES:-34}
rm -rf $R/gpurun_out/gp
