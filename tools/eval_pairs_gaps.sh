#!/bin/bash
# usage (GPU box): tools/eval_pairs_gaps.sh [n] -- GPU busy time against wall time per pair in evaluate.evaluate_pairs (the default user loop), largest idle gaps
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python $R/tools/exp_eval_pairs_loop.py ${1:-64} 2>&1 | grep evaluate_pairs
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gp -o run -- python $R/tools/exp_eval_pairs_loop.py ${1:-64} > $R/gpurun_out/gp.log 2>&1
f=$(ls $R/gpurun_out/gp/*/run_kernel_trace.csv $R/gpurun_out/gp/run_kernel_trace.csv 2>/dev/null | head -1)
GAP_SEQ=0 python $R/tools/trace_gaps.py "$f" | sed 's/ume_coarse_h_kernel.*/coarse/' | head -${GAP_LINES:-32}
rm -rf $R/gpurun_out/gp
