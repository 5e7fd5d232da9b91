#!/bin/bash
# usage (GPU box): tools/named_serial_trace.sh [n] [ragged] -- one pair's kernel sequence (start offsets, durations) of the serial named path
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
python $R/tools/exp_named_serial.py ${1:-64} $2 2>&1 | grep "named path"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ns -o run -- python $R/tools/exp_named_serial.py ${1:-64} $2 > $R/gpurun_out/ns.log 2>&1
f=$(ls $R/gpurun_out/ns/*/run_kernel_trace.csv $R/gpurun_out/ns/run_kernel_trace.csv 2>/dev/null | head -1)
GAP_SEQ=${GAP_SEQ:-20} python $R/tools/trace_gaps.py "$f" | sed 's/ume_coarse_h_kernel.*/coarse/'
rm -rf $R/gpurun_out/ns
