#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag>
# Collects the evidence the bench line's roofline object refers to, into gpurun_out/<tag>/:
#   bench_default.json           default bench run (with the CPU baseline leg)
#   kernel_stats.csv             rocprofv3 --kernel-trace --stats of the same command (no CPU leg)
#   pmc_{FETCH,WRITE}_SIZE.csv   separate --pmc passes (kernel-trace only, as gpurun requires)
#   bench_with_selection.json    named path + f1 hypothesis selection
#   bench_end_to_end.json        + f2 ICP; kernel_stats_end_to_end.csv: rocprofv3 stats of that command
TAG=${1:-profiles_run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 600 python $ROOT/bench.py > $OUT/bench_default.log 2>&1; tail -1 $OUT/bench_default.log > $OUT/bench_default.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python $ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1
f=$(ls $OUT/stats/*/run_kernel_stats.csv $OUT/stats/run_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py --steps 6 --warmup 2 --depth 1 --no-cpu-baseline > $OUT/pmc_$c.log 2>&1
  f=$(ls $OUT/pmc_$c/*/pmc_counter_collection.csv $OUT/pmc_$c/pmc_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc_$c.csv
done
timeout -s KILL 600 python $ROOT/bench.py --no-cpu-baseline --with-selection --steps 40 --warmup 5 > $OUT/bench_with_selection.log 2>&1; tail -1 $OUT/bench_with_selection.log > $OUT/bench_with_selection.json
timeout -s KILL 600 python $ROOT/bench.py --no-cpu-baseline --with-selection --with-refinement --steps 40 --warmup 5 > $OUT/bench_end_to_end.log 2>&1; tail -1 $OUT/bench_end_to_end.log > $OUT/bench_end_to_end.json
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_e2e -o run -- python $ROOT/bench.py --no-cpu-baseline --with-selection --with-refinement --steps 20 --warmup 3 > $OUT/stats_e2e.log 2>&1
f=$(ls $OUT/stats_e2e/*/run_kernel_stats.csv $OUT/stats_e2e/run_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_end_to_end.csv
rm -rf $OUT/stats_e2e
rm -rf $OUT/stats $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
ls -la $OUT
