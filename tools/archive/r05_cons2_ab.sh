#!/bin/bash
# usage (GPU box): tools/r05_cons2_ab.sh <outfile> <lib ...>  ("" = the shipped library; others are files under tools/) -- per-stage ms of the
# f1 call (HIP events inside umereg_corr_scores_profile_f32: structures / consensus / leftovers / reduction) on the plain and the
# half-overlapping KT pair, for several builds of the library, two rounds.  Builds: _build.build_native(extra_flags=[...], out="tools/lib<name>.so")
# with -DUMEREG_CONS2_PERSIST=0|1 (persistent wavefronts of the consensus pass) and -DUMEREG_CONS2_BLOCK_WAVES=1|2.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
: > $O
cd $R
for round in 1 2; do
  for lib in "$@"; do
    echo "---- round $round lib ${lib:-shipped}" >> $O
    ALTLIB=$lib timeout 300 python tools/exp_f1_stage.py 10 >> $O 2>&1
  done
done
cat $O
