#!/bin/bash
# usage (GPU box): tools/r04_f1_variants.sh <outfile> <lib1> <lib2> ...   ("" = the shipped library) -- ms per corr_scores call of several
# builds of the library at KITTI-test and nuScenes-test sizes, plain and half-overlapping pair, two rounds (box drift shows)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $(dirname $O); : > $O
cd $R
for round in 1 2; do
  for lib in "$@"; do
    for cfg in ${CFGS:-KT NS}; do
      v=def; [ $cfg != KT ] && v=defB
      echo "---- round $round lib ${lib:-shipped} $cfg" >> $O
      ALTLIB=$lib timeout 600 python tools/exp_f1_v2.py 5 plain,hard $v $cfg 2>&1 | grep "^plain\|^hard" | cut -c1-110 >> $O
    done
  done
done
cat $O
