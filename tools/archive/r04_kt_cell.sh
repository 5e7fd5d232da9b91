#!/bin/bash
# usage (GPU box): tools/r04_kt_cell.sh <outfile> <lib1> <lib2> ...  -- KITTI-test sizes: the leftovers through the queue (def), through the
# lattice + list kernel (defL) and through the lattice + cell pass (defPL), for several builds of the library ("" = shipped)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; shift
mkdir -p $(dirname $O); : > $O
cd $R
for lib in "$@"; do
  for cfg in ${CFGS:-KT}; do
    echo "---- lib ${lib:-shipped} $cfg" >> $O
    ALTLIB=$lib timeout 600 python tools/exp_f1_v2.py 5 plain,hard ${VARS:-def,defL,defPL} $cfg 2>&1 | grep "^plain\|^hard" | cut -c1-${CUT:-400} >> $O
  done
done
cat $O
