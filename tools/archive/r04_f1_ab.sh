#!/bin/bash
# usage (GPU box): tools/r04_f1_ab.sh <outdir> [base lib under tools/]  -- f1 A/B of the shipped library against another build of it
# (ms per corr_scores call at KITTI-test and nuScenes-test sizes, plain and half-overlapping pair), then rocprofv3 per-kernel times of
# the shipped library at nuScenes-test sizes and of the cell pass's ablation builds (tools/libumereg_abl*.so) if present
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-f1ab}
B=${2:-libumereg_r04base.so}
mkdir -p $O
cd $R
for cfg in KT NS; do
  v=def; [ $cfg = NS ] && v=defB
  echo "---- $cfg base ($B)" >> $O/ab.txt
  ALTLIB=$B timeout 600 python tools/exp_f1_v2.py 5 plain,hard $v $cfg 2>&1 | grep "^plain\|^hard" | cut -c1-150 >> $O/ab.txt
  echo "---- $cfg new" >> $O/ab.txt
  timeout 600 python tools/exp_f1_v2.py 5 plain,hard $v $cfg 2>&1 | grep "^plain\|^hard" | cut -c1-150 >> $O/ab.txt
done
for kind in plain hard; do
  echo "---- NS $kind per kernel (new)" >> $O/ab.txt
  bash tools/f1_v2_stats.sh 3 $kind defB NS 14 >> $O/ab.txt 2>&1
done
for m in 200000 400000; do
  if [ -f tools/libumereg_abl$m.so ]; then
    echo "---- NS plain per kernel, ablation $m (200000: cell pass without its epilogue; 400000: its sweeps over 4 candidates)" >> $O/ab.txt
    ALTLIB=libumereg_abl$m.so bash tools/f1_v2_stats.sh 3 plain defB NS 6 >> $O/ab.txt 2>&1
  fi
done
cat $O/ab.txt
