#!/bin/bash
# usage (build container): tools/exp_coarse_ablate.sh build   -> tools/libumereg_ca<mask>.so for a few masks
#       (GPU box):         tools/exp_coarse_ablate.sh run     -> coarse-stage time of each variant on the KT pair
# masks (UMEREG_COARSE_ABLATE in subspace_dist.hip; results are wrong by construction): 1 no squares, 2 no filter, 4 no MFMAs, 8 no LDS reads
cd "$(dirname "$0")/.."
MASKS="${MASKS:-0 1 2 3 4 6 8 11}"
if [ "$1" = build ]; then
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xarch_device -fno-slp-vectorize -fPIC -shared \
      -fvisibility=hidden -DUMEREG_COARSE_ABLATE=$m -I include umeregrobust_amd/csrc/*.hip -o tools/libumereg_ca$m.so &
  done
  wait
else
  for m in $MASKS; do echo -n "mask $m: "; ALTLIB=libumereg_ca$m.so timeout 120 python tools/exp_f16r_stats.py 2>&1 | grep "^coarse"; done
fi
