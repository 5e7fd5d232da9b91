#!/bin/bash
# usage (build container): tools/exp_f1_ablate.sh build   -> tools/libumereg_abl<mask>.so for a few masks
#       (GPU box):         tools/exp_f1_ablate.sh run     -> lattice-kernel time of each variant on the KT pair
cd "$(dirname "$0")/.."
MASKS="${MASKS:-0 8192}"
if [ "$1" = build ]; then
  for m in $MASKS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Xarch_device -fno-slp-vectorize -fPIC -shared \
      -fvisibility=hidden -DUMEREG_F1_ABLATE=$m -I include umeregrobust_amd/csrc/*.hip -o tools/libumereg_abl$m.so &
  done
  wait
else
  for m in $MASKS; do echo "mask $m"; ALTLIB=libumereg_abl$m.so timeout 90 python tools/exp_f1_lattice.py 3 2>&1 | grep "^plain:" | cut -c1-60; done
fi
