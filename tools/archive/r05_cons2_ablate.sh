#!/bin/bash
# usage (GPU box): tools/r05_cons2_ablate.sh  -- where the consensus pass's time goes: the consensus stage of one f1 call (tools/exp_f1_stage.py)
# on builds with parts of corr_consensus2_kernel compiled out (-DUMEREG_C2_ABLATE=<mask>, results wrong by construction; built by
# _build.build_native(extra_flags=["-DUMEREG_C2_ABLATE=<mask>"], out="tools/libc2_abl<mask>.so")) -> gpurun_out/c2_ablate.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/c2_ablate.txt; : > $O
cd $R
for m in "" 1 2 4 8 16 32; do
  echo "---- mask ${m:-0 (shipped)}" >> $O
  ALTLIB=${m:+libc2_abl$m.so} timeout 200 python tools/exp_f1_stage.py 6 2>&1 | grep "^plain\|^hard" | sed 's/lattice_build.*one_wavefront/.. one_wavefront/' | cut -c1-150 >> $O
done
cat $O
