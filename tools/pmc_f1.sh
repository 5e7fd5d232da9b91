#!/bin/bash
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_f1
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -s KILL 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $GRAFT_REPO_ROOT/tools/exp_f1g.py > $OUT/$tag.log 2>&1
done
