#!/bin/bash
# usage (GPU box): tools/f1_pmc.sh [reps]  -- SQ counter passes of the production f1 path alone (tools/exp_f1_prod.py) on the plain and
# the half-overlapping KT pair, one rocprofv3 --pmc run per counter set (kernel-trace only, as gpurun requires), raw CSVs under
# gpurun_out/f1pmc/{plain,hard}/pass<i>.csv.  tools/make_f1_sq_summary.py turns them into profiles/f1_sq_summary.json.
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
REPS=${1:-3}
for which in plain hard; do
  OUT=$ROOT/gpurun_out/f1pmc/$which; rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
             "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
             "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES_EQ_64 SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32" \
             "SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_BRANCH SQ_IFETCH SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64"; do
    i=$((i+1))
    timeout -s KILL 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $ROOT/tools/exp_f1_prod.py $REPS $which > $OUT/p$i.log 2>&1
    f=$(ls $OUT/p$i/*/pmc_counter_collection.csv $OUT/p$i/pmc_counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && grep -E "Counter_Name|corr_|lattice_|leftover_|hyp_" "$f" > $OUT/pass$i.csv
    rm -rf $OUT/p$i
  done
  ls -la $OUT
done
