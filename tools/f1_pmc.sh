#!/bin/bash
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/f1pmc; mkdir -p $OUT; cd $ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_WAVES_EQ_64 SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python tools/exp_f1_prod.py 3 plain > $OUT/p$i.log 2>&1
  f=$(ls $OUT/p$i/*/pmc_counter_collection.csv $OUT/p$i/pmc_counter_collection.csv 2>/dev/null | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if "fallback" in n or "consensus" in n:
        acc[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    for c, v in d.items():
        print(f"{n:40s} {c:28s} {sum(v) / len(v):16.0f}")
PY
  rm -rf $OUT/p$i
done
