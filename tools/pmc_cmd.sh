#!/bin/bash
# usage: tools/pmc_cmd.sh <outdir-under-gpurun_out> <python script + args...>
# SQ counter passes for an arbitrary command (own runs; --kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
rm -rf $OUT; mkdir -p $OUT
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -s KILL 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/$tag -o pmc -- python $GRAFT_REPO_ROOT/"$@" > $OUT/$tag.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    import os
    if not any(x in k for x in os.environ.get("KFILTER", "coarse,refine,dist_h,moments").split(",")):
        continue
    print(k)
    for n, v in sorted(c.items()):
        print("   %-34s %14.0f  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
