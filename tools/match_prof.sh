#!/bin/bash
# per-kernel durations + SQ counters of the matcher kernels (tools/exp_f16r_stats.py as the workload)
# usage (on the GPU box): tools/match_prof.sh <tag>   -> gpurun_out/<tag>/{stats.txt,sq.txt}
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-mprof}
mkdir -p $OUT
cd $ROOT
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/st -o run -- python tools/exp_f16r_stats.py > $OUT/st.log 2>&1
f=$(ls $OUT/st/*/run_kernel_stats.csv $OUT/st/run_kernel_stats.csv 2>/dev/null | head -1)
python - "$f" > $OUT/stats.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if any(k in n for k in ("coarse", "pform", "refine", "orthobasis")):
        print(f'{n[:60]:60s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"]) / 1e3:9.1f} min_us {float(r["MinNs"]) / 1e3:9.1f}')
PY
cat $OUT/stats.txt; grep coarse $OUT/st.log
rm -rf $OUT/st
i=0
: > $OUT/sq.txt
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout -s KILL 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python tools/exp_f16r_stats.py > $OUT/p$i.log 2>&1
  f=$(ls $OUT/p$i/*/pmc_counter_collection.csv $OUT/p$i/pmc_counter_collection.csv 2>/dev/null | head -1)
  python - "$f" >> $OUT/sq.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0]
    if "coarse" in n:
        acc[n[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    for c, v in d.items():
        print(f"{n:40s} {c:32s} {sum(v) / len(v):16.0f}")
PY
  rm -rf $OUT/p$i
done
cat $OUT/sq.txt
