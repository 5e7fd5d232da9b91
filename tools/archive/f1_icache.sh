#!/bin/bash
# usage (GPU box): tools/f1_icache.sh [plain|hard]  -- instruction-cache and wave-level counters of the f1 kernels -> gpurun_out/f1_icache.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
W=${1:-plain}
OUT=$ROOT/gpurun_out/f1ic; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_STALL SQC_TC_REQ SQ_INST_CYCLES_SMEM"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $ROOT/tools/exp_f1_prod.py 3 $W > $OUT/p$i.log 2>&1
done
python3 - $OUT > $ROOT/gpurun_out/f1_icache.txt <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list)); big = collections.defaultdict(int); rows = []
for f in glob.glob(sys.argv[1] + "/p*/**/pmc_counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in rows:
    big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
for r in rows:
    if int(r["Grid_Size"]) == big[r["Kernel_Name"]] and ("consensus2" in r["Kernel_Name"] or "flat_kernel" in r["Kernel_Name"]):
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k)
    for n, v in sorted(c.items()):
        print("   %-34s %14.0f  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
cat $ROOT/gpurun_out/f1_icache.txt
