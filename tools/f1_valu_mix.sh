#!/bin/bash
# usage (GPU box): tools/f1_valu_mix.sh [reps]  -- the DYNAMIC instruction-class mix of the f1 kernels (plain and half-overlapping KT pair):
# rocprofv3 --list-avail -> every SQ_INSTS_VALU_* class counter the part offers (+ instruction-fetch / LDS / scalar ones), collected in
# passes of <= 8 counters over tools/exp_f1_prod.py; per kernel and counter: average per launch -> gpurun_out/f1_valu_mix.txt.
# With tools/probe/valu_rate.hip's cycles per class this prices the consensus pass's VALU issue time (bench.py).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
REPS=${1:-3}
OUT=$ROOT/gpurun_out/f1mix; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -o "SQ_[A-Z0-9_]*\|TCC_EA0_[A-Z0-9_]*\|GRBM_[A-Z_]*" $OUT/avail.txt | sort -u > $ROOT/gpurun_out/counters_avail.txt
NAMES=$(grep -E "^SQ_INSTS_VALU_|^SQ_INSTS_VALU$|^SQ_IFETCH|^SQ_WAIT_IFETCH|^SQ_INST_LEVEL|^SQ_INSTS_BRANCH|^SQ_INSTS_SENDMSG|^SQ_INSTS_EXP|^SQ_VALU_MFMA_BUSY|^SQ_INSTS_SALU$|^SQ_INSTS_LDS$|^SQ_INST_CYCLES_SALU|^SQ_THREAD_CYCLES_VALU|^SQ_ACTIVE_INST_VALU$|^SQ_BUSY_CU_CYCLES|^SQ_ITEMS" $ROOT/gpurun_out/counters_avail.txt | grep -v MFMA_MOPS | tr '\n' ' ')
echo "counters: $NAMES" > $ROOT/gpurun_out/f1_valu_mix.txt
for which in plain hard; do
  set --; n=0; i=0
  for c in $NAMES GRBM_GUI_ACTIVE; do
    set -- "$@" $c; n=$((n+1))
    if [ $n -eq 6 ]; then
      i=$((i+1)); timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$which/p$i -o pmc -- python $ROOT/tools/exp_f1_prod.py $REPS $which > $OUT/$which.p$i.log 2>&1
      set --; n=0
    fi
  done
  if [ $n -gt 0 ]; then i=$((i+1)); timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$which/p$i -o pmc -- python $ROOT/tools/exp_f1_prod.py $REPS $which > $OUT/$which.p$i.log 2>&1; fi
  python3 - $OUT/$which $which >> $ROOT/gpurun_out/f1_valu_mix.txt <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
big = collections.defaultdict(int)
rows = []
for f in glob.glob(sys.argv[1] + "/p*/**/pmc_counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in rows:
    big[r["Kernel_Name"]] = max(big[r["Kernel_Name"]], int(r["Grid_Size"]))
for r in rows:
    if int(r["Grid_Size"]) == big[r["Kernel_Name"]] and "umereg::" in r["Kernel_Name"]:
        acc[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("====", sys.argv[2])
for k, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("GRBM_GUI_ACTIVE", [0])) / max(len(kv[1].get("GRBM_GUI_ACTIVE", [0])), 1)):
    g = c.get("GRBM_GUI_ACTIVE")
    if not g or sum(g) / len(g) / 8 < 20000:
        continue
    print(k, " clocks", round(sum(g) / len(g) / 8))
    for n, v in sorted(c.items()):
        if n != "GRBM_GUI_ACTIVE":
            print("   %-34s %14.0f  (n=%d)" % (n, sum(v) / len(v), len(v)))
PY
done
rm -rf $OUT/plain $OUT/hard
cat $ROOT/gpurun_out/f1_valu_mix.txt | head -120
