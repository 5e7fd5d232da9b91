#!/bin/bash
# usage (GPU box): tools/f1_pmc_cfg.sh <config: NSF, NS, LKF, KT ...> <plain|hard> [variant = defB]  -- SQ counter passes of one corr_scores job of
# tools/exp_f1_v2.py (one rocprofv3 --pmc run per counter set, kernel-trace only), summed per kernel -> gpurun_out/f1pmc_<config>_<kind>.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-NSF}; KIND=${2:-plain}; VAR=${3:-defB}
OUT=$ROOT/gpurun_out/f1pmc_cfg; rm -rf $OUT; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- python $ROOT/tools/exp_f1_v2.py 1 $KIND $VAR $CFG > $OUT/p$i.log 2>&1
  f=$(ls $OUT/p$i/*/pmc_counter_collection.csv $OUT/p$i/pmc_counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -E "Counter_Name|corr_|lattice_|leftover_|hyp_|cell_|flat_" "$f" > $OUT/pass$i.csv
  rm -rf $OUT/p$i
done
python - $OUT > $ROOT/gpurun_out/f1pmc_${CFG}_${KIND}.txt <<'PY'
import csv, sys, collections, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in sorted(glob.glob(sys.argv[1] + "/pass*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    L = max(n[k].values())
    g = c.get("GRBM_GUI_ACTIVE", 0) / max(n[k].get("GRBM_GUI_ACTIVE", 1), 1) / 8      # shader clocks per launch (summed over 8 XCDs)
    if g < 1e4: continue
    per = lambda x: c.get(x, 0) / max(n[k].get(x, 1), 1)
    print(f"{k:42s} launches {L:3d} clocks/launch {g:10.0f}  valu_busy {4 * per('SQ_ACTIVE_INST_VALU') / (g * 1024):.3f}  lds_busy {4 * per('SQ_ACTIVE_INST_LDS') / (g * 1024):.3f} "
          f"sca_busy {4 * per('SQ_ACTIVE_INST_SCA') / (g * 1024):.3f} vmem_busy {4 * per('SQ_ACTIVE_INST_VMEM') / (g * 1024):.3f} issue {per('SQ_ACTIVE_INST_ANY') / max(per('SQ_WAVE_CYCLES'), 1):.3f} wait_any {per('SQ_WAIT_ANY') / max(per('SQ_WAVE_CYCLES'), 1):.3f} "
          f"waves/simd {per('SQ_WAVE_CYCLES') * 4 / (g * 1024):.2f}  valu {per('SQ_INSTS_VALU'):.3e} salu {per('SQ_INSTS_SALU'):.3e} lds {per('SQ_INSTS_LDS'):.3e} vmem {per('SQ_INSTS_VMEM_RD'):.3e}")
PY
cat $ROOT/gpurun_out/f1pmc_${CFG}_${KIND}.txt
