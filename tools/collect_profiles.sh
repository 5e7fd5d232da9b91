#!/bin/bash
# usage (on the GPU box, from the repo root): tools/collect_profiles.sh <tag> [KT|NS|SY|K1] [quick]
# Collects the evidence the bench line refers to, into gpurun_out/<tag>/ (tools/make_sq_summary.py / make_pmc_traffic.py turn it into
# profiles/sq_summary[_<config>].json / pmc_traffic[_<config>].json and copy what is to be judged to profiles/<round>/[<config>/]):
#   library_hash.txt       umereg_build_source_hash() of the library every pass below ran on (bench.py compares it with the loaded one)
#   bench.json / bench_detail.json   the bench run of that config (KT: the default command, CPU baseline included)
#   kernel_stats.csv       rocprofv3 --kernel-trace --stats of the same command (without the CPU leg)
#   pmc_{FETCH,WRITE}_SIZE.csv   separate --pmc passes of the named-path leg (kernel-trace only, as gpurun requires)
#   sq/                    SQ / TCC / TCP counter passes of the named-path leg
#   (KT, not quick)  f1_kernel_stats.txt, e2e_kernel_stats.txt, f1 SQ passes in gpurun_out/f1pmc
TAG=${1:-profiles_run}
CFG=${2:-KT}
QUICK=${3:-}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -c "import sys; sys.path.insert(0, '$ROOT'); from umeregrobust_amd import _lib; print(_lib.load().umereg_build_source_hash().decode())" > $OUT/library_hash.txt
echo "$CFG" > $OUT/config.txt
if [ "$CFG" = "KT" ]; then BENCH_ARGS=""; else BENCH_ARGS="--config $CFG --no-cpu-baseline --e2e-pairs 8 --e2e-hard-pairs 4"; fi
if [ -z "$ONLY_PMC" ]; then    # (ONLY_PMC=1: the counter passes alone, for a look at a kernel between two builds)
timeout -s KILL 1500 python $ROOT/bench.py $BENCH_ARGS --detail $OUT/bench_detail.json > $OUT/bench.log 2>&1; grep "^{" $OUT/bench.log | tail -1 > $OUT/bench.json
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- python $ROOT/bench.py $BENCH_ARGS --no-cpu-baseline --detail $OUT/stats_detail.json > $OUT/stats.log 2>&1
f=$(ls $OUT/stats/*/run_kernel_stats.csv $OUT/stats/run_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv
rm -rf $OUT/stats $OUT/stats_detail.json
fi
SMALL="--config $CFG --steps 2 --warmup 1 --pairs-per-step 8 --depth 1 --no-cpu-baseline --no-e2e --pool 4 --resident-steps 0 --hard-steps 0 --ragged-steps 0 --plan-check-pairs 0 --detail $OUT/small_detail.json"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python $ROOT/bench.py $SMALL > $OUT/pmc_$c.log 2>&1
  f=$(ls $OUT/pmc_$c/*/pmc_counter_collection.csv $OUT/pmc_$c/pmc_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/pmc_$c.csv
  rm -rf $OUT/pmc_$c
done
mkdir -p $OUT/sq
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_TRANS SQ_INSTS_SMEM SQ_INSTS_FLAT" \
           "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq/p$i -o pmc -- python $ROOT/bench.py $SMALL > $OUT/sq/p$i.log 2>&1
  f=$(ls $OUT/sq/p$i/*/pmc_counter_collection.csv $OUT/sq/p$i/pmc_counter_collection.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $OUT/sq/pass$i.csv
  rm -rf $OUT/sq/p$i
done
rm -f $OUT/small_detail.json
cd $ROOT
if [ "$CFG" = "KT" ] && [ -z "$QUICK" ]; then
  timeout -s KILL 300 $ROOT/tools/f1_stats.sh 3 > $OUT/f1_kernel_stats.txt 2>&1
  echo "---- production path alone (tools/f1_prod_stats.sh): plain KT pair, then the half-overlapping one" >> $OUT/f1_kernel_stats.txt
  timeout -s KILL 300 $ROOT/tools/f1_prod_stats.sh 10 plain >> $OUT/f1_kernel_stats.txt 2>&1
  timeout -s KILL 300 $ROOT/tools/f1_prod_stats.sh 5 hard >> $OUT/f1_kernel_stats.txt 2>&1
  echo "---- first vs second form of the consensus pass, leftover routing (tools/exp_f1_v2.py)" >> $OUT/f1_kernel_stats.txt
  timeout -s KILL 300 python $ROOT/tools/exp_f1_v2.py 5 plain,hard,rot v1,def,defR,defC,defL 2>&1 | grep "^plain\|^hard\|^rot" >> $OUT/f1_kernel_stats.txt
  # SQ counter passes of the f1 kernels (plain and half-overlapping pair) -> gpurun_out/f1pmc (tools/make_f1_sq_summary.py)
  timeout -s KILL 900 $ROOT/tools/f1_pmc.sh 3 > $OUT/f1_pmc.log 2>&1
  cp $OUT/library_hash.txt $ROOT/gpurun_out/f1pmc/library_hash.txt
  # launches and kernel time per end-to-end pair
  timeout -s KILL 600 $ROOT/tools/e2e_stats.sh > $OUT/e2e_kernel_stats.txt 2>&1
fi
ls -la $OUT $OUT/sq
