#!/bin/bash
# usage (GPU box, repo root): tools/valu_rate.sh  -- the VALU issue-rate table (tools/probe/valu_rate.hip) into gpurun_out/valu_rate.txt,
# then the same kernels under rocprofv3 counters (shader clocks, VALU instruction and busy counts per launch) into gpurun_out/valu_rate_pmc.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
[ -x $R/tools/probe/valu_rate ] || hipcc --offload-arch=gfx950 -O2 -o $R/tools/probe/valu_rate $R/tools/probe/valu_rate.hip || exit 1
timeout 600 $R/tools/probe/valu_rate 4000 > $R/gpurun_out/valu_rate.txt 2>&1
cat $R/gpurun_out/valu_rate.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/vr_pmc
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/vr_pmc -o pmc -- \
  $R/tools/probe/valu_rate 4000 > /tmp/vr_pmc.log 2>&1
f=$(ls /tmp/vr_pmc/*/pmc_counter_collection.csv /tmp/vr_pmc/pmc_counter_collection.csv 2>/dev/null | head -1)
python3 - "$f" > $R/gpurun_out/valu_rate_pmc.txt <<'PY'
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
# one record per dispatch: the LAST (timed) launches of each (kernel, grid, workgroup) are what the table's cells are; average them all
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    acc[(r["Kernel_Name"], int(r["Grid_Size"]), int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# per launch (largest of the launches of a shape = the timed ones; the warm-up runs 1/8 of the iterations):")
print("# kernel | waves/SIMD | shader clocks (GRBM_GUI_ACTIVE / 8 XCDs) | SQ_INSTS_VALU per wave | clocks per VALU instruction per SIMD | 4 x SQ_ACTIVE_INST_VALU / (clocks x 1024)")
for (k, g, w), c in sorted(acc.items()):
    clk = max(c["GRBM_GUI_ACTIVE"]) / 8.0
    waves = g / 64
    wps = waves / 1024.0
    insts = max(c["SQ_INSTS_VALU"])
    act = max(c["SQ_ACTIVE_INST_VALU"])
    per_wave = insts / waves
    print(f"{k[:60]:60s} | {wps:4.1f} | {clk:12.0f} | {per_wave:10.0f} | {clk / (per_wave * wps):6.2f} | {4.0 * act / (clk * 1024):6.3f}")
PY
head -50 $R/gpurun_out/valu_rate_pmc.txt
