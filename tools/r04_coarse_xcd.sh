#!/bin/bash
# usage (GPU box): tools/r04_coarse_xcd.sh -- (needs the experiment library; the macro is not in the source any more) A/B of the coarse filter's XCD-aware block placement (tools/libumereg_xcd1.so built with
# -DUMEREG_COARSE_XCD=1 against the in-tree library): stage time alone, bench value, fabric bytes per launch (FETCH_SIZE / WRITE_SIZE passes)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R
B="--no-cpu-baseline --no-e2e --hard-steps 0 --resident-steps 0 --steps 20"
for round in 1 2; do
  for lib in "" libumereg_xcd1.so; do
    echo "---- round $round lib ${lib:-in-tree}"
    ALTLIB=$lib timeout 120 python tools/exp_f16r_stats.py 2>&1 | grep "^coarse\|^moments"
    ALTLIB=$lib timeout 300 python tools/bench_altlib.py $B 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('value', d['value'], 'coarse alone', r['avg_launch_ms'], 'in situ', r['in_situ_avg_launch_ms'], 'gates', d['hypothesis_quality']['counts'])"
  done
done
cd /tmp && export TMPDIR=/tmp
for lib in "" libumereg_xcd1.so; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcx; ALTLIB=$lib timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcx -o pmc -- python $R/tools/exp_f16r_stats.py > /dev/null 2>&1
    f=$(ls /tmp/pmcx/*/pmc_counter_collection.csv /tmp/pmcx/pmc_counter_collection.csv 2>/dev/null | head -1)
    python - "$f" "$c" "${lib:-in-tree}" <<'PY'
import csv,sys
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if "ume_coarse_h_kernel" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[2]]
print(sys.argv[3], sys.argv[2], "launches", len(v), "mean per launch", sum(v)/max(len(v),1))
PY
  done
done
