#!/bin/bash
# usage (GPU box): tools/r04_coarse_prio.sh -- (needs the two experiment libraries; the macro is not in the source any more) A/B of s_setprio around the coarse filter's MFMA groups / tile filter
# (tools/libumereg_prio1.so: MFMA phase high, filter low; libumereg_prio2.so: the opposite; built with -DUMEREG_COARSE_PRIO=1|2)
cd "$(dirname "$0")/.."
B="--no-cpu-baseline --no-e2e --hard-steps 0 --resident-steps 0 --steps 20"
for round in 1 2; do
  for lib in "" libumereg_prio1.so libumereg_prio2.so; do
    echo "---- round $round lib ${lib:-shipped}"
    ALTLIB=$lib timeout 120 python tools/exp_f16r_stats.py 2>&1 | grep "^coarse"
    ALTLIB=$lib timeout 300 python tools/bench_altlib.py $B 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('value', d['value'], 'coarse alone', r['avg_launch_ms'], 'in situ', r['in_situ_avg_launch_ms'])"
  done
done
