for c in K1 NS SY; do
  timeout 600 python bench.py --config $c --no-cpu-baseline --e2e-pairs 8 --e2e-hard-pairs 4 --detail gpurun_out/shape_$c.json > gpurun_out/shape_$c.log 2>&1; echo "rc $? $c"
done
timeout 600 python bench.py --kind rot --no-cpu-baseline --e2e-pairs 8 --e2e-hard-pairs 4 --detail gpurun_out/shape_rot.json > gpurun_out/shape_rot.log 2>&1; echo "rc $? rot"
python - <<'PY'
import json
for c in ("K1","NS","SY","rot"):
    import os
    if not os.path.exists(f"gpurun_out/shape_{c}.json"): print(c,"NO RESULT"); continue
    d=json.load(open(f"gpurun_out/shape_{c}.json"))
    e=d.get("end_to_end",{}); h=d.get("end_to_end_hard",{})
    print(c, d["value"], d["config"]["workload"][:60], "e2e", e.get("pairs_per_s"), e.get("rr_1.5deg_0.6m"), e.get("rr_1deg_0.1m"), "hard", h.get("pairs_per_s"), h.get("rr_1.5deg_0.6m"))
PY
