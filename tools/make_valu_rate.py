"""profiles/valu_rate.json (read by bench.py: the VALU-issue peak of the consensus pass is priced per instruction class) from the table
tools/probe/valu_rate.hip prints.  usage: python tools/make_valu_rate.py gpurun_out/valu_rate.txt [profiles/<round>]  (copies the table too)"""
import json
import os
import re
import shutil
import sys

src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else None
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
out = {"_comment": "chip rate of wave64 instructions per class, Ginst/s, from the wall time of launches that keep every SIMD of the chip on one "
                   "loop of that instruction (tools/probe/valu_rate.hip): [1, 2, 4, 8] wavefronts per SIMD, dependent chain and 8 independent "
                   "chains per wave; `best` = the largest of the eight cells = the issue peak of the class", "ops": {}}
for line in open(src):
    if line.startswith("#"):
        if "CUs" in line:
            out["device"] = line[1:].strip()
        continue
    m = re.match(r"^(.*?)\s+(dependent|8 chains)\s+\|(.*)$", line)
    if not m:
        continue
    rates = [float(x) for x in re.findall(r"([\d.]+) Ginst/s", m.group(3))]
    op = out["ops"].setdefault(m.group(1).strip(), {})
    op[m.group(2)] = rates
for op in out["ops"].values():
    op["best"] = max(max(v) for k, v in op.items() if k != "best")
json.dump(out, open(os.path.join(ROOT, "profiles", "valu_rate.json"), "w"), indent=1)
print({k: v["best"] for k, v in out["ops"].items()})
if dst:
    os.makedirs(dst, exist_ok=True)
    shutil.copy(src, os.path.join(dst, "valu_rate.txt"))
    pm = os.path.join(os.path.dirname(src), "valu_rate_pmc.txt")
    if os.path.exists(pm):
        shutil.copy(pm, os.path.join(dst, "valu_rate_pmc.txt"))
