"""profiles/sq_summary.json (read by bench.py) from the SQ / TCC counter passes of tools/collect_profiles.sh.
usage: python tools/make_sq_summary.py gpurun_out/<tag> [profiles/<round>]   (copies the evidence too)
Per kernel (largest grid of its name only: bench.py's setup also launches single-cloud variants): averages per launch of
every counter, and the derived fractions the bench line quotes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else None
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CFG = open(os.path.join(src, "config.txt")).read().strip() if os.path.exists(os.path.join(src, "config.txt")) else "KT"
SUFFIX = "" if CFG == "KT" else "_" + CFG
LIB_HASH = open(os.path.join(src, "library_hash.txt")).read().strip() if os.path.exists(os.path.join(src, "library_hash.txt")) else None
N_SIMD = 1024                      # 256 CUs x 4 SIMDs (guides/MI355X_MICROARCH.md)
N_XCD = 8                          # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (8 x the kernel's duration in shader clocks)
KERNELS = ("ume_moments_kernel", "ume_coarse_h_kernel", "match_refine_kernel", "rtume_kernel", "orthobasis_pair_kernel", "grid_scatter_kernel",
           "kp_order_kernel", "match_prob_kernel", "pack_points_kernel", "grid_hist_kernel", "grid_scan_kernel", "hypothesis_gates_kernel")
rows = []
for f in sorted(glob.glob(os.path.join(src, "sq", "pass*.csv"))):
    rows += list(csv.DictReader(open(f)))
biggest = collections.defaultdict(int)
for r in rows:
    biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if int(r["Grid_Size"]) != biggest[r["Kernel_Name"]]:
        continue
    for short in KERNELS:
        if short in r["Kernel_Name"]:
            acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"library_source_hash": LIB_HASH, "config": CFG, "_comment": "per-launch averages of rocprofv3 --pmc passes (SQ, GRBM, TCC, TCP) of `bench.py --steps 2 --warmup 1 "
                   "--pairs-per-step 8 --depth 1 --no-e2e`, the workload named in `config`, one pass per counter set (tools/collect_profiles.sh). "
                   "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts; "
                   "SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE = kernel duration in "
                   "shader clocks, summed over the 8 XCDs (checked: / 8 / launch duration = 2.09 GHz).  mfma_busy_frac = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024); cross-check for the coarse matcher: 3.125e6 MFMAs "
                   "(= 1.024e11 flop / 32768) x 32 cycles = 1.0e8 busy cycles, as counted."}
for k, c in acc.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    d = {"counters": {n: round(v, 1) for n, v in sorted(m.items())}, "launches_sampled": max(len(v) for v in c.values())}
    g = m.get("GRBM_GUI_ACTIVE")
    if g:
        if m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            d["mfma_busy_frac"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / N_XCD * N_SIMD), 4)
        d["duration_shader_clocks"] = round(g / N_XCD, 0)
        if m.get("SQ_ACTIVE_INST_VALU"):
            # SQ_ACTIVE_INST_VALU counts quad-cycles (one per non-MFMA VALU instruction issued), summed over the SIMDs
            d["valu_busy_frac"] = round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (g / N_XCD * N_SIMD), 4)
    if m.get("SQ_INSTS_MFMA"):
        d["valu_per_mfma"] = round((m.get("SQ_INSTS_VALU", 0.0) - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"], 2)
    if m.get("SQ_WAVE_CYCLES"):
        w = m["SQ_WAVE_CYCLES"]
        for key, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "issue_frac"),
                          ("SQ_INST_CYCLES_VMEM", "vmem_busy_frac")):
            if key in m:
                d[name] = round(m[key] / w, 4)
    if m.get("TCC_HIT_sum") is not None and m.get("TCC_MISS_sum") is not None and m["TCC_HIT_sum"] + m["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
    if m.get("TCP_TOTAL_CACHE_ACCESSES_sum") and m.get("TCP_TCC_READ_REQ_sum") is not None:
        d["l1_hit_rate_est"] = round(1.0 - m["TCP_TCC_READ_REQ_sum"] / m["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
    # HBM-side request counters (TCC_EA0_*: requests the L2 sends to the fabric; 32 B and 64 B requests counted apart)
    if m.get("TCC_EA0_RDREQ_sum") is not None:
        r32 = m.get("TCC_EA0_RDREQ_32B_sum", 0.0)
        d["ea_read_bytes"] = round(32.0 * r32 + 64.0 * (m["TCC_EA0_RDREQ_sum"] - r32), 0)
    if m.get("TCC_EA0_WRREQ_sum") is not None:
        w64 = m.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        d["ea_write_bytes"] = round(64.0 * w64 + 32.0 * (m["TCC_EA0_WRREQ_sum"] - w64), 0)
    out[k] = d
name = "sq_summary" + SUFFIX + ".json"
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out.items() if isinstance(v, dict)}, indent=1))
if dst:
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "profiles", name), os.path.join(dst, "sq_summary.json"))
    for f in sorted(glob.glob(os.path.join(src, "sq", "pass*.csv"))):
        shutil.copy(f, os.path.join(dst, "sq_" + os.path.basename(f)))
