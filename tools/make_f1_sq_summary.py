"""profiles/f1_sq_summary.json (read by bench.py's `f1_selection`) from the SQ counter passes of tools/f1_pmc.sh.
usage: python tools/make_f1_sq_summary.py gpurun_out/f1pmc [profiles/<round>]   (copies the raw per-kernel CSVs too)
Per pair kind (plain / hard) and f1 kernel: per-launch averages of every counter (the launches of the probe's timed calls)
and the derived fractions DESIGN 3.6 quotes."""
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
N_SIMD, N_XCD = 1024, 8
KERNELS = ("corr_consensus2_kernel", "corr_consensus_kernel", "corr_score_flat_kernel", "corr_score_kernel", "corr_score_record2_kernel",
           "lattice_list_kernel", "lattice_mark_kernel", "lattice_compact_kernel")
LIB_HASH = open(os.path.join(src, "library_hash.txt")).read().strip() if os.path.exists(os.path.join(src, "library_hash.txt")) else None
out = {"library_source_hash": LIB_HASH, "_comment": "per-launch averages of rocprofv3 --pmc passes over tools/exp_f1_prod.py (corr_scores alone, default flags, KT pair: "
                   "2 500 hypotheses x 10 000 points), one pass per counter set (tools/f1_pmc.sh).  SQ_WAVE_CYCLES / SQ_WAIT_* / "
                   "SQ_ACTIVE_INST_* count quad-cycles summed over wavefronts; GRBM_GUI_ACTIVE = duration in shader clocks summed over "
                   "the 8 XCDs.  valu_busy_frac = 4 x SQ_ACTIVE_INST_VALU / (duration x 1024 SIMDs); issue_frac = SQ_ACTIVE_INST_ANY / "
                   "SQ_WAVE_CYCLES; wait_any_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; lds_busy_frac = 4 x SQ_ACTIVE_INST_LDS / (duration x 1024)."}
for which in ("plain", "hard"):
    rows = []
    for f in sorted(glob.glob(os.path.join(src, which, "pass*.csv"))):
        rows += list(csv.DictReader(open(f)))
    biggest = collections.defaultdict(int)
    for r in rows:
        biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        name = r["Kernel_Name"]
        if int(r["Grid_Size"]) != biggest[name]:
            continue
        for short in KERNELS:
            if "::" + short + "(" in name or "::" + short + "<" in name:
                key = short + ("_lattice" if short == "corr_score_kernel" and "true>" in name.split("(")[0] else "")
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
    res = {}
    for k, c in acc.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        d = {"launches_sampled": max(len(v) for v in c.values())}
        g = m.get("GRBM_GUI_ACTIVE")
        if g:
            clocks = g / N_XCD
            d["duration_shader_clocks"] = round(clocks, 0)
            if m.get("SQ_ACTIVE_INST_VALU"):
                d["valu_busy_frac"] = round(4.0 * m["SQ_ACTIVE_INST_VALU"] / (clocks * N_SIMD), 4)
            if m.get("SQ_ACTIVE_INST_LDS"):
                d["lds_busy_frac"] = round(4.0 * m["SQ_ACTIVE_INST_LDS"] / (clocks * N_SIMD), 4)
            if m.get("SQ_ACTIVE_INST_SCA"):
                d["scalar_busy_frac"] = round(4.0 * m["SQ_ACTIVE_INST_SCA"] / (clocks * N_SIMD), 4)
            if m.get("SQ_INSTS_VALU"):
                d["valu_inst_per_simd_clock"] = round(m["SQ_INSTS_VALU"] / (clocks * N_SIMD), 4)
        if m.get("SQ_WAVE_CYCLES"):
            w = m["SQ_WAVE_CYCLES"]
            for key, name in (("SQ_WAIT_ANY", "wait_any_frac"), ("SQ_WAIT_INST_ANY", "wait_inst_frac"), ("SQ_ACTIVE_INST_ANY", "issue_frac"),
                              ("SQ_WAIT_INST_LDS", "wait_lds_frac")):
                if key in m:
                    d[name] = round(m[key] / w, 4)
            if g:
                d["avg_waves_per_simd"] = round(w * 4.0 / (g / N_XCD * N_SIMD), 2)
        for key in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VALU_TRANS", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT"):
            if key in m:
                d[key.lower()] = round(m[key], 0)
        d["counters"] = {n: round(v, 1) for n, v in sorted(m.items())}
        res[k] = d
    out[which] = res
json.dump(out, open(os.path.join(ROOT, "profiles", "f1_sq_summary.json"), "w"), indent=1)
print(json.dumps({w: {k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out[w].items()} for w in ("plain", "hard")}, indent=1))
if dst:
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "profiles", "f1_sq_summary.json"), os.path.join(dst, "f1_sq_summary.json"))
    for which in ("plain", "hard"):
        for f in sorted(glob.glob(os.path.join(src, which, "pass*.csv"))):
            shutil.copy(f, os.path.join(dst, f"f1_sq_{which}_" + os.path.basename(f)))
