"""profiles/pmc_traffic.json from the two PMC passes of tools/collect_profiles.sh.
usage: python tools/make_pmc_traffic.py gpurun_out/<tag> [profiles/<round>]   (copies the evidence files too)"""
import csv, json, os, shutil, sys, collections
src = sys.argv[1]
dst = sys.argv[2] if len(sys.argv) > 2 else None
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CFG = open(os.path.join(src, "config.txt")).read().strip() if os.path.exists(os.path.join(src, "config.txt")) else "KT"
SUFFIX = "" if CFG == "KT" else "_" + CFG
LIB_HASH = open(os.path.join(src, "library_hash.txt")).read().strip() if os.path.exists(os.path.join(src, "library_hash.txt")) else None
raw = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    rows = [r for r in csv.DictReader(open(os.path.join(src, f"pmc_{c}.csv"))) if r["Counter_Name"] == c]
    # bench.py's setup also runs single-cloud moment launches (neighbour counts for the algorithmic-bytes
    # figure); the timed path launches the kernel on both clouds of a pair at once: keep the largest grid only
    biggest = collections.defaultdict(int)
    for r in rows:
        biggest[r["Kernel_Name"]] = max(biggest[r["Kernel_Name"]], int(r["Grid_Size"]))
    for r in rows:
        if int(r["Grid_Size"]) == biggest[r["Kernel_Name"]]:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for short in ("ume_moments_kernel", "ume_coarse_h_kernel", "match_refine_kernel", "ume_dist_h_kernel"):
            if short in k:
                raw[short][c] = {"mean": round(sum(v) / len(v), 1), "launches": len(v)}
out = {"library_source_hash": LIB_HASH, "config": CFG, "_comment": "HBM-side bytes per launch from rocprofv3 PMC (separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of "
                   "`bench.py --steps 2 --warmup 1 --pairs-per-step 8 --depth 1 --no-cpu-baseline --no-e2e` (tools/collect_profiles.sh: the named-path leg of the default command, shortened and with one pair in flight so that a launch is not time-shared), the workload named in `config`; the moment kernel launch covers both clouds of a "
                   "pair). Counters are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B for wide (16 B/lane) "
                   "reads, so reads are DOUBLED per guides/MI355X_MICROARCH.md section HBM; WRITE_SIZE is used as reported. "
                   "Infinity-Cache hits are included in these counters, so this is an upper bound on DRAM traffic."}
for k, v in raw.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out[k] = int(round((2 * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"]) * 1024))
out["raw_kib"] = raw
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic" + SUFFIX + ".json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("_comment", "raw_kib")}, indent=1))
if dst:
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(ROOT, "profiles", "pmc_traffic" + SUFFIX + ".json"), os.path.join(dst, "pmc_traffic.json"))
    for f in ("bench.json", "bench_detail.json", "library_hash.txt", "bench_default.json", "kernel_stats.csv", "pmc_FETCH_SIZE.csv", "pmc_WRITE_SIZE.csv", "bench_with_selection.json",
              "bench_end_to_end.json", "kernel_stats_end_to_end.csv", "f1_kernel_stats.txt", "e2e_kernel_stats.txt"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f))
