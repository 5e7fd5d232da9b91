"""Whole-pipeline replay parity (reference evaluate.py:195-309): the CPU oracle (`oracle.evaluate_pair_full`, a restatement of the
reference's loop iteration) against this library with the oracle's five host draws per pair REPLAYED (keypoints x 2, weighted match
draw, correlation sub-samples x 2), so the two paths see the same random numbers and can be compared pair by pair:

  (a) `--kt N`      N KITTI-size HARD pairs (N = 50 000 points, 10 000 keypoints, M = 2 500 hypotheses; partial overlap, 2 cm noise,
                    20 % corrupted features) -- the benchmark size, where f1 / f2 were otherwise only compared stage by stage;
  (b) `--small N`   N reduced-size harder pairs (N = 4 096, M = 256; bench.py's RR_CHECK_HARD), enough of them for a recall figure;
  (c) `--ns N`      N nuScenes-test-size HARD pairs (35 000 points, 5 000 keypoints = hypotheses, 30 000 correlation points, no match
                    filtering): the sizes at which f1 runs its cell pass and bounds the queries outside the lattice (the oracle's
                    brute force needs ~70 s per pair on 256 cores).

Writes per-pair |dRRE|, |dRTE|, gate outcomes of both paths and the summary to the JSON given by --out (tracked copy:
profiles/r03/rr_replay.json).  Runs on the GPU box (HIP path) and on its host cores (oracle, C/OpenMP + numpy).
usage: python tools/rr_replay.py [--kt 8] [--small 128] [--ns 0] [--out gpurun_out/rr_replay.json]"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import oracle as orc  # noqa: E402
from umeregrobust_amd import evaluate  # noqa: E402
from umeregrobust_amd.host_rng import RecordingRNG, ReplayRNG  # noqa: E402
from umeregrobust_amd.synth import synth_pair_hard  # noqa: E402
from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml  # noqa: E402

GATES = ((1.5, 0.6), (1.5, 0.3), (1.0, 0.1))
RR_CHECK_HARD = dict(sector_deg=180.0, sector_shift_deg=120.0, noise_sigma=0.03, feat_corrupt=0.5)


def wilson(k, n, z=1.96):
    """95 % Wilson interval of a binomial proportion, in percent."""
    if n == 0:
        return [0.0, 100.0]
    p = k / n
    d = 1 + z * z / n
    c = (p + z * z / (2 * n)) / d
    h = z * np.sqrt(p * (1 - p) / n + z * z / (4 * n * n)) / d
    return [round(100 * max(0.0, c - h), 2), round(100 * min(1.0, c + h), 2)]


def run(label, n_pairs, N, n_kp, M, hard_kw, seed0, args, dev):
    a = SimpleNamespace(**vars(args))
    a.ume_n_samples = M
    a.pc_corr_max_size = min(args.pc_corr_max_size, N)
    rows, t_cpu, t_gpu = [], 0.0, 0.0
    for i in range(n_pairs):
        p = synth_pair_hard(seed=seed0 + i, N=N, n_kp=n_kp, voxel=0.3, **hard_kw)
        rec = RecordingRNG(np.random.RandomState(31 + i))
        t0 = time.perf_counter()
        rc = orc.evaluate_pair_full(p.src_pts, p.tgt_pts, p.src_feat, p.tgt_feat, p.gt_tform, rec, ume_max_nn=a.ume_max_nn,
                                    ume_r_nn=a.ume_r_nn, ume_n_samples=M, tau=a.tau, filter_by_ume_dist_cond=a.filter_by_ume_dist_cond,
                                    corr_ds=a.corr_ds, pc_corr_max_size=a.pc_corr_max_size, sigma=a.corr_kernel_sigma)
        t_cpu += time.perf_counter() - t0
        t = lambda x: torch.from_numpy(x).to(dev)   # noqa: E731
        pair = dict(src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None], tgt_feat=t(p.tgt_feat)[None],
                    gt_tform=t(p.gt_tform))
        t0 = time.perf_counter()
        try:
            with torch.no_grad():
                rp = evaluate.evaluate_pairs([pair], a, rng=ReplayRNG(rec.log), refine=True)
            g = (float(rp["rre"][0]), float(rp["rte"][0]))
        except ValueError as e:                      # a recorded draw that does not fit (different voxel count): reported, not hidden
            g = (float("nan"), float("nan"))
            print(f"{label} pair {i}: replay failed: {e}", flush=True)
        t_gpu += time.perf_counter() - t0
        rows.append(dict(pair=i, seed=seed0 + i, cpu_rre_deg=round(rc["rre"], 5), cpu_rte_m=round(rc["rte"], 5), hip_rre_deg=round(g[0], 5),
                         hip_rte_m=round(g[1], 5), d_rre_deg=round(abs(g[0] - rc["rre"]), 6), d_rte_m=round(abs(g[1] - rc["rte"]), 6),
                         cpu_gates=[bool(rc["rre"] <= r and rc["rte"] <= tt) for r, tt in GATES],
                         hip_gates=[bool(g[0] <= r and g[1] <= tt) for r, tt in GATES]))
        print(f"{label} pair {i}: cpu ({rc['rre']:.4f} deg, {rc['rte']:.4f} m)  hip ({g[0]:.4f}, {g[1]:.4f})", flush=True)
    n = len(rows)
    ok_c = np.array([r["cpu_gates"] for r in rows]); ok_h = np.array([r["hip_gates"] for r in rows])
    d_rre = np.array([r["d_rre_deg"] for r in rows]); d_rte = np.array([r["d_rte_m"] for r in rows])
    # a pair both paths fail can end anywhere (ICP from a wrong basin): the per-pair bound is meaningful where a gate is passed
    both_pass = ok_c[:, 0] & ok_h[:, 0]
    return dict(size=f"N={N} pts/cloud, {n_kp} keypoints, M={M} hypotheses; hard pairs: {hard_kw or 'synth_pair_hard defaults'}", pairs=n,
                gates=["1.5deg,0.6m", "1.5deg,0.3m", "1deg,0.1m"],
                cpu_rr_percent=[round(100.0 * float(v), 3) for v in ok_c.mean(0)], hip_rr_percent=[round(100.0 * float(v), 3) for v in ok_h.mean(0)],
                cpu_rr_ci95=[wilson(int(k), n) for k in ok_c.sum(0)], hip_rr_ci95=[wilson(int(k), n) for k in ok_h.sum(0)],
                pairs_with_a_different_gate_outcome=[r["pair"] for r, a_, b_ in zip(rows, ok_c, ok_h) if (a_ != b_).any()],
                max_abs_diff={"rre_deg": round(float(np.nanmax(d_rre)), 6), "rte_m": round(float(np.nanmax(d_rte)), 6)},
                max_abs_diff_where_both_pass_the_first_gate={"rre_deg": round(float(np.nanmax(d_rre[both_pass])) if both_pass.any() else 0.0, 6),
                                                             "rte_m": round(float(np.nanmax(d_rte[both_pass])) if both_pass.any() else 0.0, 6),
                                                             "pairs": int(both_pass.sum())},
                cpu_s_per_pair=round(t_cpu / max(n, 1), 2), hip_s_per_pair_incl_upload=round(t_gpu / max(n, 1), 3), per_pair=rows)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kt", type=int, default=8)
    ap.add_argument("--small", type=int, default=128)
    ap.add_argument("--ns", type=int, default=0)
    ap.add_argument("--out", default="gpurun_out/rr_replay.json")
    c = ap.parse_args()
    dev = torch.device("cuda:0")
    args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("kitti_test"))
    orc.lib()
    out = {"what": "oracle.evaluate_pair_full (CPU restatement of reference evaluate.py:195-309) vs evaluate.evaluate_pairs (this library) "
                   "with the oracle's five host draws per pair replayed", "cores": os.cpu_count()}
    if c.kt:
        out["kitti_size"] = run("KT", c.kt, 50000, 10000, 2500, {}, 9000, args, dev)
    if c.small:
        out["reduced_size"] = run("small", c.small, 4096, 4096, 256, RR_CHECK_HARD, 20000, args, dev)
    if c.ns:
        args_ns = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path("nuscenes_test"))
        out["nuscenes_size"] = run("NS", c.ns, 35000, 5000, 5000, {}, 11000, args_ns, dev)
    os.makedirs(os.path.dirname(os.path.abspath(c.out)), exist_ok=True)
    json.dump(out, open(c.out, "w"), indent=1)
    for k in ("kitti_size", "reduced_size", "nuscenes_size"):
        if k in out:
            r = out[k]
            print(k, r["pairs"], "pairs | cpu RR", r["cpu_rr_percent"], "hip RR", r["hip_rr_percent"], "| different gate outcome:",
                  r["pairs_with_a_different_gate_outcome"], "| max diff", r["max_abs_diff"], "| where both pass:",
                  r["max_abs_diff_where_both_pass_the_first_gate"], "| cpu s/pair", r["cpu_s_per_pair"], flush=True)


if __name__ == "__main__":
    main()
