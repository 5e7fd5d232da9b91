#!/usr/bin/env python3
"""bench.py -- throughput of the UMERegRobust registration hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A "step" is one pass of the named hot path over one synthetic
KITTI-shaped registration pair whose inputs are already resident in HBM: keypoint gather ->
fused ball-query + UME moments (x2 clouds) -> orthonormal bases -> MFMA subspace-distance GEMM with
fused row arg-min -> match probabilities -> tau-weighted sub-sampling (host numpy RNG, like the
reference evaluate.py:238) -> closed-form SE(3) per match -> RRE/RTE of every hypothesis.
Pairs are independent, so N GPUs run N disjoint pair streams (weak scaling); the only collective
is the final all-reduce of the metric counters.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (guides/MI355X_MICROARCH.md); ~6.3 TB/s achievable
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak (same guide)
MFMA_F16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_f16 dense peak (same guide; 2:1-sparsity figures excluded)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="KT", choices=["K1", "KT", "NS", "SY"])
    ap.add_argument("--kind", default="test", choices=["test", "rot"])
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic pairs per rank (cycled)")
    ap.add_argument("--precision", default="f16r", choices=["f16r", "f16x2", "f32"],
                    help="distance GEMM: split-f16 MFMA (fp32-class operands, default) or exact-fp32 MFMA")
    ap.add_argument("--depth", type=int, default=2,
                    help="pairs in flight: 2 overlaps the host RNG draw of pair i with the GPU work of pair i+1; 1 = serial")
    ap.add_argument("--threaded-draw", action="store_true",
                    help="host RNG draw on a worker thread (the native draw releases the GIL, so it overlaps the main "
                         "thread's kernel enqueues); off by default: at 0.08 ms per draw the hand-off jitter costs more")
    ap.add_argument("--no-batch-clouds", dest="batch_clouds", action="store_false",
                    help="run source and target clouds as two launches instead of one batch of 2")
    ap.add_argument("--dist-backend", default=None, help="(testing) torch.distributed backend override, e.g. gloo")
    ap.add_argument("--force-device", type=int, default=None,
                    help="(testing) put every rank on this device index, to exercise the N>1 path on a 1-GPU box")
    ap.add_argument("--with-selection", action="store_true",
                    help="also run SURVEY 8(f1) hypothesis selection (FeatureCorrelator) per pair and report the "
                         "registration recall of the SELECTED transform; the headline metric stays the a1-a7 path")
    ap.add_argument("--with-refinement", action="store_true",
                    help="with --with-selection: also refine the selected transform by SURVEY 8(f2) point-to-point ICP "
                         "(0.2 m, <= 200 iterations, reference evaluate.py:93-96) and report its recall / mean errors")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=16, help="max pairs timed by the CPU baseline leg (stops after ~10 s)")
    return ap.parse_args()


def main():
    a = parse()
    import torch
    import torch.distributed as dist

    import umeregrobust_amd
    from umeregrobust_amd import evaluate, ops
    from umeregrobust_amd.dist import RegistrationMetrics, init_distributed
    from umeregrobust_amd.synth import CONFIGS, synth_pair
    from umeregrobust_amd.utils.general_utils import benchmark_config_path, update_namespace_from_yaml

    ops.DEFAULT_MATCH_PRECISION = a.precision
    if a.force_device is not None:
        os.environ["LOCAL_RANK"] = str(a.force_device)
    rank, local_rank, world = init_distributed(backend=a.dist_backend)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world} (launch N>1 with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs a HIP device (no CPU fallback)"
    umeregrobust_amd.require_native()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    cfg = CONFIGS[a.config]
    args = update_namespace_from_yaml(SimpleNamespace(), benchmark_config_path(
        "nuscenes_test" if a.config == "NS" else "kitti_test"))
    args.ume_n_samples = cfg["M"]
    args.filter_by_ume_dist_cond = cfg["filter_by_ume_dist_cond"]
    n_kp = cfg["n_kp"] if args.filter_by_ume_dist_cond else min(cfg["n_kp"], args.ume_n_samples)

    # ---- synthetic inputs, resident in HBM before the timed region ---------------------------------
    pool = []
    for i in range(max(1, a.pool)):
        p = synth_pair(seed=1000 * rank + i, N=cfg["N"], n_kp=n_kp, kind=a.kind, voxel=cfg["voxel"])
        t = lambda x: torch.from_numpy(x).to(dev)
        pool.append(SimpleNamespace(
            host=p, src_pts=t(p.src_pts)[None], tgt_pts=t(p.tgt_pts)[None], src_feat=t(p.src_feat)[None],
            tgt_feat=t(p.tgt_feat)[None], src_inds=t(p.src_inds), tgt_inds=t(p.tgt_inds),
            gt=t(p.gt_tform).contiguous()))
        e = pool[-1]
        if a.with_selection:   # random <= pc_corr_max_size subsets with their features (evaluate.py:277-285)
            rs = np.random.RandomState(77 + i)
            ns = min(args.pc_corr_max_size, cfg["N"])
            si = t(rs.choice(cfg["N"], ns, replace=False)); ti = t(rs.choice(cfg["N"], ns, replace=False))
            e.corr = (e.src_pts[:, si].contiguous(), e.tgt_pts[:, ti].contiguous(), e.src_feat[:, si].contiguous(),
                      e.tgt_feat[:, ti].contiguous())
        e.pair = evaluate.PairBatch.from_clouds(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, e.src_inds, e.tgt_inds) \
            if a.batch_clouds else None
    # neighbour counts (for the algorithmic-bytes roofline), outside the timed region
    for e in pool:
        per_cloud = []
        for pts, inds, feat in ((e.src_pts, e.src_inds, e.src_feat), (e.tgt_pts, e.tgt_inds, e.tgt_feat)):
            _, cnt = ops.ume_moments(pts, pts[:, inds], feat, args.ume_max_nn, args.ume_r_nn, return_count=True)
            per_cloud.append(float((140.0 * cnt.double() + 524.0).sum().item()))
        # one moment-kernel launch covers both clouds when they are batched, one cloud otherwise
        e.mom_bytes = [sum(per_cloud)] if e.pair is not None else per_cloud
    dist_flops = 2.0 * (4 * n_kp) * (4 * n_kp) * 32                                  # Q-form GEMM, d_used = 512-equiv

    rng = np.random.RandomState(1234 + rank)
    depth = max(1, a.depth)
    pipe = evaluate.RegistrationPipeline(args, dev, depth=depth, rng=rng, threaded_draw=a.threaded_draw)
    # hypotheses, ok(1.5deg,0.6m), ok(1.5deg,0.3m), ok(1deg,0.1m): integer atomics, one tensor per stream slot
    counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(depth)]
    timing = {"moments": [], "dist": ops.TimingList()}
    mom_bytes_log = []

    def submit(i, record):
        e = pool[i % len(pool)]
        h = pipe.submit(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, src_inds=e.src_inds, tgt_inds=e.tgt_inds,
                        timing=timing if record else None, pair=e.pair)
        h.entry = e
        if record:
            mom_bytes_log.extend(e.mom_bytes)
        return h

    sel_counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(depth)]
    ref_counts = [torch.zeros(4, dtype=torch.int64, device=dev) for _ in range(depth)]
    sel_timing = []
    icp_log = []

    def finish(h):
        out = pipe.finish(h)                                    # host RNG draw + SE(3) hypotheses
        with torch.cuda.stream(pipe.stream_of(h)):              # a7 + recall gates, on the device
            ops.hypothesis_gates(out.rtume_tform[0], h.entry.gt, counts[h.slot])
            if a.with_selection:                                # f1: evaluate.py:260-296 on <= pc_corr_max_size points
                e = h.entry
                _, _, R_hat, t_hat = evaluate.pc_fcht(e.corr[0], e.corr[1], e.corr[2], e.corr[3], out.rtume_tform, e.gt[None],
                                                      args.corr_kernel_sigma, args, timing=sel_timing)
                T_sel = torch.eye(4, device=dev)[None].repeat(1, 1, 1)
                T_sel[:, :3, :3] = R_hat
                T_sel[:, :3, 3] = t_hat
                ops.hypothesis_gates(T_sel.contiguous(), e.gt, sel_counts[h.slot])
                if a.with_refinement:                           # f2: evaluate.py:63-109 (synchronous: host-side stop test)
                    t_icp = time.perf_counter()
                    reg = ops.icp_point_to_point(e.src_pts[0], e.tgt_pts[0], T_sel[0].double().cpu().numpy(), 0.2, 200)
                    icp_log.append((time.perf_counter() - t_icp, reg.iterations, reg.fitness))
                    T_ref = torch.from_numpy(reg.transformation).float().to(dev)[None].contiguous()
                    ops.hypothesis_gates(T_ref, e.gt, ref_counts[h.slot])

    def run(first, n, record):
        pending = []
        for i in range(n):
            # per-kernel event pairs (the roofline leg) on every 4th timed step only: they need the layered entry points;
            # the other steps go through the one-call a1..a5 entry
            pending.append(submit(first + i, record and i % 4 == 0))
            if len(pending) >= depth:
                finish(pending.pop(0))
        while pending:
            finish(pending.pop(0))

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, a.warmup, False)
    torch.cuda.synchronize()
    for c_ in counts + sel_counts + ref_counts:
        c_.zero_()
    icp_log.clear()
    sel_timing.clear()
    fence()
    t0 = time.perf_counter()
    run(a.warmup, a.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    counts = torch.stack(counts).sum(0).double()
    sel_counts = torch.stack(sel_counts).sum(0).double()
    ref_counts = torch.stack(ref_counts).sum(0).double()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)      # the path's one collective (32 B)
        dist.all_reduce(sel_counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(ref_counts, op=dist.ReduceOp.SUM)

    # ---- per-kernel durations measured live with events on the launch stream -----------------------
    mom_ms = [s.elapsed_time(e_) for s, e_ in timing["moments"]]
    dist_ms = [s.elapsed_time(e_) for s, e_ in timing["dist"]]
    mom_total_ms, dist_total_ms = float(np.sum(mom_ms)), float(np.sum(dist_ms))
    mom_gbs = float(np.sum(mom_bytes_log)) / (mom_total_ms * 1e-3) / 1e9
    dist_tfs = dist_flops * len(dist_ms) / (dist_total_ms * 1e-3) / 1e12
    roof_mom = {"kernel": "ume_moments_kernel", "bound": "hbm", "achieved": round(mom_gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(mom_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "avg_launch_ms": round(float(np.mean(mom_ms)), 4), "launches": len(mom_ms),
                "algorithmic_bytes_per_launch": round(float(np.mean(mom_bytes_log)), 0),
                "note": "algorithmic bytes = SURVEY 8(d) per-keypoint figure (140 n_i + 524) summed over both clouds of a pair; "
                        "they are neighbour gathers from 8 MB tables, served mostly by L2 / Infinity Cache, so the rate can exceed "
                        "the HBM peak while `traffic` (fabric bytes from PMC) stays far below the algorithmic bytes"}
    if a.precision == "f16r":
        # filter + refine: ONE f16 MFMA product per algorithmic product in the coarse kernel (the timed
        # region is that kernel alone); the fp64 refine of the ~15 candidates per row is
        # timed separately.  Algorithmic flops = 2*512 per (source, target) pair, as for the scans.
        ref_ms = [s.elapsed_time(e_) for s, e_ in timing["dist"].refine]
        roof_dist = {"kernel": "ume_coarse_h_kernel", "bound": "mfma", "achieved": round(dist_tfs, 2),
                     "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dist_tfs / MFMA_F16_PEAK_TFLOPS, 4),
                     "traffic": None, "avg_launch_ms": round(float(np.mean(dist_ms)), 4), "launches": len(dist_ms),
                     "algorithmic_flops_per_launch": dist_flops,
                     "refine_avg_launch_ms": round(float(np.mean(ref_ms)), 4) if ref_ms else None,
                     "d_used": "512-equivalent (Q-form), single f16 MFMA product (hi planes) + fp64 refine of the candidates"}
    elif a.precision == "f16x2":
        # 3 f16 MFMA products per algorithmic product (hi*hi, hi*lo, lo*hi): the flops the MFMA pipe
        # executes are 3x the algorithmic count; `achieved` stays ALGORITHMIC, `issued` is reported too
        roof_dist = {"kernel": "ume_dist_h_kernel", "bound": "mfma", "achieved": round(dist_tfs, 2),
                     "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dist_tfs / MFMA_F16_PEAK_TFLOPS, 4),
                     "traffic": None, "avg_launch_ms": round(float(np.mean(dist_ms)), 4), "launches": len(dist_ms),
                     "algorithmic_flops_per_launch": dist_flops, "issued_tflops": round(3 * dist_tfs, 2),
                     "issued_frac": round(3 * dist_tfs / MFMA_F16_PEAK_TFLOPS, 4),
                     "d_used": "512-equivalent (Q-form), split-f16 MFMA (3 products, fp32 accumulate)"}
    else:
        roof_dist = {"kernel": "ume_dist_kernel", "bound": "mfma", "achieved": round(dist_tfs, 2),
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(dist_tfs / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": None, "avg_launch_ms": round(float(np.mean(dist_ms)), 4), "launches": len(dist_ms),
                     "algorithmic_flops_per_launch": dist_flops, "d_used": "512-equivalent (Q-form), fp32 MFMA"}
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")      # filled from rocprofv3 --pmc passes, if present
    if os.path.exists(pmc):
        tr = json.load(open(pmc))
        roof_mom["traffic"] = tr.get("ume_moments_kernel")
        roof_dist["traffic"] = tr.get(roof_dist["kernel"])
    dominant = roof_mom if mom_total_ms >= dist_total_ms else roof_dist

    total_pairs = a.steps * world
    c = counts.cpu().numpy()
    result = {
        "metric": "registration_pairs_per_s", "value": round(total_pairs / elapsed, 3), "unit": "pairs/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * elapsed / a.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{a.config}: named hot path a1-a7 on synthetic KITTI-shaped pairs "
                               f"(N={cfg['N']} pts/cloud, {n_kp} keypoints/cloud, K={args.ume_max_nn}, r={args.ume_r_nn} m, "
                               f"d=32, M={args.ume_n_samples} hypotheses, tau={args.tau}, kind={a.kind})",
                   "pairs_per_step_per_gpu": 1, "sharding": f"pairs[rank::{world}] (no data-path collective)",
                   "sampler": "host numpy RNG (reference evaluate.py:238)", "distance_gemm": a.precision,
                   "pairs_in_flight": depth, "host_draw_thread": bool(a.threaded_draw), "clouds_per_moment_launch": 2 if a.batch_clouds else 1},
        "roofline": dominant,
        "rooflines": {"ume_moments_kernel": roof_mom, roof_dist["kernel"]: roof_dist},
        "hypothesis_quality": {"hypotheses": int(c[0]), "within_1.5deg_0.6m": round(c[1] / max(c[0], 1), 4),
                               "within_1.5deg_0.3m": round(c[2] / max(c[0], 1), 4),
                               "within_1deg_0.1m": round(c[3] / max(c[0], 1), 4),
                               "note": "fraction of RTUME hypotheses (not selected registrations) inside each gate"},
    }

    if a.with_selection:
        sc = sel_counts.cpu().numpy()
        sel_ms = [s_.elapsed_time(e_) for s_, e_ in sel_timing]
        result["metric"] = "registration_pairs_per_s_with_hypothesis_selection"
        result["config"]["workload"] += " + f1 hypothesis selection (FeatureCorrelator, K=20, sigma=%.2f, <=%d pts)" % (
            args.corr_kernel_sigma, args.pc_corr_max_size)
        result["selection"] = {"pairs": int(sc[0]), "rr_1.5deg_0.6m": round(sc[1] / max(sc[0], 1), 4),
                               "rr_1.5deg_0.3m": round(sc[2] / max(sc[0], 1), 4), "rr_1deg_0.1m": round(sc[3] / max(sc[0], 1), 4),
                               "corr_scores_avg_ms": round(float(np.mean(sel_ms)), 3) if sel_ms else None,
                               "note": "recall of the transform SELECTED by feature correlation among the M RTUME "
                                       "hypotheses, on synthetic pairs (no ICP refinement)"}
    if a.with_selection and a.with_refinement:
        rc = ref_counts.cpu().numpy()
        result["metric"] = "registration_pairs_per_s_with_selection_and_icp"
        result["config"]["workload"] += " + f2 point-to-point ICP (0.2 m, <= 200 iterations)"
        result["refinement"] = {"pairs": int(rc[0]), "rr_1.5deg_0.6m": round(rc[1] / max(rc[0], 1), 4),
                                "rr_1.5deg_0.3m": round(rc[2] / max(rc[0], 1), 4), "rr_1deg_0.1m": round(rc[3] / max(rc[0], 1), 4),
                                "icp_avg_ms": round(1e3 * float(np.mean([x[0] for x in icp_log])), 3) if icp_log else None,
                                "icp_avg_iterations": round(float(np.mean([x[1] for x in icp_log])), 1) if icp_log else None,
                                "icp_avg_fitness": round(float(np.mean([x[2] for x in icp_log])), 4) if icp_log else None,
                                "note": "recall after ICP refinement of the selected transform (reference evaluate.py:301-309), "
                                        "synthetic pairs; rank 0's ICP timings"}
    # ---- CPU baseline: the oracle (a port of the reference path) on this box's host cores, rank 0, N = 1 ----
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        from oracle import oracle as orc
        orc.lib()
        t_cpu = 0.0
        n_cpu = 0
        gates = lambda rre, rte: np.array([rre.size, ((rre <= 1.5) & (rte <= 0.6)).sum(), ((rre <= 1.5) & (rte <= 0.3)).sum(),
                                           ((rre <= 1.0) & (rte <= 0.1)).sum()], np.float64)
        q_cpu, q_gpu = np.zeros(4), np.zeros(4)
        for i in range(a.cpu_pairs):
            if t_cpu > 10.0:
                break
            n_cpu += 1
            e = pool[i % len(pool)]
            tc = time.perf_counter()
            rre, rte = cpu_pair(orc, e.host, args, np.random.RandomState(7 + i))
            t_cpu += time.perf_counter() - tc
            q_cpu += gates(rre, rte)
            # the same pair with the same RNG seed through the HIP path (outside any timed region)
            og = evaluate.register_pair(e.src_pts, e.tgt_pts, e.src_feat, e.tgt_feat, args, rng=np.random.RandomState(7 + i),
                                        src_inds=e.src_inds, tgt_inds=e.tgt_inds)
            cg = torch.zeros(4, dtype=torch.int64, device=dev)
            ops.hypothesis_gates(og.rtume_tform[0], e.gt, cg)
            q_gpu += cg.cpu().numpy()
        result["cpu_baseline"] = {"value": round(n_cpu / t_cpu, 4), "unit": "pairs/s", "cores": os.cpu_count(),
                                  "kind": "port",
                                  "sample": f"{n_cpu} full {a.config} pair(s) through the oracle's named path "
                                            f"(C/OpenMP scan+moments, numpy LAPACK/BLAS for QR/cdist/SVD), "
                                            f"{t_cpu:.1f} s wall",
                                  "quality_check": {
                                      "note": "fraction of RTUME hypotheses inside the gates (1.5deg,0.6m) / (1.5deg,0.3m) / (1deg,0.1m) on "
                                              "the SAME pairs and RNG seeds: CPU restatement of the reference vs this library",
                                      "cpu": [round(float(v), 4) for v in q_cpu[1:] / max(q_cpu[0], 1)],
                                      "gpu": [round(float(v), 4) for v in q_gpu[1:] / max(q_gpu[0], 1)]}}
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_pair(orc, p, args, rs):
    """The same named path on the CPU, including the tau-weighted draw (evaluate.py:233-245)."""
    src_kp = p.src_pts[p.src_inds]
    tgt_kp = p.tgt_pts[p.tgt_inds]
    ume_src = orc.ume_moments(p.src_pts, src_kp, p.src_feat, args.ume_max_nn, float(args.ume_r_nn), "f32")
    ume_tgt = orc.ume_moments(p.tgt_pts, tgt_kp, p.tgt_feat, args.ume_max_nn, float(args.ume_r_nn), "f32")
    D = orc.ume_cdist(ume_src[None], ume_tgt[None])[0]
    m = orc.row_argmin(D)
    d = D[np.arange(D.shape[0]), m]
    if args.filter_by_ume_dist_cond:
        prob = orc.match_prob(d, args.tau)
        cond = rs.choice(D.shape[0], min(D.shape[0], args.ume_n_samples), replace=False, p=prob)
    else:
        cond = np.arange(D.shape[0])
    T, _ = orc.batch_estimate_transform_ume_old(ume_src[cond], ume_tgt[m[cond]], with_dist=False)
    R_gt = np.broadcast_to(p.gt_tform[:3, :3], T[:, :3, :3].shape)
    rre = orc.relative_rotation_error(T[:, :3, :3], R_gt)
    rte = np.linalg.norm(T[:, :3, 3] - p.gt_tform[:3, 3], axis=-1)
    return rre, rte


if __name__ == "__main__":
    main()
