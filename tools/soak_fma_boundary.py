"""How many ball-query decisions of KITTI-shaped clouds depend on whether `dist2 += diff * diff` is contracted into FMAs
(pytorch3d's CUDA kernel under nvcc's default) or rounded once per operation (pytorch3d's CPU kernel, this build, the oracle)?
CPU only (C + OpenMP, tools/probe/fma_boundary.c).  usage: python tools/soak_fma_boundary.py [pairs=50] [config=KT] [jitter_m=0]
(the synthetic clouds sit on a 0.3 m lattice, where no squared distance comes near 25; jitter_m > 0 moves every point off it by
U(-jitter, jitter) per axis: generic coordinates, as the first-point-per-voxel thinning of real scans leaves them)"""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from umeregrobust_amd.synth import synth_pair_cfg   # noqa: E402

so = os.path.join(HERE, "probe", "libfma_boundary.so")
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", so,
                       os.path.join(HERE, "probe", "fma_boundary.c"), "-lm"])
lib = ctypes.CDLL(so)
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 50
config = sys.argv[2] if len(sys.argv) > 2 else "KT"
jitter = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
tot = np.zeros(4, np.int64)
balls = 0
for seed in range(n_pairs):
    p = synth_pair_cfg(seed, config, "test" if seed % 2 == 0 else "rot")
    for pts, inds in ((p.src_pts, p.src_inds), (p.tgt_pts, p.tgt_inds)):
        pts = np.ascontiguousarray(pts, np.float32)
        if jitter > 0:
            pts = (pts + np.random.RandomState(seed).uniform(-jitter, jitter, pts.shape)).astype(np.float32)
        kp = np.ascontiguousarray(pts[inds], np.float32)
        out = np.zeros(4, np.int64)
        lib.fma_boundary_count(pts.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(len(pts)), kp.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_int64(len(kp)), ctypes.c_float(5.0), out.ctypes.data_as(ctypes.c_void_p))
        tot += out
        balls += len(kp)
    if seed % 10 == 9:
        print(seed + 1, "pairs:", balls, "balls", tot.tolist(), flush=True)
print(json.dumps({"config": config, "jitter_m": jitter, "pairs": n_pairs, "balls": balls, "distance_tests": int(tot[0]),
                  "tests_where_contracted_and_uncontracted_predicates_differ": int(tot[1]),
                  "tests_with_dist2_within_2ulp_of_r2": int(tot[2]), "balls_with_a_differing_test": int(tot[3])}))
