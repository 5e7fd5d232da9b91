"""grid-build probe: ops.knn_points on clouds of several sizes (each call builds one search structure); run under
rocprofv3 --kernel-trace --stats to read the build kernels' durations (tools/grid_build_stats.sh)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from umeregrobust_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
for N in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "10000,50000").split(",")]:
    pts = torch.from_numpy((rng.uniform(-50, 50, (1, N, 3)) * np.array([1, 1, 0.05])).astype(np.float32)).to(dev)
    q = pts[:, :256].contiguous()
    for _ in range(6):
        ops.knn_points(q, pts, K=1)
    torch.cuda.synchronize()
    print("N", N, "done", flush=True)
