"""Builds probe variants of the coarse matching kernel (UMEREG_COARSE_PROBE=0..4) and times each."""
import os, sys, subprocess, ctypes, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np, torch
from umeregrobust_amd import _build

VARIANTS = {  # name -> extra compile flags
    "full": [],
    "p3": ["-DUMEREG_COARSE_PROBE=3"],
    "p0": ["-DUMEREG_COARSE_PROBE=0"],
}

def build_all():
    for name, flags in VARIANTS.items():
        out = os.path.join(ROOT, "tools", f"libumereg_{name}.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-fvisibility=hidden", "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include")] + flags + _build.sources() + ["-o", out]
        subprocess.check_call(cmd)
        print("built", out)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "build":
        build_all(); sys.exit(0)
    dev = torch.device("cuda:0")
    n = 10000
    rng = np.random.RandomState(0)
    u1 = torch.from_numpy(rng.standard_normal((n, 32, 4)).astype(np.float32)).to(dev)
    u2 = torch.from_numpy(rng.standard_normal((n, 32, 4)).astype(np.float32)).to(dev)
    for lv in os.environ.get("VARIANTS", ",".join(VARIANTS)).split(","):
        lib = ctypes.CDLL(os.path.join(ROOT, "tools", f"libumereg_{lv}.so"))
        lib.umereg_qbasis_bytes.restype = ctypes.c_size_t
        lib.umereg_ume_match_q_scratch_bytes.restype = ctypes.c_size_t
        v = ctypes.c_void_p
        qa = lib.umereg_qbasis_bytes(n, 3); qb = lib.umereg_qbasis_bytes(n, 4); sc = lib.umereg_ume_match_q_scratch_bytes(n, n)
        ws = torch.zeros(qa + qb + sc, dtype=torch.uint8, device=dev)
        m = torch.zeros(n, dtype=torch.int64, device=dev); d = torch.zeros(n, device=dev)
        st = v(torch.cuda.current_stream().cuda_stream)
        assert lib.umereg_ume_orthobasis_f32(v(u1.data_ptr()), n, 3, v(ws.data_ptr()), st) == 0
        assert lib.umereg_ume_orthobasis_f32(v(u2.data_ptr()), n, 4, v(ws.data_ptr() + qa), st) == 0
        def run():
            rc = lib.umereg_ume_match_q_f16r(v(ws.data_ptr()), v(ws.data_ptr() + qa), n, n, v(m.data_ptr()), v(d.data_ptr()),
                                             v(ws.data_ptr() + qa + qb), ctypes.c_size_t(sc), st)
            assert rc == 0
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        print(f"variant {lv}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per call (memset + coarse [+ refine])")
