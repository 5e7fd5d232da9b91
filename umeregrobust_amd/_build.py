"""Build recipe for the native library (hipcc, gfx950 only; cross-compiles without a GPU)."""
import glob
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(CSRC, "libumereg.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the ball-query distance must round once per operation (bit-exact indices);
# kernels that want FMAs call fma()/fmaf() explicitly.
# -ffp-contract=off: the reference's arithmetic has no fused multiply-adds where bit-exactness matters;
# -fno-slp-vectorize: packed-f32 VALU (v_pk_fma_f32) issued beside MFMAs is slower than the scalar pair
# on gfx950, so adjacent scalar f32 ops must not be re-packed behind our back (device side only: the
# host-side sampler in api.hip relies on the SLP vectoriser for its lock-step binary searches).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Xarch_device", "-fno-slp-vectorize", "-fPIC", "-shared",
         "-fvisibility=hidden", "-I", INCLUDE]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build_native(force=False, verbose=False):
    """Compile every HIP source into csrc/libumereg.so (in-tree, so it travels with the repo)."""
    if not force and not is_stale():
        return LIB_PATH
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH
