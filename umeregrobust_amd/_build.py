"""Build recipe for the native library (hipcc, gfx950 only; cross-compiles without a GPU)."""
import glob
import hashlib
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


HASH_MARKER = b"UMEREG_SRC_HASH="


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def dependencies():
    """Everything the binary is made of: HIP sources, private and public headers."""
    return sorted(sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")))


def source_hash():
    """sha256 over the names and contents of every dependency and the compiler flags: the identity of a build.
    It is compiled INTO the library (-DUMEREG_SOURCE_HASH, exported as umereg_build_source_hash()), so a shared object can
    be checked against the tree it claims to come from."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS[:-2]).encode())             # (without the -I path: the same sources build the same anywhere)
    for f in dependencies():
        h.update(os.path.basename(f).encode() + b"\0" + open(f, "rb").read() + b"\0")
    return h.hexdigest()


def embedded_hash(path=None):
    """The source hash a built library carries, read from the file (no dlopen: loading the library before torch would
    bring a second HIP runtime into the process, see _lib.py).  None if the file is missing or carries none."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    data = open(path, "rb").read()
    i = data.find(HASH_MARKER)
    if i < 0:
        return None
    return data[i + len(HASH_MARKER):i + len(HASH_MARKER) + 64].decode("ascii", "replace")


def is_stale():
    """True unless the in-tree library was compiled from exactly the sources (and flags) in the tree now."""
    return embedded_hash() != source_hash()


def build_native(force=False, verbose=False):
    """Compile every HIP source into csrc/libumereg.so (in-tree, so it travels with the repo).  A library whose embedded
    source hash matches the tree is reused; anything else (older sources, another flag set, no hash) is rebuilt."""
    if not force and not is_stale():
        return LIB_PATH
    want = source_hash()
    cmd = [HIPCC] + FLAGS + [f'-DUMEREG_SOURCE_HASH="{want}"'] + sources() + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    got = embedded_hash()
    if got != want:
        raise RuntimeError(f"{LIB_PATH}: built library carries source hash {got}, expected {want}")
    return LIB_PATH
