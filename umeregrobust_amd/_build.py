"""Build recipe for the native library (hipcc, gfx950 only; cross-compiles without a GPU).

Every HIP source is its own translation unit: compiled to an object under csrc/.obj/ (cached by the sha256 of the
source, the headers, the flags and the compiler version; units compile in parallel), then linked into csrc/libumereg.so.
Touching one kernel file recompiles that file only."""
import glob
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")
LIB_PATH = os.path.join(CSRC, "libumereg.so")
OBJ_DIR = os.path.join(CSRC, ".obj")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: the ball-query distance must round once per operation (bit-exact indices);
# kernels that want FMAs call fma()/fmaf() explicitly.
# -ffp-contract=off: the reference's arithmetic has no fused multiply-adds where bit-exactness matters;
# -fno-slp-vectorize: packed-f32 VALU (v_pk_fma_f32) issued beside MFMAs is slower than the scalar pair
# on gfx950, so adjacent scalar f32 ops must not be re-packed behind our back (device side only: the
# host-side sampler in api.hip relies on the SLP vectoriser for its lock-step binary searches).
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-Xarch_device", "-fno-slp-vectorize", "-fPIC",
          "-fvisibility=hidden"]
# -z defs: a kernel declared in corr_kernels.h but defined with another signature (or not at all) is an undefined host stub --
# a link error here instead of a load error on the GPU box
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden", "-Wl,-z,defs"]
FLAGS = CFLAGS + ["-shared", "-I", INCLUDE]      # (the one-command form of the same build; kept for the record in logs)

HASH_MARKER = b"UMEREG_SRC_HASH="
_cache = {}     # per process: (path, mtime_ns, size) -> sha256 of the file; "toolchain" -> hipcc --version digest


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(INCLUDE, "*.h")))


def dependencies():
    """Everything the binary is made of: HIP sources, private and public headers."""
    return sorted(sources() + headers())


def _file_digest(path):
    st = os.stat(path)
    key = (path, st.st_mtime_ns, st.st_size)
    d = _cache.get(key)
    if d is None:
        h = hashlib.sha256()
        with open(path, "rb") as f:
            for chunk in iter(lambda: f.read(1 << 20), b""):
                h.update(chunk)
        d = _cache[key] = h.hexdigest()
    return d


def toolchain_id():
    """Digest of `hipcc --version` (the compiler that would build the library here): part of the object-cache key.  It is NOT
    part of source_hash(): the library travels to the GPU box, where only the identity of the SOURCES can be re-checked."""
    d = _cache.get("toolchain")
    if d is None:
        try:
            out = subprocess.run([HIPCC, "--version"], capture_output=True, timeout=60).stdout
        except (OSError, subprocess.SubprocessError):
            out = b"unknown"
        d = _cache["toolchain"] = hashlib.sha256(HIPCC.encode() + b"\0" + out).hexdigest()
    return d


def source_hash(extra_flags=()):
    """sha256 over the names and contents of every dependency and the compiler flags: the identity of a build.
    It is compiled INTO the library (-DUMEREG_SOURCE_HASH, exported as umereg_build_source_hash()), so a shared object can
    be checked against the tree it claims to come from.  extra_flags: the `-D...` switches of an A/B variant (build_native(extra_flags=
    ..., out=tools/lib*.so)) are part of ITS identity: a counter pass taken on a variant library carries another hash than the product's
    and cannot pass for a profile of the product (bench.py's counters_match_library)."""
    h = hashlib.sha256()
    h.update(" ".join(CFLAGS + list(extra_flags) + ["-shared"]).encode())    # (without the -I path: the same sources build the same anywhere)
    for f in dependencies():
        h.update(os.path.basename(f).encode() + b"\0" + _file_digest(f).encode() + b"\0")
    return h.hexdigest()


def embedded_hash(path=None):
    """The source hash a built library carries, read from the file (no dlopen: loading the library before torch would
    bring a second HIP runtime into the process, see _lib.py).  The record is `UMEREG_SRC_HASH=` + 64 hex digits; the LAST
    well-formed occurrence counts (the marker string itself also sits in .rodata of whoever formats an error message).
    None if the file is missing or carries none."""
    path = path or LIB_PATH
    if not os.path.exists(path):
        return None
    st = os.stat(path)
    key = ("embedded", path, st.st_mtime_ns, st.st_size)
    if key in _cache:
        return _cache[key]
    import mmap
    found = None
    with open(path, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as m:
        i = m.find(HASH_MARKER)
        while i >= 0:
            cand = m[i + len(HASH_MARKER):i + len(HASH_MARKER) + 64]
            if len(cand) == 64 and all(c in b"0123456789abcdef" for c in cand):
                found = cand.decode("ascii")
            i = m.find(HASH_MARKER, i + 1)
    _cache[key] = found
    return found


def is_stale():
    """True unless the in-tree library was compiled from exactly the sources (and flags) in the tree now."""
    return embedded_hash() != source_hash()


def _unit_key(src, extra):
    h = hashlib.sha256()
    h.update((" ".join(CFLAGS + extra) + "\0" + toolchain_id() + "\0").encode())
    for f in [src] + headers():
        h.update(os.path.basename(f).encode() + b"\0" + _file_digest(f).encode() + b"\0")
    return h.hexdigest()


def _compile_unit(src, extra, verbose):
    name = os.path.splitext(os.path.basename(src))[0]
    obj = os.path.join(OBJ_DIR, name + ".o")
    stamp = obj + ".key"
    key = _unit_key(src, extra)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
        return obj, False
    cmd = [HIPCC] + CFLAGS + extra + ["-I", INCLUDE, "-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(key)
    return obj, True


def build_native(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source (in parallel, objects cached under csrc/.obj/) and link csrc/libumereg.so (in-tree, so it
    travels with the repo).  A library whose embedded source hash matches the tree is reused; anything else (older sources,
    another flag set, no hash) is rebuilt.  extra_flags / out: another build of the same sources (ablation and A/B builds of
    the tools: `-D...` switches, output under tools/), with its own object directory."""
    global OBJ_DIR
    out = out or LIB_PATH
    extra_flags = list(extra_flags)
    if extra_flags and out == LIB_PATH:
        # the embedded hash covers sources and the default flags only: a variant linked to the product's path would look current
        # to is_stale() and be loaded silently by the next plain build_native()
        raise ValueError("build_native(extra_flags=...) builds an A/B variant: give it its own `out` (e.g. tools/lib<name>.so), not the product library")
    if out == LIB_PATH and not extra_flags and not force and not is_stale():
        return LIB_PATH
    want = source_hash(extra_flags)
    obj_dir_default = OBJ_DIR
    if out != LIB_PATH or extra_flags:
        OBJ_DIR = os.path.join(CSRC, ".obj-" + hashlib.sha256((out + " ".join(extra_flags)).encode()).hexdigest()[:12])
    try:
        os.makedirs(OBJ_DIR, exist_ok=True)
        if force:
            for f in glob.glob(os.path.join(OBJ_DIR, "*.key")):
                os.remove(f)
        srcs = sources()
        with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 1)) as ex:
            # only api.hip embeds the source hash: a change elsewhere must not invalidate every object
            futs = [ex.submit(_compile_unit, s, extra_flags + ([f'-DUMEREG_SOURCE_HASH="{want}"'] if os.path.basename(s) == "api.hip" else []),
                              verbose) for s in srcs]
            objs = [f.result()[0] for f in futs]
        cmd = [HIPCC] + LDFLAGS + objs + ["-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    finally:
        OBJ_DIR = obj_dir_default
    got = embedded_hash(out)
    if got != want:
        raise RuntimeError(f"{out}: built library carries source hash {got}, expected {want}")
    return out
