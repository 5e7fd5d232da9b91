"""CPU: the C-ABI shared library builds, loads WITHOUT a GPU and exports every symbol include/umereg.h
declares; without a device every compute entry point refuses with UMEREG_ENODEV (no CPU fallback)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(REPO, "include", "umereg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(umereg_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    import umeregrobust_amd
    from umeregrobust_amd import _lib
    path = umeregrobust_amd.build_native()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    syms = header_symbols()
    assert len(syms) >= 20
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in include/umereg.h but not exported"
    # and the ctypes table mirrors the header one to one
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.load().umereg_abi_version() == _lib.ABI_VERSION == 2


def test_no_cpu_fallback_without_device():
    import torch
    from umeregrobust_amd import _lib, ops
    lib = _lib.load()
    if lib.umereg_device_count(None, 0) > 0:
        pytest.skip("a HIP device is visible; this test is about the GPU-less behaviour")
    buf = (ctypes.c_float * 1024)()
    out = (ctypes.c_float * 64)()
    rc = lib.umereg_rre_deg_f32(buf, buf, 4, out, None)
    assert rc == -2 and b"no CPU fallback" in lib.umereg_last_error()          # UMEREG_ENODEV
    assert lib.umereg_match_prob_f32(buf, 16, 0.05, out, None) == -2
    # argument errors are reported before the device probe
    assert lib.umereg_rre_deg_f32(None, buf, 4, out, None) == -1
    assert b"null" in lib.umereg_last_error()
    # the torch-facing ops refuse CPU tensors instead of computing on the host
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.ume_cdist(torch.zeros(1, 4, 32, 4), torch.zeros(1, 4, 32, 4))
    # size queries are pure host arithmetic and work anywhere
    assert lib.umereg_ume_moments_workspace_bytes(1, 50000) > 50000 * 32
    assert lib.umereg_qbasis_bytes(10000, 3) == 10112 * 512


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under umeregrobust_amd/ may import or load it."""
    pkg = os.path.join(REPO, "umeregrobust_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(root, f)).read().lower()
                assert "oracle" not in text, f"{os.path.join(root, f)} references the oracle"


def _kernels_of(path):
    text = open(path).read()
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"__launch_bounds__\s*\([^)]*\)", "", text)
    text = re.sub(r"__attribute__\s*\(\((?:[^()]|\([^()]*\))*\)\)", "", text)
    return re.findall(r"__global__\s+void\s+(\w+)\s*\(", text)


def test_corr_kernel_declarations_match_the_definitions():
    """The hypothesis-selection sources are five translation units: every kernel is DEFINED in exactly one of
    corr_knn / corr_consensus / corr_lattice / corr_leftover.hip and DECLARED exactly once in corr_kernels.h (what corr.hip,
    the file with the one call, launches through); corr.hip and the two headers define none."""
    csrc = os.path.join(REPO, "umeregrobust_amd", "csrc")
    defined = []
    for f in ("corr_knn.hip", "corr_consensus.hip", "corr_lattice.hip", "corr_leftover.hip"):
        names = [n for n in _kernels_of(os.path.join(csrc, f))]
        defined += [n for n in dict.fromkeys(names)]          # (explicit instantiations repeat a template's name in its own file)
    declared = _kernels_of(os.path.join(csrc, "corr_kernels.h"))
    assert len(defined) == len(set(defined)) >= 40, "a kernel is defined in two files"      # (42 names, 51 instantiations)
    assert len(declared) == len(set(declared)), "a kernel is declared twice"
    assert set(defined) == set(declared), (sorted(set(defined) ^ set(declared)))
    assert _kernels_of(os.path.join(csrc, "corr.hip")) == [] and _kernels_of(os.path.join(csrc, "corr_dev.h")) == []
    # ... and every launch in corr.hip names a declared kernel
    launched = set(re.findall(r"hipLaunchKernelGGL\(\(?(\w+)", open(os.path.join(csrc, "corr.hip")).read()))
    assert launched and launched <= set(declared), sorted(launched - set(declared))
