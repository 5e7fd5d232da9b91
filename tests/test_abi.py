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
