"""umeregrobust_amd -- MI355X-native (gfx950) implementation of UMERegRobust's registration hot path.

Keeps the reference's Python surface for that path (evaluate.my_ume_generation,
utils.loc_utils.{ume_cdist, batch_estimate_transform_ume_old, ume_kp_layer, FeatureCorrelator},
utils.eval_utils.relative_rotation_error) on top of hand-written HIP kernels reached through
the C ABI in include/umereg.h.  No CPU fallback: compute ops need the built extension and a GPU.
"""
from . import _lib
from ._build import LIB_PATH, build_native

__version__ = "0.1.0"


def require_native():
    """Load libumereg.so or raise; called by every GPU test and by __graft_entry__.smoke()."""
    lib = _lib.load()
    return lib


def native_loaded():
    return _lib._lib is not None
