"""umeregrobust_amd -- MI355X-native (gfx950) implementation of UMERegRobust's registration hot path.

Keeps the reference's Python surface for that path (evaluate.my_ume_generation,
utils.loc_utils.{ume_cdist, batch_estimate_transform_ume_old, ume_kp_layer, FeatureCorrelator},
utils.eval_utils.relative_rotation_error) on top of hand-written HIP kernels reached through
the C ABI in include/umereg.h.  No CPU fallback: compute ops need the built extension and a GPU.
"""
import importlib

from ._build import LIB_PATH, build_native

__version__ = "0.1.0"


def __getattr__(name):
    # `umeregrobust_amd._lib` (the ctypes loader) imports torch; it is loaded on first use so that `umeregrobust_amd.hostpin`
    # can pin a rank's host threads BEFORE torch / numpy create their pools (bench.py, evaluate.main under torch.distributed.run)
    if name == "_lib":
        return importlib.import_module("._lib", __name__)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def require_native():
    """Load libumereg.so or raise; called by every GPU test and by __graft_entry__.smoke()."""
    return importlib.import_module("._lib", __name__).load()


def native_loaded():
    return importlib.import_module("._lib", __name__)._lib is not None
