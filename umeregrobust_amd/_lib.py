"""ctypes loader for csrc/libumereg.so -- the C ABI declared in include/umereg.h.

There is no fallback of any kind: if the shared object is missing or a symbol is absent the
import of the ops fails loudly, and on a box with a GPU every op raises if the library reports
an error.
"""
import ctypes
import os

# torch must come first: PyTorch-ROCm bundles its own libamdhip64.so; loading it before
# libumereg.so makes our DT_NEEDED libamdhip64.so.7 resolve to THAT runtime, so torch's streams
# and device pointers are valid inside our kernels.  (Loaded the other way round the process
# holds two HIP runtimes and ours sees no device.)
import torch  # noqa: F401

from ._build import LIB_PATH

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
c_double = ctypes.c_double

# name -> (restype, argtypes); mirrors include/umereg.h one to one
SIGNATURES = {
    "umereg_abi_version": (c_int, []),
    "umereg_build_source_hash": (ctypes.c_char_p, []),
    "umereg_last_error": (ctypes.c_char_p, []),
    "umereg_device_count": (c_int, [ctypes.c_char_p, c_size_t]),
    "umereg_streams_run_side_by_side": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p]),
    "umereg_ball_query_workspace_bytes": (c_size_t, [c_int, c_int]),
    "umereg_ball_query_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_ball_query_ex_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                         c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_ume_moments_workspace_bytes": (c_size_t, [c_int, c_int]),
    "umereg_ume_moments_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_pack_points_f32": (c_int, [c_void_p, c_int, c_int, c_float, c_void_p, c_size_t, c_void_p]),
    "umereg_ume_moments_packed_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                              c_float, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "umereg_ume_keypoint_order": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "umereg_ume_dist_q_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    "umereg_ume_dist_q_f16x2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p]),
    "umereg_ume_match_f16x2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "umereg_ume_match_q_scratch_bytes": (c_size_t, [c_int, c_int]),
    "umereg_ume_match_q_f16r": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_void_p]),
    "umereg_ume_match_reset_f16": (c_int, [c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "umereg_ume_match_coarse_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "umereg_ume_match_refine_f16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p,
                                            c_void_p]),
    "umereg_ume_match_f16r": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "umereg_qbasis_bytes": (c_size_t, [c_int, c_int]),
    "umereg_ume_orthobasis_f32": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "umereg_ume_cdist_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "umereg_ume_cdist_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_ume_match_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "umereg_ume_match_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                     c_size_t, c_void_p]),
    "umereg_pair_match_workspace_bytes": (c_size_t, [c_int, c_int]),
    "umereg_pair_match_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_match_prob_f32": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p]),
    "umereg_ume_svdvals_f32": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "umereg_icp_workspace_bytes": (c_size_t, [c_int, c_int]),
    "umereg_icp_point_to_point_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_double, c_double,
                                              c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_icp_point_to_point_dev_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_double, c_double,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_icp_state_bytes": (c_size_t, []),
    "umereg_icp_enqueue_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_float, c_int, c_double, c_double, c_int, c_int,
                                       c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_icp_state_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "umereg_host_choice_round": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "umereg_host_choice_check": (c_int, [c_void_p, c_int, c_void_p]),
    "umereg_host_choice_mt19937": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                           c_void_p]),
    "umereg_pair_match_graph_create": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "umereg_pair_match_graph_launch": (c_int, [c_void_p, c_void_p]),
    "umereg_pair_match_graph_destroy": (c_int, [c_void_p]),
    "umereg_pair_match_graph_launch_ex": (c_int, [c_void_p, c_void_p, c_void_p]),
    "umereg_pair_match_graph_launch_from": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "umereg_pair_match_graph_solve": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    # per-call matcher options (umereg_match_opts*; None = defaults)
    "umereg_ume_match_q_scratch_bytes_ex": (c_size_t, [c_int, c_int, c_void_p]),
    "umereg_ume_match_q_f16r_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t,
                                           c_void_p, c_void_p]),
    "umereg_ume_match_coarse_f16_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "umereg_ume_match_refine_f16_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p,
                                               c_void_p, c_void_p]),
    "umereg_ume_match_workspace_bytes_ex": (c_size_t, [c_int, c_int, c_int, c_void_p]),
    "umereg_ume_match_f16r_ex": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                         c_size_t, c_void_p, c_void_p]),
    "umereg_pair_match_workspace_bytes_ex": (c_size_t, [c_int, c_int, c_void_p]),
    "umereg_pair_match_ex_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "umereg_pair_match_graph_create_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_void_p,
                                                  c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "umereg_pair_match_ragged_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                             c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p,
                                             c_void_p]),
    "umereg_pair_match_graph_create_cap": (c_int, [c_int, c_int, c_int, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                                   c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "umereg_pair_match_graph_launch_ragged": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                                      c_int, c_void_p, c_void_p]),
    "umereg_voxel_first_index_workspace_bytes": (c_size_t, [c_int]),
    "umereg_voxel_first_index_f32": (c_int, [c_void_p, c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_host_permutation_mt19937": (c_int, [c_void_p, c_void_p, ctypes.c_int64, ctypes.c_int64, c_void_p, c_void_p]),
    "umereg_rtume_solve_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p,
                                       c_void_p, c_void_p]),
    "umereg_hypothesis_gates_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "umereg_knn_workspace_bytes": (c_size_t, [c_int, c_int]),
    "umereg_knn_points_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                      c_size_t, c_void_p]),
    "umereg_nn1_pair_workspace_bytes": (c_size_t, [c_int, c_int]),
    "umereg_nn1_pair_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_feature_spatial_var_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                               c_size_t, c_void_p]),
    "umereg_corr_weighted_features_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p,
                                                  c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_corr_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "umereg_corr_scores_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                       c_float, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_corr_workspace_bytes_ex": (c_size_t, [c_int, c_int, c_int, c_int]),
    "umereg_corr_scores_ex_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                          c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "umereg_rre_deg_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "umereg_corr_scores_profile_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                               c_float, c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "umereg_corr_select_best_f32": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
}

ABI_VERSION = 2


class MatchOpts(ctypes.Structure):
    """umereg_match_opts (include/umereg.h): per-call options of the filter + refine matcher."""
    _fields_ = [("variant", ctypes.c_int32), ("splits", ctypes.c_int32), ("share_mask", ctypes.c_int64),
                ("force_exhaustive", ctypes.c_int32), ("reserved", ctypes.c_int32)]

    def __init__(self, variant=0, splits=0, share_mask=-1, force_exhaustive=0):
        super().__init__(int(variant), int(splits), int(share_mask), int(force_exhaustive), 0)

    def key(self):
        return (self.variant, self.splits, self.share_mask, self.force_exhaustive)


def opts_ptr(opts):
    """ctypes argument for a `const umereg_match_opts*` parameter (None -> NULL = the defaults)."""
    return None if opts is None else ctypes.addressof(opts)


_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and type every symbol.  Raises NativeLibraryError when the .so is absent:
    build it with `python -c "import __graft_entry__ as g; g.build()"`."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: the HIP extension has not been built "
            "(run __graft_entry__.build()); umeregrobust_amd has no CPU fallback")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.umereg_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: {lib.umereg_abi_version()} != {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().umereg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
