"""Loader for libmsmhip.so (the C ABI declared in include/msmhip.h).

There is no Python or CPU fallback: if the shared library is missing the import fails loudly, and
every device entry point fails with MSM_ERR_NOGPU on a machine without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSM_LIB_PATH") or os.path.join(_HERE, "libmsmhip.so")  # override: A/B builds while profiling

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)

MSM_OK = 0
ERR_NAMES = {-1: "MSM_ERR_INVALID", -2: "MSM_ERR_HIP", -3: "MSM_ERR_OUTSIDE", -4: "MSM_ERR_NOTFOUND", -5: "MSM_ERR_NOGPU",
             -6: "MSM_ERR_STATE", -7: "MSM_ERR_CAPACITY", -8: "MSM_ERR_ROTATION"}


class MsmError(RuntimeError):
    """Raised for any non-zero status of the C ABI (the reference throws MeshException / MeshregException)."""

    def __init__(self, code, message):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "MSM_ERR"), code, message))
        self.code = code


class GroupParams(C.Structure):
    _fields_ = [("simmeasure", C.c_int32), ("fixnan", C.c_int32), ("lambda_", C.c_double), ("mu", C.c_double),
                ("kappa", C.c_double), ("k_exp", C.c_double), ("rexp", C.c_double), ("range", C.c_double), ("percentile", C.c_double)]


class CostParams(C.Structure):
    _fields_ = [("kind", C.c_int32), ("simmeasure", C.c_int32), ("rmode", C.c_int32), ("reserved", C.c_int32),
                ("lambda_", C.c_double), ("mu", C.c_double), ("kappa", C.c_double), ("k_exp", C.c_double),
                ("rexp", C.c_double), ("range", C.c_double), ("percentile", C.c_double)]


# name -> (restype, argtypes); mirrors include/msmhip.h one to one
_VP = C.c_void_p
SIGNATURES = {
    "msm_abi_version": (C.c_int, []),
    "msm_last_error": (C.c_char_p, []),
    "msm_device_count": (C.c_int, []),
    "msm_icosphere_counts": (C.c_int, [C.c_int, c_ip, c_ip]),
    "msm_icosphere": (C.c_int, [C.c_int, C.c_double, c_dp, c_ip]),
    "msm_mesh_adjacency": (C.c_int, [c_ip, C.c_int32, C.c_int32, c_ip, c_ip, c_ip, c_ip]),
    "msm_vertex_areas": (C.c_int, [c_dp, c_ip, C.c_int32, C.c_int32, c_dp]),
    "msm_resample_anatomy_grid": (C.c_int, [c_dp, C.c_int32, c_ip, C.c_int32, C.c_int32, C.c_double, c_ip, c_ip, c_dp, c_ip, c_ip, c_ip, c_dp, c_ip, c_ip]),
    "msm_cp_spacings": (C.c_int, [c_dp, c_ip, C.c_int32, C.c_int32, c_dp, c_dp]),
    "msm_label_sampling_grid": (C.c_int, [C.c_int, C.c_double, C.c_int, C.c_int32, c_dp, c_ip, c_dp, c_ip]),
    "msm_rescale_sampling_grid": (C.c_int, [c_dp, C.c_int32, c_dp, c_dp]),
    "msm_rotation_matrix": (C.c_int, [c_dp, c_dp, c_dp]),
    "msm_cp_rotations": (C.c_int, [c_dp, c_dp, C.c_int32, c_dp]),
    "msm_estimate_triplets": (C.c_int, [c_ip, C.c_int32, c_ip]),
    "msm_estimate_pairs": (C.c_int, [c_ip, C.c_int32, C.c_int32, c_ip]),
    "msm_octree_signature": (C.c_int, [c_dp, c_ip, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]),
    "msm_ray_table_check": (C.c_int, [c_dp, c_ip, C.c_int32, C.c_int32, C.c_int32, C.c_uint64, C.POINTER(C.c_int64)]),
    "msm_ctx_create": (_VP, [C.c_int]),
    "msm_ctx_create_on_stream": (_VP, [C.c_int, _VP]),
    "msm_ctx_destroy": (None, [_VP]),
    "msm_ctx_synchronize": (C.c_int, [_VP]),
    "msm_ctx_stream": (_VP, [_VP]),
    "msm_store_release_i64": (None, [_VP, C.c_int64]),
    "msm_load_acquire_i64": (C.c_int64, [_VP]),
    "msm_min_acquire_i64": (C.c_int64, [_VP, C.c_int32]),
    "msm_ctx_time_queries": (C.c_int, [_VP, C.c_int]),
    "msm_ctx_query_kernel_ms": (C.c_int, [_VP, c_dp]),
    "msm_query_lanes": (C.c_int, [C.c_int64]),
    "msm_group_context": (_VP, [_VP]),
    "msm_ctx_wait_stream": (C.c_int, [_VP, _VP]),
    "msm_ctx_staging_stats": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "msm_host_alloc": (_VP, [_VP, C.c_size_t]),
    "msm_host_free": (None, [_VP, _VP]),
    "msm_host_register": (C.c_int, [_VP, _VP, C.c_size_t]),
    "msm_mesh_create": (_VP, [_VP, c_dp, C.c_int32, c_ip, C.c_int32]),
    "msm_mesh_destroy": (None, [_VP]),
    "msm_mesh_update_coords": (C.c_int, [_VP, c_dp]),
    "msm_mesh_get_coords": (C.c_int, [_VP, c_dp]),
    "msm_mesh_set_features": (C.c_int, [_VP, c_dp, C.c_int32]),
    "msm_mesh_sizes": (C.c_int, [_VP, c_ip, c_ip, c_ip]),
    "msm_mesh_octree_stats": (C.c_int, [_VP, c_lp]),
    "msm_mesh_octree_signature": (C.c_int, [_VP, c_lp, C.POINTER(C.c_uint64)]),
    "msm_octree_forest_signatures": (C.c_int, [_VP, c_dp, C.c_int32, c_ip, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    "msm_query_triangles": (C.c_int, [_VP, c_dp, C.c_int32, c_ip, c_ip, c_dp, C.c_int]),
    "msm_closest_vertex": (C.c_int, [_VP, c_dp, C.c_int32, c_ip]),
    "msm_adaptive_barycentric_weights": (C.c_int, [_VP, _VP, c_dp, c_ip, c_ip, c_dp, C.c_int64, c_lp]),
    "msm_metric_resample": (C.c_int, [_VP, c_dp, C.c_int32, _VP, c_dp, c_dp, c_dp]),
    "msm_create_exclusion": (C.c_int, [c_dp, C.c_int32, C.c_int32, C.c_double, C.c_double, c_dp]),
    "msm_sphere_project_warp": (C.c_int, [_VP, c_dp, c_dp, C.c_int32]),
    "msm_mesh_sphere_project_warp": (C.c_int, [_VP, _VP, c_dp]),
    "msm_barycentric_coords_resample": (C.c_int, [_VP, c_dp, c_dp, C.c_int32, c_dp]),
    "msm_smooth_data": (C.c_int, [_VP, c_dp, C.c_int32, _VP, C.c_double, c_dp, c_dp, c_dp]),
    "msm_mesh_unfold": (C.c_int, [_VP, C.c_double, c_ip, c_ip]),
    "msm_mesh_prepare_search": (C.c_int, [_VP, C.c_int, c_ip]),
    "msm_variance_normalise": (C.c_int, [c_dp, C.c_int32, C.c_int32, c_dp]),
    "msm_mcmc_optimise": (C.c_int, [c_dp, c_dp, c_ip, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_int32, C.c_uint64, c_ip]),
    "msm_fusion_icm_step": (C.c_int, [c_dp, c_dp, c_ip, C.c_int32, c_dp, c_ip, C.c_int32, C.c_int32, C.c_int32, c_ip]),
    "msm_pairwise_icm": (C.c_int, [c_dp, c_dp, c_ip, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_ip]),
    "msm_nearest_neighbour": (C.c_int, [_VP, c_dp, C.c_int32, c_dp, C.c_int32, c_dp, c_dp, c_dp]),
    "msm_cost_create": (_VP, [_VP, C.POINTER(CostParams)]),
    "msm_cost_destroy": (None, [_VP]),
    "msm_cost_set_meshes": (C.c_int, [_VP, _VP, _VP, _VP]),
    "msm_cost_set_anatomical": (C.c_int, [_VP, _VP, c_dp, c_dp, C.c_int32, c_ip, C.c_int32, c_ip, c_ip, c_dp, c_ip, c_ip]),
    "msm_cost_reset_source": (C.c_int, [_VP, _VP]),
    "msm_cost_reset_cpgrid": (C.c_int, [_VP, _VP]),
    "msm_cost_set_source_features": (C.c_int, [_VP, c_dp, C.c_int32]),
    "msm_cost_set_cfweight": (C.c_int, [_VP, c_dp, C.c_int32]),
    "msm_cost_set_spacings": (C.c_int, [_VP, c_dp, C.c_double]),
    "msm_cost_set_labels": (C.c_int, [_VP, c_dp, C.c_int32, c_dp]),
    "msm_cost_set_triplets": (C.c_int, [_VP, c_ip, C.c_int32]),
    "msm_cost_set_pairs": (C.c_int, [_VP, c_ip, C.c_int32]),
    "msm_cost_get_source_data": (C.c_int, [_VP]),
    "msm_cost_patches": (C.c_int, [_VP, c_ip, c_ip, c_ip, C.c_int64]),
    "msm_cost_absolute_weights": (C.c_int, [_VP, c_dp]),
    "msm_cost_unary_table": (C.c_int, [_VP, c_dp]),
    "msm_cost_unary_table_async": (C.c_int, [_VP]),
    "msm_cost_unary_table_fetch": (C.c_int, [_VP, c_dp]),
    "msm_cost_unary_batch": (C.c_int, [_VP, c_ip, c_ip, C.c_int32, c_dp]),
    "msm_cost_triplet_batch": (C.c_int, [_VP, c_ip, c_ip, c_ip, c_ip, C.c_int32, c_dp]),
    "msm_cost_triplet_octets_prefetch": (C.c_int, [_VP, _VP, C.c_int32, _VP]),
    "msm_cost_prefetch_stats": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "msm_cost_triplet_octets": (C.c_int, [_VP, _VP, C.c_int32, _VP]),  # addresses as integers: called once per label step (ctypes casts cost microseconds)
    "msm_cost_pairwise_batch": (C.c_int, [_VP, c_ip, c_ip, c_ip, C.c_int32, c_dp]),
    "msm_cost_pairwise_table": (C.c_int, [_VP, c_dp]),
    "msm_cost_triplet_table": (C.c_int, [_VP, C.c_int32, C.c_int32, c_dp]),
    "msm_cost_total": (C.c_int, [_VP, c_ip, c_dp, c_dp]),
    "msm_cost_enable_timing": (C.c_int, [_VP, C.c_int]),
    "msm_cost_kernel_times": (C.c_int, [_VP, c_dp, C.c_int32, c_ip]),
    "msm_cost_counters": (C.c_int, [_VP, c_lp]),
    "msm_group_create": (_VP, [_VP, C.POINTER(GroupParams), C.c_int32]),
    "msm_group_fusion_move": (C.c_int, [_VP, c_ip, C.c_int32, c_dp, c_dp]),
    "msm_group_destroy": (None, [_VP]),
    "msm_group_time_moves": (C.c_int, [_VP, C.c_int]),
    "msm_group_move_kernels_ms": (C.c_int, [_VP, c_dp]),
    "msm_group_set_template": (C.c_int, [_VP, _VP, c_dp]),
    "msm_group_set_controlgrid": (C.c_int, [_VP, c_dp, c_ip, C.c_int32, C.c_int32]),
    "msm_group_set_subject": (C.c_int, [_VP, C.c_int32, _VP, c_dp, C.c_int32]),
    "msm_group_reset_cpgrid": (C.c_int, [_VP, C.c_int32, c_dp]),
    "msm_group_set_labels": (C.c_int, [_VP, c_dp, C.c_int32]),
    "msm_group_set_pair_layout": (C.c_int, [_VP, C.c_int32]),
    "msm_group_set_rotation_mode": (C.c_int, [_VP, C.c_int32]),
    "msm_group_setup": (C.c_int, [_VP]),
    "msm_group_setup_subjects": (C.c_int, [_VP, c_ip, C.c_int32]),
    "msm_group_export_subject": (C.c_int, [_VP, C.c_int32, c_dp, c_ip, c_ip, C.c_int64, c_lp]),
    "msm_group_import_subject": (C.c_int, [_VP, C.c_int32, c_dp, c_ip, c_ip, C.c_int64]),
    "msm_group_export_subject_dev": (C.c_int, [_VP, C.c_int32, _VP, _VP, _VP, C.c_int64, c_lp]),
    "msm_group_import_subject_dev": (C.c_int, [_VP, C.c_int32, _VP, _VP, _VP, C.c_int64]),
    "msm_group_export_subjects_dev": (C.c_int, [_VP, c_ip, C.c_int32, _VP, C.c_int64, _VP, C.c_int64, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "msm_group_import_subjects_dev": (C.c_int, [_VP, c_ip, C.c_int32, _VP, C.c_int64, _VP, C.c_int64, _VP, C.c_int64, C.POINTER(C.c_int64)]),
    "msm_group_setup_more_subjects": (C.c_int, [_VP, c_ip, C.c_int32]),
    "msm_group_fusion_move_dev": (C.c_int, [_VP, c_ip, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_int64, _VP, _VP]),
    "msm_group_finalize": (C.c_int, [_VP]),
    "msm_group_sizes": (C.c_int, [_VP, c_ip, c_ip, c_ip]),
    "msm_group_dims": (C.c_int, [_VP, c_ip, c_ip, c_ip, c_ip, c_ip]),
    "msm_group_get_pairs": (C.c_int, [_VP, c_ip]),
    "msm_group_get_triplets": (C.c_int, [_VP, c_ip]),
    "msm_group_patch": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, c_ip, c_dp, C.c_int32, c_ip]),
    "msm_group_pairwise_batch": (C.c_int, [_VP, c_ip, c_ip, c_ip, C.c_int32, c_dp]),
    "msm_group_triplet_batch": (C.c_int, [_VP, c_ip, c_ip, c_ip, c_ip, C.c_int32, c_dp]),
}

_lib = None


def lib():
    """The loaded library.  Raises ImportError when libmsmhip.so has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("newmsm_amd: %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError = header / library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status):
    if status != MSM_OK:
        raise MsmError(status, lib().msm_last_error().decode("utf-8", "replace"))
    return status
