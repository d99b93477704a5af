"""ctypes binding of the C ABI declared in include/stevi_hip.h (libstevi_hip.so).

This is the only way the Python side reaches the HIP kernels; there is no CPU fallback.  Loading fails
loudly when the library has not been built (run `python -c "import __graft_entry__ as g; g.build()"` or
`make -C libstevi_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libstevi_hip.so")

SVH_MAX_DIMS = 4

# svh_status
OK, EMPTY_RESULT, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_OUT_OF_MEMORY = range(7)
# svh_memspace / svh_dtype
HOST, DEVICE = 0, 1
F32, I32, U32, U8, U64, I16, U16 = 0, 1, 2, 3, 4, 5, 6

# every symbol include/stevi_hip.h declares
EXPORTS = [
    "svh_context_create", "svh_context_destroy", "svh_context_set_stream", "svh_context_set_option", "svh_context_synchronize", "svh_context_trim",
    "svh_status_string", "svh_last_error", "svh_device_available", "svh_device_alloc", "svh_device_free", "svh_device_free_detached", "svh_device_cache_trim", "svh_context_get_device", "svh_device_upload", "svh_device_download", "svh_device_copy",
    "svh_host_alloc", "svh_host_free", "svh_host_cache_trim", "svh_host_is_pinned",
    "svh_test_set_option",  # include/stevi_hip_test.h
    "svh_profile_enable", "svh_profile_filter", "svh_profile_sampling", "svh_profile_reset", "svh_profile_collect", "svh_profile_count", "svh_profile_get",
    "svh_unfold", "svh_unfold_oriented", "svh_unfold_shape", "svh_census_features", "svh_census_transform",
    "svh_feature_cost_volume", "svh_unfold_cost_volume", "svh_unfold_cost_volume_minima", "svh_unfold_cost_volume_winner", "svh_sgm_cost_volume", "svh_sgm_cost_volume_minima", "svh_sgm_cost_volume_winner", "svh_sgm_cost_volume_textbook",
    "svh_extract_selected_index", "svh_selected_index_to_disp", "svh_selected_cost", "svh_truncated_cost_volume",
    "svh_refine_disp_cost_interpolation", "svh_stereo_match", "svh_keys_to_index", "svh_census_shard_keys",
    "svh_census_shard_region1_is_global", "svh_census_shard_finish", "svh_census_exchange_keys", "svh_census_band_match", "svh_unfold_cost_volume_2d", "svh_extract_selected_2d_index", "svh_selected_2d_index_to_disp",
    "svh_truncated_bidirectional_cost_volume", "svh_refine_disp_2d_cost_interpolation", "svh_refine_disp_2d_cost_patch_interpolation",
    "svh_on_demand_features", "svh_on_demand_truncated_cost_volume", "svh_cacheless_patch_match", "svh_cacheless_patch_match_init", "svh_patch_match",
    "svh_feature_cost_volume_2d", "svh_average_pooling_downsample", "svh_unfold_compressed", "svh_unfold_compressed_shape",
    "svh_channels_mean", "svh_channels_norm", "svh_channels_zero_mean_norm", "svh_zeromean_feature_volume", "svh_normalized_feature_volume",
    "svh_zeromean_normalized_feature_volume", "svh_feature_volume_for_match_func", "svh_guided_cost_volume", "svh_hierarchical_truncated_cost_volume",
]


class SvhOnDemandParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("match_func", "search_dims", "h_radius", "v_radius", "lower0", "upper0", "lower1", "upper1")]


class SvhArray(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("ndim", C.c_int32),
        ("dtype", C.c_int32),
        ("memspace", C.c_int32),
        ("reserved", C.c_int32),
        ("shape", C.c_int64 * SVH_MAX_DIMS),
        ("strides", C.c_int64 * SVH_MAX_DIMS),
    ]


class SvhStereoParams(C.Structure):
    _fields_ = [
        ("match_func", C.c_int32),
        ("disp_direction", C.c_int32),
        ("h_radius", C.c_int32),
        ("v_radius", C.c_int32),
        ("disp_lower", C.c_int32),
        ("disp_count", C.c_int32),
        ("sgm_directions", C.c_int32),
        ("P1", C.c_float),
        ("P2", C.c_float),
        ("Pout", C.c_float),
        ("margins", C.c_int32 * 4),
        ("refine_kernel", C.c_int32),
        ("refine_h_radius", C.c_int32),
        ("refine_v_radius", C.c_int32),
        ("shard_begin", C.c_int32),
        ("shard_count", C.c_int32),
    ]


class SvhError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libstevi_hip: {message} (status {status})")
        self.status = status


_lib = None


def load():
    """Loads libstevi_hip.so.  torch (when present) is imported first so that the library binds to the HIP
    runtime PyTorch already loaded (same SONAME libamdhip64.so.7) and streams / device pointers are shared."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP library has not been built. "
            "Run `make -C libstevi_amd/csrc` (needs hipcc); there is no CPU fallback.")
    try:
        import torch  # noqa: F401  (side effect: loads torch's libamdhip64)
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    A = P(SvhArray)
    ctx = C.c_void_p
    i32 = C.c_int32
    sig = {
        "svh_context_create": (C.c_int, [P(C.c_void_p), C.c_int, C.c_void_p]),
        "svh_context_destroy": (C.c_int, [ctx]),
        "svh_context_set_stream": (C.c_int, [ctx, C.c_void_p]),
        "svh_context_set_option": (C.c_int, [ctx, C.c_char_p, C.c_int]),
        "svh_test_set_option": (C.c_int, [ctx, C.c_char_p, C.c_int]),
        "svh_context_synchronize": (C.c_int, [ctx]),
        "svh_context_trim": (C.c_int, [ctx]),
        "svh_status_string": (C.c_char_p, [C.c_int]),
        "svh_last_error": (C.c_char_p, [ctx]),
        "svh_device_available": (C.c_int, []),
        "svh_profile_enable": (C.c_int, [ctx, C.c_int]),
        "svh_profile_filter": (C.c_int, [ctx, C.c_char_p]),
        "svh_profile_reset": (C.c_int, [ctx]),
        "svh_profile_collect": (C.c_int, [ctx]),
        "svh_profile_count": (C.c_int, [ctx]),
        "svh_profile_get": (C.c_int, [ctx, C.c_int, C.c_char_p, C.c_size_t, P(C.c_double), P(C.c_int64)]),
        "svh_unfold": (C.c_int, [ctx, A, C.c_int, C.c_int, P(i32), A]),
        "svh_unfold_shape": (C.c_int, [A, C.c_int, C.c_int, P(i32), P(C.c_int64)]),
        "svh_unfold_oriented": (C.c_int, [ctx, A, C.c_int, C.c_int, P(i32), C.c_int, A]),
        "svh_census_features": (C.c_int, [ctx, A, A]),
        "svh_census_transform": (C.c_int, [ctx, A, C.c_int, C.c_int, P(i32), A]),
        "svh_feature_cost_volume": (C.c_int, [ctx, C.c_int, C.c_int, A, A, i32, i32, A]),
        "svh_unfold_cost_volume": (C.c_int, [ctx, C.c_int, C.c_int, A, A, C.c_int, C.c_int, i32, i32, A]),
        "svh_sgm_cost_volume": (C.c_int, [ctx, C.c_int, C.c_int, A, C.c_float, C.c_float, P(i32), C.c_float, A]),
        "svh_sgm_cost_volume_textbook": (C.c_int, [ctx, C.c_int, C.c_int, A, C.c_float, C.c_float, P(i32), C.c_float, A]),
        "svh_unfold_cost_volume_minima": (C.c_int, [ctx, C.c_int, C.c_int, A, A, C.c_int, C.c_int, i32, i32, A, A, P(C.c_int)]),
        "svh_unfold_cost_volume_winner": (C.c_int, [ctx, C.c_int, C.c_int, A, A, C.c_int, C.c_int, i32, i32, A, A, P(C.c_int)]),
        "svh_sgm_cost_volume_minima": (C.c_int, [ctx, C.c_int, C.c_int, A, A, C.c_float, C.c_float, C.c_float, P(i32), C.c_float, A]),
        "svh_sgm_cost_volume_winner": (C.c_int, [ctx, C.c_int, C.c_int, A, A, C.c_int, C.c_float, C.c_float, C.c_float, P(i32), C.c_float, A, A, P(C.c_int)]),
        "svh_device_copy": (C.c_int, [ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
        "svh_census_exchange_keys": (C.c_int, [ctx, C.c_void_p, A, C.c_int]),
        "svh_context_get_device": (C.c_int, [ctx]),
        "svh_device_free_detached": (C.c_int, [C.c_int, C.c_void_p]),
        "svh_device_cache_trim": (C.c_int, [C.c_int]),
        "svh_device_alloc": (C.c_int, [ctx, C.c_size_t, P(C.c_void_p)]),
        "svh_device_free": (C.c_int, [ctx, C.c_void_p]),
        "svh_device_upload": (C.c_int, [ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
        "svh_device_download": (C.c_int, [ctx, C.c_void_p, C.c_void_p, C.c_size_t]),
        "svh_host_alloc": (C.c_int, [C.c_size_t, P(C.c_void_p)]),
        "svh_host_free": (C.c_int, [C.c_void_p]),
        "svh_host_cache_trim": (C.c_int, []),
        "svh_host_is_pinned": (C.c_int, [C.c_void_p, C.c_size_t]),
        "svh_extract_selected_index": (C.c_int, [ctx, C.c_int, A, A]),
        "svh_selected_index_to_disp": (C.c_int, [ctx, C.c_int, A, i32, A]),
        "svh_selected_cost": (C.c_int, [ctx, A, A, A]),
        "svh_truncated_cost_volume": (C.c_int, [ctx, C.c_int, C.c_int, A, A, C.c_int, C.c_int, C.c_int, A]),
        "svh_refine_disp_cost_interpolation": (C.c_int, [ctx, C.c_int, A, A, A]),
        "svh_stereo_match": (C.c_int, [ctx, P(SvhStereoParams), A, A, A, A, A, A, A]),
        "svh_keys_to_index": (C.c_int, [ctx, C.c_int, A, i32, A]),
        "svh_unfold_cost_volume_2d": (C.c_int, [ctx, C.c_int, C.c_int, A, A, C.c_int, C.c_int, i32, i32, i32, i32, A]),
        "svh_feature_cost_volume_2d": (C.c_int, [ctx, C.c_int, C.c_int, A, A, i32, i32, i32, i32, A]),
        "svh_extract_selected_2d_index": (C.c_int, [ctx, C.c_int, A, A]),
        "svh_selected_2d_index_to_disp": (C.c_int, [ctx, A, i32, i32, A]),
        "svh_truncated_bidirectional_cost_volume": (C.c_int, [ctx, A, A, C.c_int, C.c_int, A]),
        "svh_refine_disp_2d_cost_interpolation": (C.c_int, [ctx, C.c_int, C.c_int, A, A, A]),
        "svh_refine_disp_2d_cost_patch_interpolation": (C.c_int, [ctx, C.c_int, A, A, A]),
        "svh_average_pooling_downsample": (C.c_int, [ctx, A, C.c_int, C.c_int, A]),
        "svh_on_demand_features": (C.c_int, [ctx, C.c_int, A, C.c_int, C.c_int, A]),
        "svh_on_demand_truncated_cost_volume": (C.c_int, [ctx, P(SvhOnDemandParams), A, A, A, C.c_int, A]),
        "svh_cacheless_patch_match": (C.c_int, [ctx, P(SvhOnDemandParams), A, A, C.c_int, C.c_int, C.c_uint64, A, P(i32)]),
        "svh_cacheless_patch_match_init": (C.c_int, [ctx, P(SvhOnDemandParams), A, A, C.c_int, C.c_int, C.c_uint64, A, A, P(i32)]),
        "svh_patch_match": (C.c_int, [ctx, P(SvhOnDemandParams), A, A, C.c_int, C.c_int, C.c_uint64, A, A, P(i32)]),
        "svh_unfold_compressed": (C.c_int, [ctx, A, P(i32), C.c_int, C.c_int, P(i32), A]),
        "svh_unfold_compressed_shape": (C.c_int, [A, P(i32), C.c_int, C.c_int, P(i32), P(C.c_int64)]),
        "svh_channels_mean": (C.c_int, [ctx, A, A]),
        "svh_channels_norm": (C.c_int, [ctx, A, A]),
        "svh_channels_zero_mean_norm": (C.c_int, [ctx, A, A, A]),
        "svh_zeromean_feature_volume": (C.c_int, [ctx, A, A, A]),
        "svh_normalized_feature_volume": (C.c_int, [ctx, A, A, A]),
        "svh_zeromean_normalized_feature_volume": (C.c_int, [ctx, A, A, A, A]),
        "svh_feature_volume_for_match_func": (C.c_int, [ctx, C.c_int, A, A]),
        "svh_guided_cost_volume": (C.c_int, [ctx, C.c_int, C.c_int, A, A, A, i32, A, A]),
        "svh_hierarchical_truncated_cost_volume": (C.c_int, [ctx, C.c_int, C.c_int, C.c_int, A, A, P(i32), P(i32), i32, i32, A, A]),
        "svh_census_shard_keys": (C.c_int, [ctx, P(SvhStereoParams), A, A, A]),
        "svh_census_shard_region1_is_global": (C.c_int, [P(SvhStereoParams), A, A]),
        "svh_census_shard_finish": (C.c_int, [ctx, P(SvhStereoParams), A, A, A, A, A]),
        "svh_census_band_match": (C.c_int, [ctx, P(SvhStereoParams), A, A, C.c_int32, C.c_int32, A]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
