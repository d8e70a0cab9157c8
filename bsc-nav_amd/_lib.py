"""ctypes binding of libbscnav.so (include/bscnav.h).

The shared library is the product; this module only loads it and declares the
signatures.  There is no fallback of any kind: a missing library or a missing
GPU raises immediately.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BSC_LIB_PATH: an A/B build of the same sources (csrc/build.sh with BSC_OUT / BSC_EXTRA_FLAGS); never a different implementation
LIB_PATH = os.environ.get("BSC_LIB_PATH") or os.path.join(_HERE, "libbscnav.so")

BSC_MODE_EXACT, BSC_MODE_MEAN, BSC_MODE_MAX = 0, 1, 2
MODES = {"exact": BSC_MODE_EXACT, "mean": BSC_MODE_MEAN, "max": BSC_MODE_MAX}


class BscConfig(C.Structure):
    _fields_ = [
        ("height", C.c_int32), ("width", C.c_int32), ("grid_size", C.c_int32), ("min_h", C.c_int32),
        ("max_h", C.c_int32), ("patch_grid", C.c_int32), ("token_dim", C.c_int32), ("iter_size", C.c_int32),
        ("cache_size", C.c_int32), ("mode", C.c_int32), ("voxel_capacity", C.c_int32), ("max_points", C.c_int32),
        ("token_capacity", C.c_int64), ("cell_size", C.c_double), ("min_depth", C.c_double),
        ("max_depth", C.c_double), ("K", C.c_double * 9), ("Kinv", C.c_double * 9), ("Kpatch", C.c_double * 9),
    ]


DRAW_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32))

# every symbol include/bscnav.h declares: name -> (restype, argtypes)
_VP, _I32, _I64, _F64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
SIGNATURES = {
    "bsc_last_error": (C.c_char_p, []),
    "bsc_version": (C.c_char_p, []),
    "bsc_create": (_I32, [C.POINTER(BscConfig), _I32, _VP, C.POINTER(_VP)]),
    "bsc_destroy": (None, [_VP]),
    "bsc_reset": (_I32, [_VP]),
    "bsc_ingest": (_I32, [_VP, _I32, _VP, _VP, _I32, _VP, _VP, _VP, _VP, _VP, DRAW_FN, _VP]),
    "bsc_ingest_typed": (_I32, [_VP, _I32, _VP, _VP, _I32, _VP, _I32, _VP, _VP, _VP, _VP, DRAW_FN, _VP]),
    "bsc_flush": (_I32, [_VP, DRAW_FN, _VP]),
    "bsc_counters": (_I32, [_VP, _VP]),
    "bsc_geometry": (_I32, [_VP, _VP, _VP, _VP, _I64] + [_VP] * 8),
    "bsc_sort_pairs_u32": (_I32, [_VP, _VP, _VP, _I64, _I32, _I32, _VP, _VP]),
    "bsc_export_rgb": (_I32, [_VP, _VP, _VP, _VP]),
    "bsc_export_occupied": (_I32, [_VP, _VP]),
    "bsc_export_heightmap": (_I32, [_VP, _VP, _VP]),
    "bsc_export_cache": (_I32, [_VP, _VP, _VP, _VP]),
    "bsc_export_store": (_I32, [_VP, _VP, _VP, _VP, _VP]),
    "bsc_export_dense": (_I32, [_VP, _VP, _VP]),
    "bsc_import_rgb": (_I32, [_VP, _I64, _VP, _VP, _VP]),
    "bsc_import_store": (_I32, [_VP, _I64, _I64, _VP, _VP, _VP, _VP]),
    "bsc_import_dense": (_I32, [_VP, _I64, _VP, _VP]),
    "bsc_pool_query": (_I32, [_VP, _VP, _I32, _I32, _I32, _VP]),
    "bsc_localize": (_I32, [_VP, _VP, _I32, _I32, _F64, _VP, _I32, _I32, _VP, _VP, _VP]),
    "bsc_cluster_centers": (_I32, [_VP, _I32, _I32, _VP, _VP, _F64, _I32, _VP, _VP, _VP, _VP]),
    "bsc_frontier_mask": (_I32, [_VP, _VP, _VP]),
    "bsc_frontier_clusters": (_I32, [_VP, _VP, _I32, _I32, _I32, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "bsc_import_cv_map": (_I32, [_VP, _VP]),
    "bsc_dense_gather": (_I32, [_VP, _I64, _VP, _VP, _VP]),
    "bsc_dense_replace": (_I32, [_VP, _I64, _VP, _VP, _VP]),
    "bsc_dense_gather_rgb": (_I32, [_VP, _I64, _VP, _VP, _VP]),
    "bsc_dense_replace_full": (_I32, [_VP, _I64, _VP, _VP, _VP, _VP, _VP]),
    "bsc_import_heightmap": (_I32, [_VP, _VP, _VP]),
    "bsc_keys_dev": (_I32, [_VP, C.POINTER(_VP), C.POINTER(_I64)]),
    "bsc_point_log_enable": (_I32, [_VP, _I64]),
    "bsc_point_log_read": (_I32, [_VP, _VP, _VP, _I64, C.POINTER(_I64)]),
    "bsc_replay_colour": (_I32, [_I64, _VP, _VP, _I64, _VP, _VP, _VP]),
    "bsc_kernel_stats": (_I32, [_VP, _I32, _I32, _VP]),
    "bsc_sync": (_I32, [_VP]),
    "bsc_stream_wait_chain": (_I32, [_VP, _VP]),
    "bsc_enc_embed_layernorm": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, C.c_float, _VP]),
    "bsc_enc_final_layernorm": (_I32, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, C.c_float, _VP]),
    "bsc_enc_bias_layernorm": (_I32, [_VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, _I32, C.c_float, _VP]),
    "bsc_enc_attention": (_I32, [_VP, _I32, _I32, _I32, _I32, _VP, _VP]),
    "bsc_enc_attention_dyn": (_I32, [_VP, _I32, _I32, _I32, _I32, _VP, _VP, _VP]),
    "bsc_enc_preprocess_patches": (_I32, [_VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _VP, _VP, _VP]),
    "bsc_enc_preprocess_patches_typed": (_I32, [_VP, _I32, _I32, _I32, _I32, _I32, _I32, _VP, _I32, _VP, _VP, _VP]),
    "bsc_host_choice_draws": (_I32, [_VP, _VP, C.c_uint32, C.c_uint32, _VP]),
    "bsc_host_shuffled_sample": (_I32, [_VP, _VP, _I64, _I32, _VP, _VP]),
    "bsc_enc_split_weights": (_I32, [_VP, _I32, _I32, C.c_float, _VP, _VP]),
    "bsc_enc_gemm_split": (_I32, [_VP, _I64, _I32, _VP, _I32, _VP, _VP, _VP, C.c_float, C.c_float, _I32, _I32, C.c_float, _VP]),
    "bsc_enc_gemm_split_ln": (_I32, [_VP, _I64, _I32, _VP, _I32, _VP, _VP, _VP, C.c_float, C.c_float, _I32, _I32, C.c_float, _VP, _VP,
                                     C.c_float, _VP]),
    "bsc_enc_gemm_split_ws": (_I32, [_VP, _I64, _I32, _VP, _I32, _VP, _VP, _VP, C.c_float, C.c_float, _I32, _I32, C.c_float, _VP, _VP,
                                     C.c_float, _VP, _I64, _VP]),
    "bsc_enc_embed_layernorm_f32": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I32, _I32, _I32, _I32, C.c_float, _VP, _VP, _VP, _VP, _VP]),
    "bsc_enc_final_layernorm_f32": (_I32, [_VP, _VP, _VP, _I32, _I32, _I32, _I32, C.c_float, _VP, _VP]),
    "bsc_enc_layernorm_split": (_I32, [_VP, _VP, _VP, _I64, _I32, C.c_float, C.c_float, _VP, _VP]),
    "bsc_enc_split_rows": (_I32, [_VP, _I64, _I32, C.c_float, _VP, _VP]),
    "bsc_enc_attention_split": (_I32, [_VP, _I32, _I32, _I32, _I32, _VP, C.c_float, _VP, _VP]),
    "bsc_enc_add_layernorm": (_I32, [_VP, _VP, _VP, _VP, _VP, _VP, _I64, _I32, C.c_float, _VP]),
}

_lib = None


def load():
    """Load libbscnav.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(bsc-nav_amd/csrc/build.sh).  bsc_nav_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class BscError(RuntimeError):
    pass


def check(status):
    if status != 0:
        msg = load().bsc_last_error().decode("utf-8", "replace")
        raise BscError(f"libbscnav status {status}: {msg}")
