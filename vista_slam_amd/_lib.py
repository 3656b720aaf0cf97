"""ctypes binding of libsta_mi355.so (the C-ABI declared in include/sta_mi355.h).

There is NO fallback: if the HIP library is missing or fails to load, importing the product path
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)

Two builds of the same translation unit (vista_slam_amd/build.py): `load()` = libsta_mi355.so, the product, which exports
include/sta_mi355.h and nothing else; `load_test()` = libsta_mi355_test.so (-DSTA_TEST_HOOKS), which additionally exports the
kernel-level test / micro-benchmark entry points of include/sta_mi355_debug.h.  Only tests/ and tools/ load the second one
(`STAFrontend(..., lib=_lib.load_test())`, or `_lib.use_test_hooks()` at the top of a tool).
"""
import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "libsta_mi355.so")
TEST_LIB_PATH = os.path.join(PKG, "libsta_mi355_test.so")

STA_PREC_F16 = 1
STA_PREC_F16X3 = 3
STA_PREC_F16X3H = 5
STA_PREC_F16X3M = 6
PRECISIONS = {"f16": STA_PREC_F16, "f16x3": STA_PREC_F16X3, "f16x3h": STA_PREC_F16X3H, "f16x3m": STA_PREC_F16X3M}


class StaConfig(C.Structure):
    _fields_ = [("patch_size", C.c_int32), ("enc_embed_dim", C.c_int32), ("enc_depth", C.c_int32),
                ("enc_num_heads", C.c_int32), ("dec_embed_dim", C.c_int32), ("dec_depth", C.c_int32),
                ("dec_num_heads", C.c_int32), ("mlp_ratio", C.c_int32), ("rope_base", C.c_float),
                ("ln_eps", C.c_float), ("precision", C.c_int32)]


_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float
_fp = C.c_void_p   # device float* passed as integer address

# name -> (restype, argtypes); mirrors include/sta_mi355.h (the product ABI)
SIGNATURES = {
    "sta_default_config": (None, [C.POINTER(StaConfig)]),
    "sta_create": (_i, [C.POINTER(StaConfig), _i, C.POINTER(_vp)]),
    "sta_destroy": (_i, [_vp]),
    "sta_set_precision": (_i, [_vp, _i]),
    "sta_set_deterministic": (_i, [_vp, _i]),
    "sta_set_side_lanes": (_i, [_vp, _i]),
    "sta_pipeline_streams": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_i)]),
    "sta_reserve": (_i, [_vp, _i, _i, _i, _i, C.POINTER(_vp), _i]),
    "sta_alloc_stats": (_i, [_vp, C.POINTER(_i64)]),
    "sta_num_expected_tensors": (_i, [_vp]),
    "sta_num_loaded_tensors": (_i, [_vp]),
    "sta_load_tensor": (_i, [_vp, C.c_char_p, _vp, C.POINTER(_i64), _i, _i]),
    "sta_finalize_weights": (_i, [_vp]),
    "sta_encode": (_i, [_vp, _fp, _i, _i, _i, _fp, _vp]),
    "sta_decode": (_i, [_vp, _fp, _fp, _i, _i, _i, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "sta_decode_pos": (_i, [_vp, _fp, _fp, _vp, _vp, _i, _i, _i, C.POINTER(_vp), C.POINTER(_vp), _vp]),
    "sta_head_pose": (_i, [_vp, _fp, _i, _i64, _fp, _fp, _vp]),
    "sta_head_pts": (_i, [_vp, _fp, _i64, _fp, _i64, _fp, _i64, _fp, _i64, _i, _i, _i, _fp, _fp, _vp]),
    "sta_forward_pair": (_i, [_vp, _fp, _fp, _i, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                              C.POINTER(_vp), _vp]),
    "sta_encode_u8hwc": (_i, [_vp, _fp, _i, _i, _i, _fp, _vp]),
    "sta_encoder_norm": (_i, [_vp, _fp, C.c_int64, _fp, _vp]),
    "sta_forward_pair_u8hwc": (_i, [_vp, _fp, _fp, _i, _i, _i, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp),
                                    C.POINTER(_vp), _vp]),
    "sta_estimate_intrinsics": (_i, [_vp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _fp, _vp]),
    "sta_estimate_scale": (_i, [_vp, _fp, _fp, _fp, _fp, _i64, _fp, _vp]),
    "sta_preprocess_geometry": (_i, [_i, _i, _i, _i, _i, _i, C.POINTER(_i), C.POINTER(_i)]),
    "sta_preprocess_frame": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _vp]),
    "sta_world_pointcloud": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _f, _fp, _fp, _fp, C.POINTER(_i64), _vp]),
    "sta_mat_to_se3": (_i, [_vp, _fp, _i, _fp, _vp]),
    "sta_regress_views": (_i, [_vp, _fp, C.POINTER(_vp), _i, C.c_char_p, _f, _i, _i, _fp, C.POINTER(C.c_float),
                               C.POINTER(_i), C.POINTER(_i), _fp, _fp, _fp, _fp, _vp]),
    "sta_regress_views_begin": (_i, [_vp, _fp, C.POINTER(_vp), _i, _i, _i, _fp, _vp]),
    "sta_regress_views_finish": (_i, [_vp, C.c_char_p, _f, C.POINTER(C.c_float), C.POINTER(_i), C.POINTER(_i), _fp, _fp, _fp, _fp, _vp]),
    "sta_regress_views_abort": (_i, [_vp, _vp]),
    "sta_pack_compact": (_i, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i, _i, _i, _fp, _i64, _vp]),
    "sta_rope2d_inplace": (_i, [_fp, _i64, _i64, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "sta_rope2d_inplace_dtype": (_i, [_vp, _i, _i64, _i64, _vp, _i, _i, _i, _i, _f, _f, _vp]),
    "sta_flops_per_pair": (C.c_double, [_vp, _i, _i]),
    "sta_workspace_bytes": (_i64, [_vp]),
    "sta_weight_bytes": (_i64, [_vp]),
    "sta_enable_stage_timing": (_i, [_vp, _i]),
    "sta_get_stage_ms": (_i, [_vp, C.POINTER(_f)]),
    "sta_kernel_timing": (_i, [_vp, _i]),
    "sta_kernel_timing_read": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "sta_kernel_clock_read": (_i, [_vp, C.POINTER(C.c_float)]),
    "sta_kernel_timing_filter": (_i, [_vp, _i, _i, _i, _i, _i]),
    "sta_kernel_timing_dump_shapes": (_i, [_vp, _i, C.POINTER(_i), C.POINTER(C.c_float), C.POINTER(_i), C.POINTER(_i)]),
    "sta_range_report": (_i, [_vp, C.POINTER(C.c_ulonglong), _i]),
    "sta_last_error": (C.c_char_p, []),
    "sta_version": (C.c_char_p, []),
}

# include/sta_mi355_debug.h: kernel-level test / micro-benchmark entry points, exported by libsta_mi355_test.so only
TEST_SIGNATURES = {
    "sta_set_gemm_variant": (_i, [_vp, _i]),
    "sta_kernel_timing_dump": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(_i), C.POINTER(_i)]),
    "sta_kernel_stamps_dump": (_i, [_vp, _i, C.POINTER(C.c_double), C.POINTER(_i)]),
    "sta_bench_gemm_stamps": (_i, [_vp, _i, _i, _i, _i, C.POINTER(C.c_double), C.POINTER(C.c_ulonglong), _i, _vp]),
    "sta_bench_gemm": (_i, [_vp, _i, _i, _i, _i, _i, _i, C.POINTER(_f), _vp]),
    "sta_bench_gemm_last_ghz": (C.c_float, []),
    "sta_bench_attention": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, C.POINTER(_f), _vp]),
    "sta_debug_gemm": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _vp]),
    "sta_debug_qkv_rope": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _fp, _fp, _fp, _vp]),
    "sta_debug_attention": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _vp]),
    "sta_debug_attention_pose": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _vp]),
    "sta_debug_set_tail_hint": (_i, [_vp, _i]),
    "sta_debug_set_option": (_i, [_vp, _i, _i]),
    "sta_debug_pick_family": (_i, [_i, _i, C.c_longlong, _i, _i, _i, _i, _i, _i]),
    "sta_debug_conv3x3": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _i, _fp, _fp, _vp]),
    "sta_debug_convt": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _vp]),
    "sta_debug_up2": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _i, _fp, _vp]),
    "sta_debug_layernorm": (_i, [_vp, _fp, _fp, _fp, _i, _i, _f, _fp, _fp, _vp]),
    "sta_debug_head_final": (_i, [_vp, _fp, _fp, _fp, _i64, _fp, _fp, _vp]),
    "sta_debug_svd_orthogonalize": (_i, [_vp, _fp, _fp, _i, _vp]),
}

_lib = None
_test_lib = None


class StaError(RuntimeError):
    pass


_last_called = None      # the library object of the most recent C call: sta_last_error() is per library (two builds may be loaded)


def _bind(lib, table, strict):
    def note(result, _func, _args):
        global _last_called
        _last_called = lib
        return result
    for name, (res, args) in table.items():
        fn = getattr(lib, name, None) if not strict else getattr(lib, name)    # strict: AttributeError if the .so does not export it
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
            if name != "sta_last_error":
                fn.errcheck = note


def load_other(path):
    """A SECOND build of the library next to the product one (tools/ab_inproc.py: same-process A/B of two builds).  Binds
    the symbols that build exports; never used by the product path."""
    import torch  # noqa: F401
    lib = C.CDLL(path)
    _bind(lib, SIGNATURES, strict=False)
    _bind(lib, TEST_SIGNATURES, strict=False)
    return lib


def _missing(path):
    return StaError(f"{path} not found: the MI355X STA frontend has no CPU fallback. "
                    "Build it with `python -m vista_slam_amd.build` (needs hipcc) or `__graft_entry__.build()`.")


def load():
    """Load libsta_mi355.so (the product library) and bind every symbol of include/sta_mi355.h; raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise _missing(LIB_PATH)
    # torch must be imported BEFORE the library: the ROCm wheel bundles its own libamdhip64 and the
    # process must end up with exactly one HIP runtime (loading /opt/rocm's first makes the second
    # initialisation fail with "no ROCm-capable device is detected").
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    _bind(lib, SIGNATURES, strict=True)
    _lib = lib
    return lib


def load_test():
    """Load libsta_mi355_test.so: the product ABI plus the kernel-level test / micro-benchmark entry points
    (include/sta_mi355_debug.h).  tests/ and tools/ only."""
    global _test_lib
    if _test_lib is not None:
        return _test_lib
    if not os.path.exists(TEST_LIB_PATH):
        raise _missing(TEST_LIB_PATH)
    import torch  # noqa: F401
    lib = C.CDLL(TEST_LIB_PATH)
    _bind(lib, SIGNATURES, strict=True)
    _bind(lib, TEST_SIGNATURES, strict=True)
    _test_lib = lib
    return lib


def use_test_hooks():
    """tools/: make the test-hooks build THIS process's default library, so that every STAFrontend() created afterwards has the
    sta_debug_* / sta_bench_* entry points.  Never called by the product path."""
    global _lib
    _lib = load_test()
    return _lib


def check(rc):
    if rc != 0:
        lib = _last_called or _lib or _test_lib or load()
        msg = lib.sta_last_error()
        raise StaError((msg or b"unknown error").decode())
