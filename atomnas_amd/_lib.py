"""ctypes binding of libatomnas_hip.so (C ABI in include/atomnas_hip.h).

The product path has no CPU fallback: if the shared object is missing or a call fails, an exception is raised.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ATOMNAS_HIP_LIB") or os.path.join(_HERE, "libatomnas_hip.so")  # override: experiment builds

vp, i32, i64, f32, f64, u64 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double,
                               ctypes.c_ulonglong)

# name -> argument ctypes (all return int status)
SIGNATURES = {
    "atomnas_dwconv_fwd": [vp, i32, i64, vp, vp, i32, vp, i32, vp, i32, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "atomnas_dwconv_bwd": [vp, i32, i64, vp, i32, i64, vp, vp, vp, vp, i32, i64, vp, vp, i32, vp, i32, vp, i32, i64, vp, vp, i32, i32,
                           vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "atomnas_pw_gemm_nt": [i32, vp, i32, i64, vp, i32, i64, vp, vp, vp, i32, vp, i32, vp, i32, i64, i32, vp, i32, vp, i32, i64, vp, vp,
                           i32, vp, vp, i32, i32, i64, i32, i32, i32, vp],
    "atomnas_pw_gemm_tn": [i32, vp, i32, i64, vp, i32, i64, vp, vp, vp, i32, i32, i32, vp, i32, i64, vp, i32, i64, vp, vp, vp, i32, i32,
                           vp, i64, i64, i64, vp, i64, i32, vp],
    "atomnas_expand_bwd": [vp, i32, i64, vp, vp, i32, vp, i32, vp, i32, vp, i32, vp, vp, i64, vp, i32, vp, i64, i32, i32, i32, vp],
    "atomnas_project_bwd": [vp, i32, vp, i32, vp, i32, i64, vp, vp, i32, vp, i32, i64, vp, i32, vp, i64, i64, vp, i64, i64, i32, i32, i32, vp],
    "atomnas_bn_finalize_fwd": [vp, i32, i32, f64, vp, vp, f32, f32, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp],
    "atomnas_bn_eval_coeffs": [vp, vp, vp, vp, f32, vp, vp, i32, vp, vp],
    "atomnas_bn_finalize_bwd": [vp, i32, i32, f64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp],
    "atomnas_bn_apply": [vp, i32, vp, vp, i32, vp, i32, vp, i32, i64, i32, i32, vp],
    "atomnas_bnbwd_apply": [vp, i32, vp, i32, vp, vp, vp, vp, i32, i64, i32, i32, vp],
    "atomnas_bn_act_pool": [vp, i32, vp, vp, i32, vp, i32, vp, f32, u64, vp, i32, i32, i32, i32, vp],
    "atomnas_pool_act_bwd": [vp, i32, vp, f32, vp, i32, vp, vp, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp],
    "atomnas_act_bwd_stats": [vp, i32, vp, i32, vp, vp, i32, vp, i32, vp, i32, i64, i32, i32, vp],
    "atomnas_se_squeeze": [vp, i32, i64, vp, vp, i32, vp, i32, i32, i64, i32, i32, i32, i32, vp],
    "atomnas_se_mlp_fwd": [vp, i32, i32, i64, vp, vp, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, vp],
    "atomnas_se_scale": [vp, i32, i64, vp, vp, i32, vp, i32, vp, i32, i64, i64, i32, i32, i32, vp],
    "atomnas_se_bwd_gate": [vp, i32, i64, vp, i32, i64, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, i32, i32,
                            i32, i32, i32, i32, i32, vp],
    "atomnas_se_bwd_apply": [vp, i32, i64, vp, i32, i64, vp, vp, i32, vp, vp, i32, vp, i32, i64, vp, i32, i64, i32, i32, i32, vp],
    "atomnas_im2col_stem": [vp, vp, i32, i32, i32, i32, i32, vp],
    "atomnas_ce_smooth": [vp, i32, vp, f32, i32, i32, vp, vp, i32, f32, vp, i32, vp],
    "atomnas_colsum": [vp, i32, vp, i64, i32, i32, vp],
    "atomnas_fused_rmsprop_ema": [vp, vp, vp, vp, vp, vp, i64, vp, f64, f64, i32, f64, vp, vp, vp],
    "atomnas_vec_sum": [vp, i32, f32, vp, vp],
    "atomnas_ema_update": [vp, vp, i64, vp, vp],
    "atomnas_scale_by": [vp, i64, vp, i32, vp],
    "atomnas_zero": [vp, i64, vp],
    "atomnas_add_i64": [vp, i64, i64, vp],
    "atomnas_reg_grad": [vp, vp, vp, i32, i32, vp, vp, vp],
    "atomnas_reg_value": [vp, vp, i32, i32, vp, f32, vp, vp, vp],
    "atomnas_pack_weights": [vp, vp, vp, i32, i32, vp],
    "atomnas_reduce_defer": [i32, vp],
    "atomnas_reduce_flush": [vp],
    "atomnas_gamma_mask": [vp, vp, vp, i32, f32, i32, vp, vp, vp, vp],
    "atomnas_mask_index": [vp, i32, vp, vp, vp],
    "atomnas_gather_dim": [vp, vp, vp, i64, i64, i64, i64, i32, i32, i32, vp],
    "atomnas_gather_jobs": [vp, i32, i64, vp],
    "atomnas_gram": [vp, i32, i64, i32, vp, i64, vp, vp, i32, vp],
    "atomnas_image_preprocess": [vp, vp, i32, i32, vp, vp, vp, i32, i32, vp],
    "atomnas_xb_coeffs": [vp, vp, vp, i32, vp, i32, vp, i32, i32, vp, i32, vp, vp, vp],
    "atomnas_fold_jobs": [vp, i32, i32, i64, i64, vp],
}
NO_STATUS = {"atomnas_last_error": (ctypes.c_char_p, []), "atomnas_abi_version": (i32, []),
             "atomnas_runtime_version": (i32, []), "atomnas_expand_bwd_supported": (i32, [i32, i32, i32]),
             "atomnas_project_bwd_supported": (i32, [i32, i32, i32]),
             "atomnas_project_bwd_dp_supported": (i32, [i64, i32, i32, i32, i32, i64, i32, i64, i32, i32]),
             "atomnas_dwconv_cw_supported": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
             "atomnas_dwconv_mm_supported": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
             "atomnas_se_pool_parts": (i32, [i32, i32, i32])}

ABI_VERSION = 9   # include/atomnas_hip.h ATOMNAS_ABI_VERSION
_lib = None


class AtomnasHipError(RuntimeError):
    pass


def exported_symbols():
    return sorted(list(SIGNATURES) + list(NO_STATUS))


def load():
    """Loads the library (once).  torch must already be imported so that its HIP runtime is the one in the process."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- must come first: the library then binds to the HIP runtime torch ships (same SONAME)
    if not os.path.exists(LIB_PATH):
        raise AtomnasHipError(
            "libatomnas_hip.so is missing (%s). Build it with `python -m atomnas_amd.build`; there is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    for name, (res, args) in NO_STATUS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    if lib.atomnas_abi_version() != ABI_VERSION:
        raise AtomnasHipError("%s has ABI version %d, this package binds version %d: rebuild with `python -m atomnas_amd.build`"
                              % (LIB_PATH, lib.atomnas_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


# Optional per-launch timing (bench.py's roofline leg): when PROFILE is a list, every call is bracketed by events on the
# launch stream and (name, tag, start_event, end_event) is appended.  `tag` is set by the caller through profile_tag().
PROFILE = None
_TAG = [None]


def profile_tag(tag):
    _TAG[0] = tag


def call(name, *args):
    lib = load()
    if PROFILE is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib, name)(*args)
        e1.record()
        PROFILE.append((name, _TAG[0], e0, e1))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise AtomnasHipError("%s failed (rc=%d): %s" % (name, rc, lib.atomnas_last_error().decode()))
