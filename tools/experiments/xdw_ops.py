"""ctypes bindings of the round-4 E-elimination experiment (atomnas_amd/csrc/experimental/, NOT part of the product library):
the expand 1x1 + BatchNorm + activation recomputed on chip inside the depthwise kernels.  The experiment library is built by
tools/build_xdw_experiment.sh (atomnas_amd/csrc/build/variants/libxdw.so = the product library + these entry points) and loaded by
pointing ATOMNAS_HIP_LIB at it before atomnas_amd is imported; `available()` tells whether the loaded library has them."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from atomnas_amd import _lib  # noqa: E402
from atomnas_amd.ops import _ld, _p, _rows, _ss, _stream, _chk_cuda, dt_code, stat_rows_for  # noqa: E402

LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "atomnas_amd", "csrc", "build", "variants",
                        "libxdw.so")
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
_SIG = {
    "atomnas_gram_stats": [vp, i32, vp, vp, i32, i32, i32, vp, i32, vp],
    "atomnas_xdw_fwd": [vp, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i64, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "atomnas_xdw_bwd": [vp, i64, vp, i64, vp, vp, vp, vp, i32, i32, vp, i32, vp, vp, i32, vp, i32, vp, i64, vp, vp, i32, i32, vp, i32, i32,
                        i32, i32, i32, i32, vp],
    "atomnas_xdw_supported": [i32, i32, i32, i32, i32, i32, i32, i32],
}
_bound = [False]


def available():
    lib = _lib.load()
    if not hasattr(lib, "atomnas_xdw_fwd"):
        return False
    if not _bound[0]:
        for name, args in _SIG.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = i32
        _bound[0] = True
    return True


def call(name, *args):
    lib = _lib.load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise _lib.AtomnasHipError("%s failed (rc=%d): %s" % (name, rc, lib.atomnas_last_error().decode()))


def xdw_supported(N, H, W, inp, C, k, stride, dtype):
    """1 when the fused expand + depthwise kernels (csrc/xdw.hip) have instances for a branch segment of C hidden channels"""
    return bool(_lib.load().atomnas_xdw_supported(int(N), int(H), int(W), int(inp), int(C), int(k), int(stride), dt_code(dtype)))


def gram_stats(gram_m, sx, wexp, inp, C, stats, stat_ld):
    """statistics row [2][stat_ld] of the expand BatchNorm from the Gram matrix of the block input (include/atomnas_hip.h)"""
    call("atomnas_gram_stats", _p(gram_m), inp, _p(sx), _p(wexp), wexp.stride(0), inp, C, _p(stats), stat_ld, _stream())


def xdw_fwd(x, inp, wexp, in_scale, in_shift, act, w_taps, y, stats, stat_ld, N, H, W, C, k, stat_rows=None):
    """y = dwconv_k(act(in_scale * (x wexp^T) + in_shift)): expand + BN + activation on chip in front of the depthwise conv"""
    _chk_cuda(x, wexp, y, w_taps)
    if _lib.PROFILE is not None:
        _lib.profile_tag("N%d H%d C%d k%d s1 inp%d" % (N, H, C, k, inp))
    call("atomnas_xdw_fwd", _p(x), _ld(x), inp, _p(wexp), wexp.stride(0), _p(in_scale), _p(in_shift), int(act), _p(w_taps), w_taps.stride(0),
         _p(y), _ss(y), _p(stats), stat_ld, _rows(stats, stat_rows), N, H, W, C, k, dt_code(x.dtype), _stream())


def xdw_bwd(g, yraw, c1, c2, c3, x, inp, wexp, in_scale, in_shift, act, w_taps, h, dw, stats, stat_ld, N, H, W, C, k, stat_rows=None,
            dw_ws=None):
    """atomnas_dwconv_bwd with the expand output recomputed from the block input x (include/atomnas_hip.h)"""
    _chk_cuda(g, x, h, w_taps, wexp)
    rows = _rows(stats, stat_rows) if stats is not None else stat_rows_for(C)
    if dw is not None and dw_ws is None:
        dw_ws = torch.empty(rows * C * k * k, dtype=torch.float32, device=x.device)
    if _lib.PROFILE is not None:
        _lib.profile_tag("N%d H%d C%d k%d s1 inp%d" % (N, H, C, k, inp))
    call("atomnas_xdw_bwd", _p(g), _ss(g), _p(yraw), _ss(yraw), _p(c1), _p(c2), _p(c3), _p(x), _ld(x), inp, _p(wexp), wexp.stride(0),
         _p(in_scale), _p(in_shift), int(act), _p(w_taps), w_taps.stride(0), _p(h), _ss(h), _p(dw), _p(stats), stat_ld, rows, _p(dw_ws),
         N, H, W, C, k, dt_code(x.dtype), _stream())


