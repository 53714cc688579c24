"""Thin tensor-level wrappers over the C ABI (pointers + shapes out of torch tensors, current torch stream).

Activations are 2-D tensors [M, ld] (NHWC flattened, ld % 8 == 0) of dtype float32 or bfloat16.
Nothing here computes on the host; a missing library or a failed launch raises.
"""
import ctypes

import os

import torch

from . import _lib
from ._lib import call

PRO_NONE, PRO_BNRELU, PRO_BNBWD = 0, 1, 2
STAT_NONE, STAT_SQ, STAT_Z = 0, 1, 2
HYP_LR, HYP_RHO, HYP_EMA_DECAY, HYP_GRAD_SCALE = 0, 1, 2, 3


# Launch recorder (tools/make_bench_shapes.py -> tests/golden/bench_shapes.json -> tests/test_bench_shapes_gpu.py): when RECORD is a
# list every wrapper below appends a dict that names its entry point and everything that selects a kernel instance or a launch geometry
# (sizes, prologue / epilogue / statistics modes, layouts, pitches, workspace sizes) -- no pointers, no data.
RECORD = None


def _lay(t):
    """layout descriptor of an activation argument: None (absent), "slab" or the row pitch of a plain tensor"""
    if t is None:
        return None
    return "slab" if isinstance(t, Slab) else int(t.stride(0))


def _rec(entry, **kw):
    if RECORD is not None:
        kw["entry"] = entry
        RECORD.append(kw)


def stat_rows_for(c):
    """Partial rows of a statistics buffer [rows][2][c] (include/atomnas_hip.h): every producing workgroup owns one row, so
    more rows allow more concurrent workgroups; few-channel tensors (one or two channel slabs) need the most."""
    if c <= 64:
        return 1024
    if c < 1024:
        return 512
    return 128


def _rows(stats, stat_rows):
    if stats is None:
        return 0
    if stat_rows is not None:
        return int(stat_rows)
    if stats.dim() != 3:
        raise ValueError("pass stat_rows with a flat statistics buffer")
    return stats.shape[0]


# cap of the partial-output scratch of the weight-gradient GEMM (MiB).  The scratch bounds the number of row chunks = workgroups per
# output tile: 64 MiB measured 27.62 / 27.78 ms per step against 27.84 / 28.01 at 32 (two boxes, round 5), 16: 28.46, 128 / 256: as 64
_TN_WS_FLOATS = int(os.environ.get("ATOMNAS_TN_WS_MB", "64")) << 18


# ---- deferred fixed-order reductions (csrc/reduce.hip): while on, the weight-gradient kernels leave their per-workgroup partials in
# their workspaces and only RECORD the reduction; reduce_flush() then sums all recorded jobs with one launch per 56 jobs.  The
# workspaces are kept alive here until the flush.  Flush before anything reads the gradient arena.
_DEFER = [False]
_DEFER_KEEP = []
_DEFER_BYTES = [0]
_DEFER_LIMIT = 512 << 20   # partial workspaces kept alive before an early flush: bounds the extra peak memory of the deferral


# ---- fold jobs of the fused block (csrc/reduce.hip k_fold_jobs): a layer's weight gradient is accumulated in a padded scratch matrix by ONE
# weight-gradient GEMM and folded into the contiguous gradient tensor afterwards.  Inside a deferred-reduction window the folds of all
# blocks since the last flush run as ONE launch behind the flush (the GEMM's partials reach the scratch only there); otherwise at once.
_FOLD_PENDING = []   # (manager, first job, number of jobs)


def fold_jobs(mgr, first, n):
    """dst += src; src = 0 for jobs first .. first + n - 1 of mgr.fold_table (runtime.ArenaManager)"""
    if n <= 0:
        return
    if _DEFER[0]:
        if _FOLD_PENDING and _FOLD_PENDING[-1][0] is mgr and _FOLD_PENDING[-1][1] == first + n:   # backward walks the table downwards
            _FOLD_PENDING[-1] = (mgr, first, _FOLD_PENDING[-1][2] + n)
        elif _FOLD_PENDING and _FOLD_PENDING[-1][0] is mgr and _FOLD_PENDING[-1][1] + _FOLD_PENDING[-1][2] == first:
            _FOLD_PENDING[-1] = (mgr, _FOLD_PENDING[-1][1], _FOLD_PENDING[-1][2] + n)
        else:
            _FOLD_PENDING.append((mgr, first, n))
        return
    _fold_launch(mgr, first, n)


def _fold_launch(mgr, first, n):
    blk = mgr.fold_blk0
    call("atomnas_fold_jobs", _p(mgr.fold_table), int(first), int(n), int(blk[first]), int(blk[first + n] - blk[first]), _stream())


def _fold_flush():
    for mgr, first, n in _FOLD_PENDING:
        _fold_launch(mgr, first, n)
    del _FOLD_PENDING[:]


def reduce_defer(on):
    call("atomnas_reduce_defer", int(bool(on)), _stream())
    if not on:
        _fold_flush()   # atomnas_reduce_defer(0) has flushed the recorded reductions on this stream
    _DEFER[0] = bool(on)
    if not on:
        del _DEFER_KEEP[:]
    _DEFER_BYTES[0] = 0


def reduce_flush():
    call("atomnas_reduce_flush", _stream())
    _fold_flush()
    del _DEFER_KEEP[:]
    _DEFER_BYTES[0] = 0


def _keep(ws):
    """called BEFORE the launch that writes partials into ws: when the kept workspaces pass the limit the jobs recorded so far are
    flushed first (one more reduce launch; the sums per job are the same whenever they run), then ws starts the next batch"""
    if _DEFER[0] and ws is not None:
        nbytes = ws.numel() * ws.element_size()
        if _DEFER_KEEP and _DEFER_BYTES[0] + nbytes > _DEFER_LIMIT:
            reduce_flush()
        _DEFER_KEEP.append(ws)
        _DEFER_BYTES[0] += nbytes
    return ws


def tn_workspace(nu, nv, dev):
    """scratch for the per-row-chunk partial outputs of atomnas_pw_gemm_tn (at most 64 MiB)"""
    return torch.empty(max(2 * nu * nv, min(256 * nu * nv, _TN_WS_FLOATS)), dtype=torch.float32, device=dev)



class Slab:
    """Slab-major activation (include/atomnas_hip.h): C channels in slabs of 16, every slab a contiguous [M][16] matrix.
    The 6x-expanded hidden tensors of a block live in this layout: a kernel workgroup that owns a channel range then streams
    contiguous memory.  `seg(o)` is the same tensor seen from channel o on (o a multiple of 16: a branch segment)."""
    __slots__ = ("t", "M", "C", "ss", "off")

    def __init__(self, M, C, dtype, device, zero=False, _t=None, _off=0):
        self.M, self.C = int(M), int(C)
        self.ss = self.M * 16
        n = (self.C + 15) // 16 * self.ss
        self.t = _t if _t is not None else (torch.zeros if zero else torch.empty)(n, dtype=dtype, device=device)
        self.off = _off

    def seg(self, o):
        if o == 0:
            return self
        if o % 16:
            raise ValueError("slab segments start at multiples of 16 channels")
        return Slab(self.M, self.C - o, self.t.dtype, self.t.device, _t=self.t, _off=self.off + (o // 16) * self.ss)

    @property
    def dtype(self):
        return self.t.dtype

    @property
    def device(self):
        return self.t.device

    @property
    def is_cuda(self):
        return self.t.is_cuda

    def data_ptr(self):
        return self.t.data_ptr() + self.off * self.t.element_size()

    def to_plain(self):
        """[M, C16] torch tensor with the same contents (tests / debugging)"""
        ns = (self.C + 15) // 16
        v = self.t[self.off:self.off + ns * self.ss].view(ns, self.M, 16)
        return v.permute(1, 0, 2).reshape(self.M, ns * 16)

    @staticmethod
    def from_plain(x2d, C=None):
        M, ld = x2d.shape
        C = ld if C is None else C
        s = Slab(M, C, x2d.dtype, x2d.device, zero=True)
        ns = (C + 15) // 16
        buf = torch.zeros(M, ns * 16, dtype=x2d.dtype, device=x2d.device)
        buf[:, :min(ld, ns * 16)] = x2d[:, :min(ld, ns * 16)]
        s.t.view(ns, M, 16).copy_(buf.view(M, ns, 16).permute(1, 0, 2))
        return s


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def _ss(t):
    return t.ss if isinstance(t, Slab) else 0


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dt_code(dtype):
    if dtype == torch.float32:
        return 0
    if dtype == torch.bfloat16:
        return 1
    raise TypeError("activations must be float32 or bfloat16, got %s" % dtype)


def _ld(t):
    if isinstance(t, Slab):
        return 16
    assert t.dim() == 2 and t.stride(1) == 1, "activation must be [M, ld] with unit channel stride"
    return t.stride(0)


def pad8(c):
    return (c + 7) // 8 * 8


def _chk_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.AtomnasHipError("atomnas_amd kernels run on the GPU only (got a %s tensor)" % t.device)


def dwconv_fwd(x, in_scale, in_shift, in_relu, w_taps, y, stats, stat_ld, N, H, W, C, k, stride, stat_rows=None):
    _chk_cuda(x, y, w_taps)
    if _lib.PROFILE is not None:
        _lib.profile_tag("N%d H%d C%d k%d s%d" % (N, H, C, k, stride))
    _rec("dwconv_fwd", N=N, H=H, W=W, C=C, k=k, stride=stride, x=_lay(x), y=_lay(y), fused_in=in_scale is not None, act=int(in_relu),
         ldw=int(w_taps.stride(0)), stats=stats is not None, stat_ld=int(stat_ld), stat_rows=_rows(stats, stat_rows), dt=dt_code(x.dtype))
    call("atomnas_dwconv_fwd", _p(x), _ld(x), _ss(x), _p(in_scale), _p(in_shift), int(in_relu), _p(w_taps), w_taps.stride(0), _p(y),
         _ld(y), _ss(y), _p(stats), stat_ld, _rows(stats, stat_rows), N, H, W, C, k, stride, dt_code(x.dtype), _stream())


def dwconv_bwd(g, yraw, c1, c2, c3, x, in_scale, in_shift, in_relu, w_taps, h, dw, stats, stat_ld, N, H, W, C, k, stride,
               stat_rows=None, dw_ws=None):
    """dw_ws: scratch [part_rows][C][k*k] for the per-workgroup weight-gradient partials (allocated here when omitted)"""
    _chk_cuda(g, x, h, w_taps)
    rows = _rows(stats, stat_rows) if stats is not None else stat_rows_for(C)
    if dw is not None and dw_ws is None:
        dw_ws = torch.empty(rows * C * k * k, dtype=torch.float32, device=x.device)
    _keep(dw_ws)
    if _lib.PROFILE is not None:
        _lib.profile_tag("N%d H%d C%d k%d s%d" % (N, H, C, k, stride))
    _rec("dwconv_bwd", N=N, H=H, W=W, C=C, k=k, stride=stride, g=_lay(g), yraw=_lay(yraw), x=_lay(x), h=_lay(h), fused_in=in_scale is not None,
         act=int(in_relu), ldw=int(w_taps.stride(0)), dw=dw is not None, stats=stats is not None, stat_ld=int(stat_ld), part_rows=int(rows),
         dt=dt_code(x.dtype))
    call("atomnas_dwconv_bwd", _p(g), _ld(g), _ss(g), _p(yraw), _ld(yraw) if yraw is not None else 0, _ss(yraw), _p(c1), _p(c2), _p(c3),
         _p(x), _ld(x), _ss(x), _p(in_scale), _p(in_shift), int(in_relu), _p(w_taps), w_taps.stride(0), _p(h), _ld(h), _ss(h), _p(dw),
         _p(stats), stat_ld, rows,
         _p(dw_ws), N, H, W, C, k, stride, dt_code(x.dtype), _stream())


def gram(x, M, inp, gram_out, sx_out, ws=None):
    """G = X^T X [inp][inp] and column sums sx [inp] of the block input x [M, ld] (include/atomnas_hip.h)"""
    _chk_cuda(x, gram_out, sx_out)
    if ws is None:
        ws = torch.empty(2048 * (inp * inp + inp), dtype=torch.float32, device=x.device)
    if _lib.PROFILE is not None:
        _lib.profile_tag("M%d inp%d" % (M, inp))
    _rec("gram", M=int(M), inp=int(inp), x=_lay(x), ws_floats=int(ws.numel()), dt=dt_code(x.dtype))
    call("atomnas_gram", _p(x), _ld(x), M, inp, _p(ws), ws.numel(), _p(gram_out), _p(sx_out), dt_code(x.dtype), _stream())


def xb_coeffs(c2, c3, wexp, gram, sx, inp, C, mp, vb, dwe):
    """inp x inp sized corrections of the expand backward without E (include/atomnas_hip.h)"""
    _rec("xb_coeffs", inp=int(inp), C=int(C), ldwe=int(wexp.stride(0)), ldm=int(mp.stride(0)))
    call("atomnas_xb_coeffs", _p(c2), _p(c3), _p(wexp), wexp.stride(0), _p(gram), inp, _p(sx), inp, C, _p(mp), mp.stride(0), _p(vb), _p(dwe),
         _stream())


def gemm_nt(a, wp, c, M, N, K, a_mode=PRO_NONE, a2=None, ac1=None, ac2=None, ac3=None, a_relu=False, add=None, z=None,
            zscale=None, zshift=None, mask=False, bias=None, stats=None, stat_mode=STAT_NONE, stat_rows=None):
    _chk_cuda(a, wp, c)
    if _lib.PROFILE is not None:
        _lib.profile_tag("M%d N%d K%d pro%d st%d%s%s" % (M, N, K, a_mode, stat_mode, "+add" if add is not None else "", "+mask" if mask else ""))
    out_f32 = 1 if (c.dtype == torch.float32 and a.dtype != torch.float32) else 0
    _rec("pw_gemm_nt", M=int(M), N=int(N), K=int(K), a_mode=int(a_mode), a=_lay(a), a2=_lay(a2), a_relu=int(a_relu), ldw=int(wp.stride(0)), c=_lay(c),
         out_f32=out_f32, add=_lay(add), z=_lay(z), mask=int(mask), bias=bias is not None, stat_mode=int(stat_mode) if stats is not None else 0,
         stat_rows=_rows(stats, stat_rows), dt=dt_code(a.dtype))
    call("atomnas_pw_gemm_nt", a_mode, _p(a), _ld(a), _ss(a), _p(a2), _ld(a2) if a2 is not None else 0, _ss(a2), _p(ac1), _p(ac2), _p(ac3),
         int(a_relu), _p(wp), wp.stride(0), _p(c), _ld(c), _ss(c), out_f32, _p(add), _ld(add) if add is not None else 0, _p(z),
         _ld(z) if z is not None else 0, _ss(z), _p(zscale), _p(zshift), int(mask), _p(bias), _p(stats), stat_mode, _rows(stats, stat_rows),
         M, N, K, dt_code(a.dtype), _stream())


def gemm_tn(u, NU, v, NV, out, si, sj, M, u_mode=PRO_NONE, u2=None, uc1=None, uc2=None, uc3=None, u_relu=False, v_mode=PRO_NONE,
            v2=None, vc1=None, vc2=None, vc3=None, v_relu=False, ws=None):
    """ws: scratch for the per-row-chunk partial outputs (allocated here when omitted; pass False for a single-chunk reduction)"""
    _chk_cuda(u, v, out)
    if ws is None:
        ws = tn_workspace(NU, NV, u.device)
    elif ws is False:
        ws = None
    _keep(ws)
    if _lib.PROFILE is not None:
        _lib.profile_tag("M%d NU%d NV%d pro%d,%d" % (M, NU, NV, u_mode, v_mode))
    _rec("pw_gemm_tn", M=int(M), NU=int(NU), NV=int(NV), u_mode=int(u_mode), u=_lay(u), u2=_lay(u2), u_relu=int(u_relu), v_mode=int(v_mode), v=_lay(v),
         v2=_lay(v2), v_relu=int(v_relu), si=int(si), sj=int(sj), ws_floats=int(ws.numel()) if ws is not None else 0, dt=dt_code(u.dtype))
    call("atomnas_pw_gemm_tn", u_mode, _p(u), _ld(u), _ss(u), _p(u2), _ld(u2) if u2 is not None else 0, _ss(u2), _p(uc1), _p(uc2), _p(uc3),
         int(u_relu), NU, v_mode, _p(v), _ld(v), _ss(v), _p(v2), _ld(v2) if v2 is not None else 0, _ss(v2), _p(vc1), _p(vc2), _p(vc3), int(v_relu),
         NV, _p(out), si, sj, M, _p(ws), ws.numel() if ws is not None else 0, dt_code(u.dtype), _stream())


def expand_bwd_supported(inp, hid, dtype):
    return bool(_lib.load().atomnas_expand_bwd_supported(int(inp), int(hid), dt_code(dtype)))


def expand_bwd_workspace(inp, hid, dev):
    """scratch for the per-workgroup partials of the expand weight gradient: one per resident workgroup (at most 1024), 64 MiB cap"""
    return torch.empty(min(1024 * inp * hid, 16 << 20), dtype=torch.float32, device=dev)


def expand_bwd(h, c1, x, wt_pack, add, gx, dwe, M, inp, hid, ws=None, mp=None, vb=None):
    """Fused backward of the expand convolution without its raw output E (include/atomnas_hip.h): gx = (c1*h) * We (+ add), dwe += (c1*h)^T x,
    and with mp / vb (atomnas_xb_coeffs) gx += x M + v."""
    _chk_cuda(h, x, gx, dwe, wt_pack)
    wt, ldw = wt_pack, wt_pack.stride(0)
    if ws is None:
        ws = expand_bwd_workspace(inp, hid, x.device)
    _keep(ws)
    if _lib.PROFILE is not None:
        _lib.profile_tag("M%d N%d K%d fusedbwd" % (M, inp, hid))
    _rec("expand_bwd", M=int(M), inp=int(inp), hid=int(hid), h=_lay(h), x=_lay(x), ldw=int(ldw), add=_lay(add), gx=_lay(gx), ws_floats=int(ws.numel()),
         mp=mp is not None, ldm=int(mp.stride(0)) if mp is not None else 0, vb=vb is not None, dt=dt_code(x.dtype))
    call("atomnas_expand_bwd", _p(h), _ld(h), _ss(h), _p(c1), _p(x), _ld(x), _p(wt), ldw, _p(add), _ld(add) if add is not None else 0, _p(gx), _ld(gx),
         _p(dwe), _p(ws), ws.numel(), _p(mp), mp.stride(0) if mp is not None else 0, _p(vb), M, inp, hid, dt_code(x.dtype), _stream())


def project_bwd_supported(oup, hid, dtype):
    return bool(_lib.load().atomnas_project_bwd_supported(int(oup), int(hid), dt_code(dtype)))


def project_bwd_dp_supported(M, oup, hid, g, z, gh, stat_rows):
    """whether atomnas_project_bwd serves these tensors (layouts, sizes)"""
    return bool(_lib.load().atomnas_project_bwd_dp_supported(int(M), int(oup), int(hid), _ld(g), _ld(z), _ss(z), _ld(gh), _ss(gh), int(stat_rows),
                                                             dt_code(g.dtype)))


def project_bwd(g, wpt_pack, z, zscale, zshift, act, gh, stats, dwp, si, sj, M, oup, hid, stat_rows=None, ws=None):
    """Fused backward of the projection (include/atomnas_hip.h): g = dP (bnbwd_apply's output); masked input gradient gh with the
    BN-backward statistics, and dwp[o*si + n*sj] += dP^T act(bn(z)), from one pass over z (oup % 8 == 0, oup <= 64)."""
    _chk_cuda(g, z, gh, dwp, wpt_pack)
    if ws is None:
        ws = torch.empty(min(512 * oup * hid, 16 << 20), dtype=torch.float32, device=g.device)
    _keep(ws)
    if _lib.PROFILE is not None:
        _lib.profile_tag("M%d N%d K%d fusedbwd+dP" % (M, hid, oup))
    _rec("project_bwd", M=int(M), oup=int(oup), hid=int(hid), g=_lay(g), ldw=int(wpt_pack.stride(0)), z=_lay(z), act=int(act), gh=_lay(gh),
         stat_rows=_rows(stats, stat_rows), si=int(si), sj=int(sj), ws_floats=int(ws.numel()), dt=dt_code(g.dtype))
    call("atomnas_project_bwd", _p(g), _ld(g), _p(wpt_pack), wpt_pack.stride(0), _p(z), _ld(z), _ss(z), _p(zscale), _p(zshift), int(act), _p(gh),
         _ld(gh), _ss(gh), _p(stats), _rows(stats, stat_rows), _p(dwp), si, sj, _p(ws), ws.numel(), M, oup, hid, dt_code(g.dtype), _stream())


def bn_finalize_fwd(stats, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, scale, shift, save_mean,
                    save_invstd, C, stat_rows=None, stat_ld=None, cmap=None):
    _rec("bn_finalize_fwd", C=int(C), stat_rows=_rows(stats, stat_rows), stat_ld=int(pad8(C) if stat_ld is None else stat_ld), count=float(count),
         momentum=-1.0 if momentum is None else float(momentum), running=running_mean is not None, cmap=cmap is not None)
    call("atomnas_bn_finalize_fwd", _p(stats), _rows(stats, stat_rows), pad8(C) if stat_ld is None else stat_ld, float(count), _p(gamma), _p(beta), eps,
         -1.0 if momentum is None else momentum,
         _p(running_mean), _p(running_var), _p(nbt), _p(scale), _p(shift), _p(save_mean), _p(save_invstd), C, _p(cmap), _stream())


def bn_eval_coeffs(gamma, beta, running_mean, running_var, eps, scale, shift, C, cmap=None):
    call("atomnas_bn_eval_coeffs", _p(gamma), _p(beta), _p(running_mean), _p(running_var), eps, _p(scale), _p(shift), C, _p(cmap), _stream())


def bn_finalize_bwd(stats2, count, gamma, save_mean, save_invstd, rho_ptr, penalty, dgamma, dbeta, c1, c2, c3, C, stat_rows=None,
                    stat_ld=None, cmap=None):
    _rec("bn_finalize_bwd", C=int(C), stat_rows=_rows(stats2, stat_rows), stat_ld=int(pad8(C) if stat_ld is None else stat_ld), count=float(count),
         cmap=cmap is not None)
    call("atomnas_bn_finalize_bwd", _p(stats2), _rows(stats2, stat_rows), pad8(C) if stat_ld is None else stat_ld, float(count), _p(gamma), _p(save_mean), _p(save_invstd),
         _p(rho_ptr), _p(penalty),
         _p(dgamma), _p(dbeta), _p(c1), _p(c2), _p(c3), C, _p(cmap), _stream())


def bn_apply(x, scale, shift, relu, res, y, M, C):
    _rec("bn_apply", M=int(M), C=int(C), x=_lay(x), act=int(relu), res=_lay(res), y=_lay(y), dt=dt_code(x.dtype))
    call("atomnas_bn_apply", _p(x), _ld(x), _p(scale), _p(shift), int(relu), _p(res), _ld(res) if res is not None else 0, _p(y),
         _ld(y), M, C, dt_code(x.dtype), _stream())


def bnbwd_apply(g, x, c1, c2, c3, y, M, C):
    """y = c1*g + c2*x + c3 (plain [M, C] tensors): the differentiated BatchNorm output, materialised once"""
    _chk_cuda(g, x, y)
    _rec("bnbwd_apply", M=int(M), C=int(C), g=_lay(g), x=_lay(x), y=_lay(y), dt=dt_code(g.dtype))
    call("atomnas_bnbwd_apply", _p(g), _ld(g), _p(x), _ld(x), _p(c1), _p(c2), _p(c3), _p(y), _ld(y), M, C, dt_code(g.dtype), _stream())


def bn_act_pool(x, scale, shift, relu, pooled, keep, drop_p, seed, step_ptr, N, HW, C):
    _rec("bn_act_pool", N=int(N), HW=int(HW), C=int(C), x=_lay(x), act=int(relu), drop_p=float(drop_p), keep=keep is not None, dt=dt_code(x.dtype))
    call("atomnas_bn_act_pool", _p(x), _ld(x), _p(scale), _p(shift), int(relu), _p(pooled), _ld(pooled), _p(keep), float(drop_p),
         int(seed) & 0xFFFFFFFFFFFFFFFF, _p(step_ptr), N, HW, C, dt_code(x.dtype), _stream())


def pool_act_bwd(dpooled, keep, drop_p, x, scale, shift, relu, g, stats2, N, HW, C, stat_rows=None):
    _rec("pool_act_bwd", N=int(N), HW=int(HW), C=int(C), x=_lay(x), act=int(relu), drop_p=float(drop_p), keep=keep is not None,
         stat_rows=_rows(stats2, stat_rows), dt=dt_code(x.dtype))
    call("atomnas_pool_act_bwd", _p(dpooled), _ld(dpooled), _p(keep), float(drop_p), _p(x), _ld(x), _p(scale), _p(shift), int(relu),
         _p(g), _ld(g), _p(stats2), _rows(stats2, stat_rows), N, HW, C, dt_code(x.dtype), _stream())


def act_bwd_stats(dy, z, scale, shift, relu, g, stats2, M, C, stat_rows=None):
    _rec("act_bwd_stats", M=int(M), C=int(C), dy=_lay(dy), z=_lay(z), masked=scale is not None, act=int(relu), g=_lay(g), stat_rows=_rows(stats2, stat_rows),
         dt=dt_code(dy.dtype))
    call("atomnas_act_bwd_stats", _p(dy), _ld(dy), _p(z), _ld(z), _p(scale), _p(shift), int(relu), _p(g),
         _ld(g) if g is not None else 0, _p(stats2), _rows(stats2, stat_rows), M, C, dt_code(dy.dtype), _stream())


def se_pool_parts(N, HW, C):
    """planes [parts][N][C] the per-image sums of se_squeeze / se_bwd_gate's dgate pass are produced in (include/atomnas_hip.h)"""
    return int(_lib.load().atomnas_se_pool_parts(int(N), int(HW), int(C)))


def se_squeeze(d, scale, shift, act, pooled_parts, N, HW, C):
    """pooled_parts: fp32 [parts][N][C], parts = se_pool_parts(N, HW, C)"""
    parts = pooled_parts.shape[0]
    call("atomnas_se_squeeze", _p(d), _ld(d), _ss(d), _p(scale), _p(shift), int(act), _p(pooled_parts), pooled_parts.stride(1), parts,
         pooled_parts.stride(0), N, HW, C, dt_code(d.dtype), _stream())


def se_mlp_fwd(pooled_parts, pooled, cmap, w1p, b1, w2t, b2p, act, hpre, gate, N, HT, hid):
    """w1p [hid][HT], w2t [hid][HT], b2p [HT]: the gate's dense layers packed over the padded channel layout (runtime.py);
    pooled [N][HT] receives the sum of the planes of pooled_parts"""
    call("atomnas_se_mlp_fwd", _p(pooled_parts), pooled_parts.stride(1), pooled_parts.shape[0], pooled_parts.stride(0), _p(pooled), _p(cmap),
         _p(w1p), _p(b1), _p(w2t), _p(b2p), int(act), _p(hpre), _p(gate), N, HT, hid, _stream())


def se_scale(d, scale, shift, act, gate, out, M, HW, C):
    call("atomnas_se_scale", _p(d), _ld(d), _ss(d), _p(scale), _p(shift), int(act), _p(gate), gate.stride(0), _p(out), _ld(out), _ss(out), M,
         HW, C, dt_code(d.dtype), _stream())


def se_bwd_gate(ds, d, scale, shift, act, gate, pooled, cmap, w1p, w2t, hpre,  # dgate: fp32 [parts][N][HT] workspace
                dgate, dz2, dz1, dpooled, dw1, db1, dw2, db2, N, HW, HT, total,
                hid, se_act=None):
    call("atomnas_se_bwd_gate", _p(ds), _ld(ds), _ss(ds), _p(d), _ld(d), _ss(d), _p(scale), _p(shift), int(act), _p(gate), _p(pooled),
         gate.stride(0), _p(cmap), _p(w1p), _p(w2t), _p(hpre), _p(dgate), dgate.shape[0], dgate.stride(0), _p(dz2), _p(dz1), _p(dpooled), _p(dw1), _p(db1), _p(dw2), _p(db2),
         int(act if se_act is None else se_act), N, HW, HT, total, hid, dt_code(d.dtype), _stream())


def se_bwd_apply(ds, d, scale, shift, act, gate, dpooled, g, stats2, M, HW, C, stat_rows=None):
    call("atomnas_se_bwd_apply", _p(ds), _ld(ds), _ss(ds), _p(d), _ld(d), _ss(d), _p(scale), _p(shift), int(act), _p(gate), _p(dpooled),
         gate.stride(0), _p(g), _ld(g), _ss(g), _p(stats2), _rows(stats2, stat_rows), M, HW, C, dt_code(d.dtype), _stream())


def im2col_stem(img, col, N, H, W):
    assert img.dtype == torch.float32 and img.is_contiguous()
    _rec("im2col_stem", N=int(N), H=int(H), W=int(W), col=_lay(col), dt=dt_code(col.dtype))
    call("atomnas_im2col_stem", _p(img), _p(col), _ld(col), N, H, W, dt_code(col.dtype), _stream())


def ce_smooth(logits, target, eps, B, K, loss_per_sample, dlogits, gscale, topk):
    assert logits.dtype == torch.float32 and target.dtype == torch.int64
    _rec("ce_smooth", B=int(B), K=int(K), ldl=_lay(logits), dl=_lay(dlogits), dt=dt_code(dlogits.dtype) if dlogits is not None else 0)
    call("atomnas_ce_smooth", _p(logits), _ld(logits), _p(target), float(eps), B, K, _p(loss_per_sample), _p(dlogits),
         _ld(dlogits) if dlogits is not None else 0, float(gscale), _p(topk),
         dt_code(dlogits.dtype) if dlogits is not None else 0, _stream())


def colsum(x, out, M, C):
    _rec("colsum", M=int(M), C=int(C), x=_lay(x), dt=dt_code(x.dtype))
    call("atomnas_colsum", _p(x), _ld(x), _p(out), M, C, dt_code(x.dtype), _stream())


def fused_rmsprop_ema(p, g, sq, buf, ema, wd_chunk, n, hyper, alpha, eps, eps_inside_sqrt, momentum, l2_value=None, ws=None):
    if l2_value is not None and ws is None:
        ws = torch.empty(4096, dtype=torch.float32, device=p.device)
    call("atomnas_fused_rmsprop_ema", _p(p), _p(g), _p(sq), _p(buf), _p(ema), _p(wd_chunk), n, _p(hyper), float(alpha), float(eps),
         int(eps_inside_sqrt), float(momentum), _p(l2_value), _p(ws), _stream())


def vec_sum(x, n, scale, out):
    call("atomnas_vec_sum", _p(x), int(n), float(scale), _p(out), _stream())


def ema_update(shadow, x, n, hyper):
    call("atomnas_ema_update", _p(shadow), _p(x), n, _p(hyper), _stream())


def zero_(t):
    """t[...] = 0 with the library's fill kernel (contiguous storage, 16-byte aligned); returns t"""
    if isinstance(t, Slab):
        zero_(t.t)
        return t
    assert t.is_contiguous()
    if t.numel():
        call("atomnas_zero", _p(t), t.numel() * t.element_size(), _stream())
    return t


def zeros(*shape, dtype, device):
    return zero_(torch.empty(*shape, dtype=dtype, device=device))


def add_i64(t, v):
    assert t.dtype == torch.int64 and t.is_contiguous()
    call("atomnas_add_i64", _p(t), t.numel(), int(v), _stream())


def scale_by(x, n, hyper, idx):
    call("atomnas_scale_by", _p(x), n, _p(hyper), idx, _stream())


def pack_weights(arena, packbuf, jobs_dev, njobs, dtype):
    call("atomnas_pack_weights", _p(arena), _p(packbuf), _p(jobs_dev), njobs, dt_code(dtype), _stream())


def gamma_mask(params, ema, jobs_dev, njobs, threshold, mode, mask, index, kept):
    call("atomnas_gamma_mask", _p(params), _p(ema), _p(jobs_dev), njobs, float(threshold), mode, _p(mask), _p(index), _p(kept),
         _stream())


def reg_grad(p, g, jobs_dev, njobs, use_sign, mult_ptr=None, grad_out=None):
    call("atomnas_reg_grad", _p(p), _p(g), _p(jobs_dev), njobs, int(use_sign), _p(mult_ptr), _p(grad_out), _stream())


def reg_value(p, jobs_dev, njobs, use_abs, mult_ptr, post_scale, out, ws=None):
    if ws is None:
        ws = torch.empty(64 * njobs, dtype=torch.float32, device=p.device)
    call("atomnas_reg_value", _p(p), _p(jobs_dev), njobs, int(use_abs), _p(mult_ptr), float(post_scale), _p(out), _p(ws), _stream())


# ---- shrink plumbing.  A shrink computes ALL alive masks, their ascending kept-channel indices and counts in one launch
# (atomnas_gamma_mask); register_masks() makes them known here, so that the per-tensor protocol of the reference
# (info['mask_hook'](new, old, mask): models/compress_utils.py:31-37) neither recomputes an index per tensor nor synchronises for a count.
# While gather_defer(True) is on, gathers are RECORDED and run as one launch at gather_flush() (atomnas_gather_jobs).
_MASKS = {}        # (data_ptr, numel) -> (mask tensor kept alive, index tensor or None = identity, kept count)
_GATHER = [None]   # list of recorded jobs while deferring


def register_mask(mask, index, kept):
    """mask: bool / uint8 device tensor; index: int32 device tensor of its kept positions (None: every channel kept); kept: int"""
    _MASKS[(mask.data_ptr(), mask.numel())] = (mask, index, int(kept))


def clear_masks():
    _MASKS.clear()


def mask_count(mask):
    """number of kept channels: from the registry, else one device -> host synchronisation (the reference's mask.sum().item())"""
    r = _MASKS.get((mask.data_ptr(), mask.numel()))
    return r[2] if r is not None else int(mask.detach().sum().item())


def gather_deferring():
    return _GATHER[0] is not None


def gather_defer(on):
    if on:
        _GATHER[0] = []
    else:
        gather_flush()
        _GATHER[0] = None


def gather_flush():
    """runs every recorded gather in one launch (job table in device memory); a no-op when nothing is recorded"""
    import numpy as np
    jobs = _GATHER[0]
    if not jobs:
        return 0
    dt = np.dtype([("src", "<u8"), ("dst", "<u8"), ("index", "<u8"), ("s_os", "<i8"), ("s_ds", "<i8"), ("d_os", "<i8"), ("d_ds", "<i8"),
                   ("outer", "<i4"), ("n_kept", "<i4"), ("inner", "<i4"), ("blk0", "<u4")])
    tab = np.zeros(len(jobs), dtype=dt)
    blk = 0
    for q, (src, dst, index, s_os, s_ds, d_os, d_ds, outer, n_kept, inner) in enumerate(jobs):
        tab[q] = (src.data_ptr(), dst.data_ptr(), index.data_ptr() if index is not None else 0, s_os, s_ds, d_os, d_ds, outer, n_kept, inner, blk)
        blk += (outer * n_kept * inner + 255) // 256
    dev = jobs[0][1].device
    table = torch.from_numpy(tab.view(np.uint8)).to(dev)
    call("atomnas_gather_jobs", _p(table), len(jobs), blk, _stream())
    n = len(jobs)
    _GATHER[0] = [] if _GATHER[0] is not None else None
    _keepalive = (table, jobs)   # the launch is asynchronous: the table and the tensors live until the stream has passed it
    torch.cuda.current_stream().synchronize()
    del _keepalive
    return n


def copy_job(dst, src):
    """dst <- src as an identity job of the deferred gather table (runtime.ArenaManager.materialize migrates ~2,500 tensors per arena
    rebuild).  fp32 tensors on the same GPU, src contiguous, dst contiguous or a 2-D band [rows][cols(,1,1)] of a wider matrix;
    returns False when the pair is not of that form or nothing is being deferred (the caller copies with torch then)."""
    if _GATHER[0] is None or dst.dtype != torch.float32 or src.dtype != torch.float32 or not dst.is_cuda or src.device != dst.device:
        return False
    if dst.numel() == 0:
        return True
    if dst.shape != src.shape or not src.is_contiguous():
        return False
    if dst.is_contiguous():
        _GATHER[0].append((src, dst, None, 0, 1, 0, 1, 1, dst.numel(), 1))
        return True
    if dst.dim() >= 2 and all(d == 1 for d in dst.shape[2:]) and dst.stride(1) == 1:
        rows, cols = dst.shape[0], dst.shape[1]
        _GATHER[0].append((src, dst, None, cols, 1, dst.stride(0), 1, rows, cols, 1))
        return True
    return False


def gather_by_mask(dst, src, mask, dim):
    """dst <- src[mask] (dim 0) or src[:, mask] (dim 1) on device, fp32, arbitrary strides: mask -> ascending kept-channel
    index (atomnas_mask_index), then an index-packed gather (atomnas_gather_dim).  The integer index is bit-exact with
    torch.nonzero by construction (stable prefix sum)."""
    _chk_cuda(dst, src, mask)
    if src.dtype != torch.float32 or dst.dtype != torch.float32:
        raise TypeError("gather_by_mask moves fp32 master tensors")
    n = src.shape[dim]
    reg = _MASKS.get((mask.data_ptr(), mask.numel()))
    if reg is not None:
        index, kept = reg[1], None
    else:
        m8 = mask.to(torch.uint8).contiguous()
        index = torch.empty(n, dtype=torch.int32, device=src.device)
        kept = torch.zeros(1, dtype=torch.int32, device=src.device)
        call("atomnas_mask_index", _p(m8), n, _p(index), _p(kept), _stream())
    n_kept = dst.shape[dim]
    if reg is not None and reg[2] != n_kept:
        raise ValueError("gather_by_mask: the destination keeps %d channels, the registered mask %d" % (n_kept, reg[2]))
    if n_kept == 0:
        return index, kept

    def run(sv_, dv_, s_os, s_ds, d_os, d_ds, outer, inner):
        if _GATHER[0] is not None:
            _GATHER[0].append((sv_, dv_, index, s_os, s_ds, d_os, d_ds, outer, n_kept, inner))
        elif index is None:
            raise RuntimeError("identity masks are only registered for deferred gathers")
        else:
            call("atomnas_gather_dim", _p(sv_), _p(dv_), _p(index), s_os, s_ds, d_os, d_ds, outer, n_kept, inner, _stream())
    if dim == 0:
        inner = 1
        for s_ in src.shape[1:]:
            inner *= s_
        sv, dv = src.reshape(n, inner) if src.is_contiguous() else None, dst.reshape(n_kept, inner) if dst.is_contiguous() else None
        if sv is None or dv is None:
            raise ValueError("dim-0 gather needs contiguous tensors")
        run(sv, dv, 0, inner, 0, inner, 1, inner)
    elif dim == 1:
        outer = src.shape[0]
        inner = 1
        for s_ in src.shape[2:]:
            inner *= s_
        if inner != 1 and not (src.is_contiguous() and dst.is_contiguous()):
            raise ValueError("dim-1 gather of strided tensors supports trailing singleton dimensions only")
        s_os, s_ds = (src.stride(0), src.stride(1)) if inner == 1 else (src.shape[1] * inner, inner)
        d_os, d_ds = (dst.stride(0), dst.stride(1)) if inner == 1 else (dst.shape[1] * inner, inner)
        run(src, dst, s_os, s_ds, d_os, d_ds, outer, inner)
    else:
        raise NotImplementedError()
    return index, kept
