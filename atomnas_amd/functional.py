"""Forward / backward executors of the hot path: sequences of C-ABI kernel launches over arena views, wrapped in
torch.autograd.Function so that the reference's `loss.backward()` drives them.

Activations cross module boundaries as ordinary torch tensors of logical shape [N, C, H, W] in channels_last memory
format (i.e. NHWC in HBM) and the model's compute dtype; inside a block the 6x-expanded tensors are [M, HT] buffers.
Parameter gradients are written straight into the gradient arena (p.grad views) instead of being returned to autograd.

Reference call sites: InvertedResidualChannels.forward (models/mobilenet_base.py:371-382), ConvBNReLU (:120-142),
MobileNetV2.forward (models/mobilenet_supernet.py:169-173), CrossEntropyLabelSmooth.forward (utils/optim.py:199-207).
"""
import os

import torch
from torch import nn

from . import ops
from .ops import PRO_BNBWD, PRO_BNRELU, PRO_NONE, STAT_SQ, STAT_Z
from .ops import Slab
from .runtime import pad8, pads

ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SWISH = 0, 1, 2, 3   # the integer `relu` / `mask` arguments of the C ABI


def act_code(module):
    """Maps the activation module a ConvBNReLU holds to the kernels' activation flag."""
    if module is None:
        return ACT_NONE
    if isinstance(module, nn.ReLU6):
        return ACT_RELU6
    if isinstance(module, nn.ReLU):
        return ACT_RELU
    if type(module).__name__ == "Swish":   # models/mobilenet_base.py:72-80
        return ACT_SWISH
    raise NotImplementedError("activation %s is not supported by the HIP path (ReLU / ReLU6 / Swish)" % type(module).__name__)


# ---------------------------------------------------------------------------------------------- layout plumbing
def to_2d(x, dtype):
    """[N,C,H,W] tensor -> ([M, C] NHWC view/copy in `dtype`, (N, H, W, C))."""
    if x.dim() != 4:
        raise ValueError("expected a 4-D activation, got shape %s" % (tuple(x.shape),))
    if not x.is_cuda:
        raise ops._lib.AtomnasHipError("atomnas_amd runs on the GPU only: input tensor is on %s" % x.device)
    N, C, H, W = x.shape
    if C % 8 != 0:
        raise ValueError("block input channels must be a multiple of 8, got %d" % C)
    if x.dtype != dtype:
        x = x.to(dtype)
    x = x.contiguous(memory_format=torch.channels_last)
    return x.permute(0, 2, 3, 1).reshape(N * H * W, C), (N, H, W, C)


def to_4d(y2d, N, H, W, C):
    return y2d.view(N, H, W, C).permute(0, 3, 1, 2)


def _f32(n, dev, zero=False):
    t = torch.empty(n, dtype=torch.float32, device=dev)
    return ops.zero_(t) if zero else t


class StatBuf:
    """Partial-row statistics buffer [rows][2][c] (include/atomnas_hip.h): every producer writes all rows of its channel range
    with plain stores, the BatchNorm finalize sums them in a fixed order -- no initialisation, no atomics."""
    __slots__ = ("t", "rows", "c")

    def __init__(self, t, rows, c):
        self.t, self.rows, self.c = t, rows, c

    def at(self, off):
        """buffer pointer advanced to channel `off` (a branch segment of a fused hidden tensor)"""
        return self.t[off:] if off else self.t


def _stats(c, dev, mgr=None):
    """Uninitialised StatBuf for c channels.  With a manager it is a slice of the per-step statistics workspace."""
    rows = ops.stat_rows_for(c)
    n = rows * 2 * c
    if mgr is not None:
        v = mgr.take_stats(n)
        if v is not None:
            return StatBuf(v, rows, c)
    return StatBuf(torch.empty(n, dtype=torch.float32, device=dev), rows, c)


# ---------------------------------------------------------------------------------------------- batch norm helpers
class BNState:
    """Per-call coefficients of one (possibly branch-fused) BatchNorm: scale/shift for the apply, mean/invstd for backward."""
    __slots__ = ("scale", "shift", "mean", "invstd")


def bn_forward_coeffs(bn, stats, count, dev):
    """bn: dict of arena views (gamma, beta, rm, rv, C, mods).  Uses batch statistics when the BN modules are in training
    mode (updating running statistics with their momentum; momentum None = cumulative average), running statistics otherwise."""
    C = bn["C"]
    cmap = bn.get("cmap")   # fused block: contiguous [total] parameter vectors against the padded-segment kernel layout (runtime.py)
    Ck = bn["Cpad"] if cmap is not None else C   # channels of the kernel layout: one launch covers them, padding gets zeros
    Cp = pad8(Ck)
    st = BNState()
    st.scale, st.shift = _f32(Cp, dev), _f32(Cp, dev)
    mod = bn["mods"][0]
    eps = mod.eps
    if mod.training or not mod.track_running_stats:
        st.mean, st.invstd = _f32(Cp, dev), _f32(Cp, dev)
        track = mod.track_running_stats
        ops.bn_finalize_fwd(stats.t, count, bn["gamma"], bn["beta"], eps, mod.momentum, bn["rm"] if track else None,
                            bn["rv"] if track else None, mod.num_batches_tracked if track else None, st.scale, st.shift, st.mean,
                            st.invstd, Ck, stat_rows=stats.rows, stat_ld=stats.c, cmap=cmap)
        if track:
            bn["mgr"].bn_trained = True
    else:
        st.mean = st.invstd = None
        ops.bn_eval_coeffs(bn["gamma"], bn["beta"], bn["rm"], bn["rv"], eps, st.scale, st.shift, Ck, cmap=cmap)
    return st


def bn_uses_batch_stats(bn):
    mod = bn["mods"][0]
    return mod.training or not mod.track_running_stats


def bn_backward_coeffs(bn, st, stats2, count, dev):
    """-> (c1, c2, c3) with dx = c1*g + c2*x + c3; writes dgamma / dbeta into the gradient arena."""
    C = bn["C"]
    cmap = bn.get("cmap")
    Ck = bn["Cpad"] if cmap is not None else C
    Cp = pad8(Ck)
    c1, c2, c3 = _f32(Cp, dev), _f32(Cp, dev), _f32(Cp, dev)
    if st.mean is None:
        raise RuntimeError("backward through a BatchNorm in eval mode is not supported")
    ops.bn_finalize_bwd(stats2.t, count, bn["gamma"], st.mean, st.invstd, None, None, bn["dgamma"], bn["dbeta"], c1, c2, c3, Ck,
                        stat_rows=stats2.rows, stat_ld=stats2.c, cmap=cmap)
    return c1, c2, c3


# ---------------------------------------------------------------------------------------------- atomic block
_FUSED_PROJECT_BWD = bool(int(os.environ.get("ATOMNAS_FUSED_PROJECT_BWD", "1")))   # experiment switch (A/B against the two-GEMM form)
# widest block output that takes the fused kernel.  The library covers oup <= 96, but the 80/96-wide instances hold 223 VGPR + 96 AGPR
# (one wave per SIMD) and measured slower than the two-GEMM form on the 14x14 stages: 36.53 vs 36.30 ms/step (r03, bs256 bf16).
_FUSED_PROJECT_BWD_MAXOUP = int(os.environ.get("ATOMNAS_FUSED_PROJECT_BWD_MAXOUP", "48"))
_FUSED_EXPAND_BWD = int(os.environ.get("ATOMNAS_FUSED_EXPAND_BWD", "48"))   # experiment switch: widest inp that takes the fused kernel (0: never)
# Expand backward without the raw expand output E (csrc/xbwd.hip): with dE = c1*h + c2*E + c3 and E = x We^T the c2 / c3 terms are
# inp x inp sized corrections (Gram matrix of x), so the wide GEMMs read h only.  Widest block input that takes this form (0: never;
# bf16 only).  48 = stages 1-3 with the streaming kernel k_expand_bwd_s (same-box A/Bs: 31.15 -> 30.20 ms for inp <= 24, -> 30.11 with
# the per-segment launches of 40 -> 720; the 31.41 of profiles/r04_expand_bwd_noe_ab.txt for 48 predates that kernel).
_EXPAND_BWD_NOE = int(os.environ.get("ATOMNAS_EXPAND_BWD_NOE", "48"))
_PLAIN_HIDDEN = bool(int(os.environ.get("ATOMNAS_PLAIN_HIDDEN", "0")))
# fused block (AtomNAS+): one weight-gradient GEMM per layer into a padded scratch + fold jobs (0: one GEMM per kernel-size segment)
_FUSED_WG_BATCH = bool(int(os.environ.get("ATOMNAS_FUSED_WG_BATCH", "1")))
_DP_TENSOR = bool(int(os.environ.get("ATOMNAS_DP_TENSOR", "1")))   # experiment switch: 0 = the BatchNorm-backward prologue in every GEMM tile
TAIL_TAP = None   # set to a list by tests to receive the dropout keep mask of every tail forward
# set to a list by tests to receive, for every activation the forward applies, (kind, plan, raw tensor, scale, shift): the pre-activation
# is raw * scale + shift per channel -- what a test needs to compare ReLU masks with the oracle's (tests/test_block_gpu.py)
ACT_TAP = None


def _tap(kind, pl, raw, st):
    if ACT_TAP is not None and st is not None:
        ACT_TAP.append((kind, pl, raw, st.scale, st.shift))


# set to a list by tests to receive (name, tensor) for every intermediate an executor produces in a step: activations, gradients,
# BatchNorm coefficients ("<plan>.<bn>.invstd" etc.) -- tests/test_bench_shapes_gpu.py checks them for finiteness at the bench's sizes
STEP_TAP = None


def _stap(pl, **tensors):
    if STEP_TAP is not None:
        for what, t in tensors.items():
            if t is None:
                continue
            if isinstance(t, BNState):
                for f in ("scale", "shift", "mean", "invstd"):
                    if getattr(t, f) is not None:
                        STEP_TAP.append(("%s.%s.%s" % (pl.name, what, f), getattr(t, f)))
            elif isinstance(t, (tuple, list)):
                for q, u in enumerate(t):
                    STEP_TAP.append(("%s.%s%d" % (pl.name, what, q + 1), u))
            else:
                STEP_TAP.append(("%s.%s" % (pl.name, what), t))
_CHECK_LOSS_SEED = bool(int(os.environ.get("ATOMNAS_CHECK_LOSS_SEED", "0")))   # experiment switch (same-box A/B of the two layouts)


def _hidden(pl, M, C, T, dev):
    """hidden tensor of a block: slab-major (ops.Slab) for expanding blocks, plain [M, C] for the narrow non-expanding one"""
    return Slab(M, C, T, dev) if (pl.expand and not _PLAIN_HIDDEN) else torch.empty(M, C, dtype=T, device=dev)


def _seg(t, o):
    return t.seg(o) if isinstance(t, Slab) else (t[:, o:] if o else t)


def block_forward(pl, x2d, N, H, W, need_grad):
    """InvertedResidualChannels.forward on arena views.  Returns (out2d, saved) -- saved is None when need_grad is False."""
    dev, T = x2d.device, x2d.dtype
    M = N * H * W
    s = pl.stride
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    M2 = N * Ho * Wo
    HT = pl.HT
    act = pl.act
    sv = {}
    if pl.expand:
        bs = bn_uses_batch_stats(pl.bne)
        E = _hidden(pl, M, HT, T, dev)
        stE = _stats(HT, dev, pl.bne["mgr"]) if bs else None
        ops.gemm_nt(x2d, pl.We_pack, E, M, HT, pl.inp, stats=stE.t if bs else None, stat_mode=STAT_SQ if bs else 0,
                    stat_rows=stE.rows if bs else None)
        bE = bn_forward_coeffs(pl.bne, stE, M, dev)
        _tap("expand", pl, E, bE)
    else:
        E, bE = x2d, None
    bsd = bn_uses_batch_stats(pl.bnd)
    D = _hidden(pl, M2, HT, T, dev)
    stD = _stats(HT, dev, pl.bnd["mgr"]) if bsd else None
    for i in range(pl.nb):
        o, c = pl.seg[i], pl.segpad(pl.hid[i])
        xin = _seg(E, o) if pl.expand else E
        ops.dwconv_fwd(xin, bE.scale[o:] if bE else None, bE.shift[o:] if bE else None, act if bE else 0, pl.taps[i], _seg(D, o),
                       stD.at(o) if bsd else None, HT, N, H, W, c, pl.ks[i], s, stat_rows=stD.rows if bsd else None)
    bD = bn_forward_coeffs(pl.bnd, stD, M2, dev)
    _tap("dw", pl, D, bD)
    bsp = bn_uses_batch_stats(pl.bnp)
    Pr = torch.empty(M2, pl.oup, dtype=T, device=dev)
    stP = _stats(pl.oup, dev, pl.bnp["mgr"]) if bsp else None
    se = None
    if pl.se:
        # SqueezeAndExcitation (models/mobilenet_base.py:109-112) on the activated depthwise output A = act(bn(D)):
        # squeeze -> two tiny dense layers -> gate; the gated tensor S is the projection's operand
        HWo = Ho * Wo
        se = dict(pooled=_f32(N * HT, dev).view(N, HT), gate=_f32(N * HT, dev).view(N, HT), hpre=_f32(N * pl.se_hid, dev).view(N, pl.se_hid))
        parts = ops.se_pool_parts(N, HWo, HT)
        pooled_parts = _f32(parts * N * HT, dev).view(parts, N, HT)
        ops.se_squeeze(D, bD.scale, bD.shift, int(act), pooled_parts, N, HWo, HT)
        ops.se_mlp_fwd(pooled_parts, se["pooled"], pl.cmap, pl.se_w1p, pl.se_b1, pl.se_w2t, pl.se_b2p, pl.se_act, se["hpre"], se["gate"], N, HT, pl.se_hid)
        Sx = _hidden(pl, M2, HT, T, dev)
        ops.se_scale(D, bD.scale, bD.shift, int(act), se["gate"], Sx, M2, HWo, HT)
        se["S"] = Sx
        ops.gemm_nt(Sx, pl.Wp_pack, Pr, M2, pl.oup, HT, stats=stP.t if bsp else None, stat_mode=STAT_SQ if bsp else 0,
                    stat_rows=stP.rows if bsp else None)
    else:
        ops.gemm_nt(D, pl.Wp_pack, Pr, M2, pl.oup, HT, a_mode=PRO_BNRELU, ac1=bD.scale, ac2=bD.shift, a_relu=int(act),
                    stats=stP.t if bsp else None, stat_mode=STAT_SQ if bsp else 0, stat_rows=stP.rows if bsp else None)
    bP = bn_forward_coeffs(pl.bnp, stP, M2, dev)
    out = torch.empty(M2, pl.oup, dtype=T, device=dev)
    ops.bn_apply(Pr, bP.scale, bP.shift, False, x2d if pl.res else None, out, M2, pl.oup)
    _stap(pl, E=E if pl.expand else None, bne=bE, D=D, bnd=bD, P=Pr, bnp=bP, out=out)
    if need_grad:
        sv = dict(x=x2d, E=E, D=D, P=Pr, bE=bE, bD=bD, bP=bP, dims=(N, H, W, Ho, Wo), se=se)
        return out, sv
    return out, None


def block_backward(pl, sv, G):
    """G = dL/d(out) [M2, oup] -> dL/dx [M, inp]; parameter gradients go to the gradient arena."""
    dev, T = G.device, G.dtype
    N, H, W, Ho, Wo = sv["dims"]
    M, M2 = N * H * W, N * Ho * Wo
    HT, s, act = pl.HT, pl.stride, pl.act
    x2d, E, D, Pr, bE, bD, bP = sv["x"], sv["E"], sv["D"], sv["P"], sv["bE"], sv["bD"], sv["bP"]
    # shared pw_bn backward: statistics pass over (G, P), then coefficients
    st2P = _stats(pl.oup, dev, pl.bnp["mgr"])
    ops.act_bwd_stats(G, Pr, None, None, False, None, st2P.t, M2, pl.oup, stat_rows=st2P.rows)
    p1, p2, p3 = bn_backward_coeffs(pl.bnp, bP, st2P, M2, dev)
    se = sv.get("se")
    # projection weight gradient: dWp[n][k] = sum_m dP[m][n] * A'[m][k], A' = act(bn(D)) (gated by the SE when there is one).
    # The fused block's projection weight is one contiguous [oup, total] tensor: one launch per branch segment.
    wp_jobs = ([(sg, h, pl.Wp_grad[stt:], pl.total) for sg, stt, h in zip(pl.seg, pl.start, pl.hid)] if pl.fused
               else [(0, HT, pl.Wp_grad, HT)])
    # early stages (oup <= 48, no SE): the weight gradient rides in the input-gradient kernel below (one pass over D), which takes the
    # differentiated pw_bn output dP as a tensor
    g = _hidden(pl, M2, HT, T, dev)
    st2D = _stats(HT, dev, pl.bnd["mgr"])
    fused_pb = (_FUSED_PROJECT_BWD and _DP_TENSOR and se is None and not pl.fused and pl.expand and pl.oup <= _FUSED_PROJECT_BWD_MAXOUP
                and ops.project_bwd_supported(pl.oup, HT, T) and ops.project_bwd_dp_supported(M2, pl.oup, HT, G, D, g, st2D.rows))
    # late stages (oup >= 80: every row of dP feeds 23..54 GEMM tiles): the differentiated pw_bn output dP = p1*G + p2*P + p3 is
    # materialised once (a few MB) and the GEMMs below read it without a prologue -- the input-gradient GEMM then takes the streaming
    # kernel (k_gemm_nt_st), measured 84 / 102 / 74 us against 158 / 198 / 208 us with the prologue (14x14 80 / 96 wide, 7x7)
    dP = None
    if _DP_TENSOR and not fused_pb and T == torch.bfloat16 and pl.oup % 8 == 0:
        dP = torch.empty(M2, pl.oup, dtype=T, device=dev)
        ops.bnbwd_apply(G, Pr, p1, p2, p3, dP, M2, pl.oup)
    if pl.fused and _FUSED_WG_BATCH:
        # ONE launch over the padded width into the layer's scratch matrix, folded into the contiguous [oup, total] gradient per segment
        mgr = pl.mgr
        wp_jobs = [(0, HT, mgr.FW[pl.Wp_scratch_off:pl.Wp_scratch_off + pl.oup * HT], HT)]
    for sg, nv, out, si in ([] if fused_pb else wp_jobs):
        if se is not None and dP is not None:
            ops.gemm_tn(dP, pl.oup, _seg(se["S"], sg), nv, out, si, 1, M2)
        elif se is not None:
            ops.gemm_tn(G, pl.oup, _seg(se["S"], sg), nv, out, si, 1, M2, u_mode=PRO_BNBWD, u2=Pr, uc1=p1, uc2=p2, uc3=p3)
        elif dP is not None:
            ops.gemm_tn(dP, pl.oup, _seg(D, sg), nv, out, si, 1, M2, v_mode=PRO_BNRELU, vc1=bD.scale[sg:], vc2=bD.shift[sg:],
                        v_relu=int(act))
        else:
            ops.gemm_tn(G, pl.oup, _seg(D, sg), nv, out, si, 1, M2, u_mode=PRO_BNBWD, u2=Pr, uc1=p1, uc2=p2, uc3=p3, v_mode=PRO_BNRELU,
                        vc1=bD.scale[sg:], vc2=bD.shift[sg:], v_relu=int(act))
    if pl.fused and _FUSED_WG_BATCH:
        ops.fold_jobs(pl.mgr, pl.fold_first, pl.fold_np)
    if se is not None:
        # gradient wrt the gated tensor, then back through the gate (models/mobilenet_base.py:109-112) and the activation
        HWo = Ho * Wo
        dS = _hidden(pl, M2, HT, T, dev)
        if dP is not None:   # no prologue: the wide-output GEMM takes the streaming kernel (k_gemm_nt_st)
            ops.gemm_nt(dP, pl.WpT_pack, dS, M2, HT, pl.oup)
        else:
            ops.gemm_nt(G, pl.WpT_pack, dS, M2, HT, pl.oup, a_mode=PRO_BNBWD, a2=Pr, ac1=p1, ac2=p2, ac3=p3)
        nh = N * pl.se_hid
        dz2, dpooled = (_f32(N * HT, dev).view(N, HT) for _ in range(2))
        parts = ops.se_pool_parts(N, HWo, HT)
        dgate = _f32(parts * N * HT, dev).view(parts, N, HT)
        dz1 = _f32(nh, dev).view(N, pl.se_hid)
        ops.se_bwd_gate(dS, D, bD.scale, bD.shift, int(act), se["gate"], se["pooled"], pl.cmap, pl.se_w1p, pl.se_w2t, se["hpre"], dgate, dz2,
                        dz1, dpooled, pl.se_dw1, pl.se_db1, pl.se_dw2, pl.se_db2, N, HWo, HT, pl.total, pl.se_hid, se_act=pl.se_act)
        ops.se_bwd_apply(dS, D, bD.scale, bD.shift, int(act), se["gate"], dpooled, g, st2D.t, M2, HWo, HT, stat_rows=st2D.rows)
    elif fused_pb:
        # dP once (a narrow tensor), then the streaming fused kernel: no prologue, nothing behind a branch
        dPf = torch.empty(M2, pl.oup, dtype=T, device=dev)
        ops.bnbwd_apply(G, Pr, p1, p2, p3, dPf, M2, pl.oup)
        ops.project_bwd(dPf, pl.WpT_pack, D, bD.scale, bD.shift, int(act), g, st2D.t, pl.Wp_grad, HT, 1, M2, pl.oup, HT, stat_rows=st2D.rows)
    else:
        # projection input gradient, masked by the depthwise activation, with the depthwise-BN backward statistics
        if dP is not None:
            ops.gemm_nt(dP, pl.WpT_pack, g, M2, HT, pl.oup, z=D, zscale=bD.scale, zshift=bD.shift, mask=int(act), stats=st2D.t,
                        stat_mode=STAT_Z, stat_rows=st2D.rows)
        else:
            ops.gemm_nt(G, pl.WpT_pack, g, M2, HT, pl.oup, a_mode=PRO_BNBWD, a2=Pr, ac1=p1, ac2=p2, ac3=p3, z=D, zscale=bD.scale,
                        zshift=bD.shift, mask=int(act), stats=st2D.t, stat_mode=STAT_Z, stat_rows=st2D.rows)
    d1, d2, d3 = bn_backward_coeffs(pl.bnd, bD, st2D, M2, dev)
    # depthwise backward per branch
    if pl.expand:
        h = _hidden(pl, M, HT, T, dev)
        st2E = _stats(HT, dev, pl.bne["mgr"])
    else:
        h = ops.zeros(M, pl.inp, dtype=T, device=dev) if pl.nb > 1 else torch.empty(M, pl.inp, dtype=T, device=dev)
        st2E = None
    for i in range(pl.nb):
        o, c = pl.seg[i], pl.segpad(pl.hid[i])
        if pl.expand:
            ops.dwconv_bwd(_seg(g, o), _seg(D, o), d1[o:], d2[o:], d3[o:], _seg(E, o), bE.scale[o:], bE.shift[o:], act, pl.taps[i],
                           _seg(h, o), pl.Wd_grad[i], st2E.at(o), HT, N, H, W, c, pl.ks[i], s, stat_rows=st2E.rows)
        else:
            if pl.nb > 1:
                raise NotImplementedError("non-expanding block with more than one branch")
            ops.dwconv_bwd(_seg(g, o), _seg(D, o), d1[o:], d2[o:], d3[o:], E, None, None, 0, pl.taps[i], h, pl.Wd_grad[i], None, 0, N, H,
                           W, c, pl.ks[i], s)
    _stap(pl, p=(p1, p2, p3), dP=dP, g=g, d=(d1, d2, d3), h=h)
    if not pl.expand:
        if pl.res:
            h = h + G
        return h
    e1, e2, e3 = bn_backward_coeffs(pl.bne, bE, st2E, M, dev)
    _stap(pl, e=(e1, e2, e3))
    Gx = torch.empty(M, pl.inp, dtype=T, device=dev)
    if (T == torch.bfloat16 and not pl.fused and pl.inp <= min(_EXPAND_BWD_NOE, 64) and pl.inp % 8 == 0 and x2d.stride(0) % 8 == 0):   # atomnas_gram: inp <= 64, row pitch % 8
        return _expand_backward_noe(pl, x2d, h, e1, e2, e3, G if pl.res else None, Gx, M, HT, dev, T)
    # expand weight gradient dWe[n][k] = sum_m dE[m][n] * x[m][k]  (written transposed: out[i=k][j=n] -> dWe[n*inp + k]); the
    # fused block's expand weight is one contiguous [total, inp] tensor: one launch per branch segment
    we_jobs = ([(sg, hh, pl.We_grad[stt * pl.inp:]) for sg, stt, hh in zip(pl.seg, pl.start, pl.hid)] if pl.fused
               else [(0, HT, pl.We_grad)])
    if pl.fused and _FUSED_WG_BATCH:
        we_jobs = [(0, HT, pl.mgr.FW[pl.We_scratch_off:pl.We_scratch_off + HT * pl.inp])]
    for sg, nv, out in we_jobs:
        ops.gemm_tn(x2d, pl.inp, _seg(h, sg), nv, out, 1, pl.inp, M, v_mode=PRO_BNBWD, v2=_seg(E, sg), vc1=e1[sg:], vc2=e2[sg:],
                    vc3=e3[sg:])
    if pl.fused and _FUSED_WG_BATCH:
        ops.fold_jobs(pl.mgr, pl.fold_first + pl.fold_np, pl.fold_ne)
    # expand input gradient (+ residual branch)
    ops.gemm_nt(h, pl.WeT_pack, Gx, M, pl.inp, HT, a_mode=PRO_BNBWD, a2=E, ac1=e1, ac2=e2, ac3=e3, add=G if pl.res else None)
    return Gx


def _plan_buffer(pl, name, make):
    """small per-plan scratch that survives across steps (created on first use, i.e. in an eager step before any graph capture)"""
    cache = pl.__dict__.setdefault("_scratch", {})
    t = cache.get(name)
    if t is None:
        t = cache[name] = make()
    return t


def _expand_backward_noe(pl, x2d, h, e1, e2, e3, res, Gx, M, HT, dev, T):
    """Backward of the expand convolution from ONE hidden stream (models/mobilenet_base.py:316-320 backward).  With the BatchNorm
    backward dE = e1*h + e2*E + e3 and E = x We^T:
        dX  = (e1*h) We + x M + v (+ residual),        M = We^T diag(e2) We,  v = e3^T We
        dWe = (e1*h)^T x + diag(e2) We (X^T X) + e3 (sum x)^T
    The e2 / e3 terms are inp x inp sized (atomnas_gram + atomnas_xb_coeffs); the wide GEMMs read h alone (e1 as their scale
    prologue), where the BNBWD-prologue forms read h and E."""
    inp = pl.inp
    gram = torch.empty(inp * inp, dtype=torch.float32, device=dev)
    sx = torch.empty(inp, dtype=torch.float32, device=dev)
    ops.gram(x2d, M, inp, gram, sx, ws=_plan_buffer(pl, "gram_ws", lambda: torch.empty(2048 * (inp * inp + inp), dtype=torch.float32, device=dev)))
    # M packed as a gemm_nt weight (padding stays zero: the buffer is created zeroed once and only its inp x inp corner is rewritten)
    mp = _plan_buffer(pl, "xb_mp", lambda: ops.zeros((inp + 63) // 64 * 64, (inp + 31) // 32 * 32, dtype=T, device=dev))
    vb = torch.empty(pad8(inp), dtype=torch.float32, device=dev)
    ops.xb_coeffs(e2, e3, pl.We_pack, gram, sx, inp, HT, mp, vb, pl.We_grad)
    if inp <= _FUSED_EXPAND_BWD and ops.expand_bwd_supported(inp, HT, T):
        # one pass over h: both gradients, x M + v added inside the kernel
        ops.expand_bwd(h, e1, x2d, pl.WeT_pack, res, Gx, pl.We_grad, M, inp, HT, mp=mp, vb=vb)
        return Gx
    if (inp <= _FUSED_EXPAND_BWD and not pl.fused and pl.nb > 1 and isinstance(h, ops.Slab)
            and all(ops.expand_bwd_supported(inp, pl.segpad(hh), T) for hh in pl.hid)):
        # 40 -> 720: the accumulators of the whole hidden width do not fit, those of one branch segment (240 channels) do: one launch
        # of the streaming kernel per segment; the input gradient accumulates through `add` (a launch reads and writes its own rows of
        # Gx only), x M + v rides in the first launch
        for i in range(pl.nb):
            o, c = pl.seg[i], pl.segpad(pl.hid[i])
            ops.expand_bwd(_seg(h, o), e1[o:], x2d, pl.WeT_pack[:, o:], res if i == 0 else Gx, Gx, pl.We_grad[o * inp:], M, inp,
                           c, mp=mp if i == 0 else None, vb=vb if i == 0 else None)
        return Gx
    gx1 = torch.empty(M, inp, dtype=T, device=dev)
    ops.gemm_nt(x2d, mp, gx1, M, inp, inp, bias=vb, add=res)
    zeros = _plan_buffer(pl, "xb_zero%d" % e1.numel(), lambda: ops.zeros(e1.numel(), dtype=torch.float32, device=dev))
    ops.gemm_tn(x2d, inp, h, HT, pl.We_grad, 1, inp, M, v_mode=PRO_BNRELU, vc1=e1, vc2=zeros, v_relu=0)
    ops.gemm_nt(h, pl.WeT_pack, Gx, M, inp, HT, a_mode=PRO_BNRELU, ac1=e1, ac2=zeros, a_relu=0, add=gx1)
    return Gx


class BlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, pl):
        T = pl.mgr.compute_dtype
        x2d, (N, H, W, C) = to_2d(x, T)
        if C != pl.inp:
            raise ValueError("block %s expects %d input channels, got %d" % (pl.name, pl.inp, C))
        need = torch.is_grad_enabled() or ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        out, sv = block_forward(pl, x2d, N, H, W, True)
        ctx.pl, ctx.sv = pl, sv
        Ho, Wo = sv["dims"][3], sv["dims"][4]
        return to_4d(out, N, Ho, Wo, pl.oup)

    @staticmethod
    def backward(ctx, gout):
        pl, sv = ctx.pl, ctx.sv
        G, _ = to_2d(gout, pl.mgr.compute_dtype)
        N, H, W, _, _ = sv["dims"]
        gx = block_backward(pl, sv, G)
        _stap(pl, gx=gx)
        ctx.sv = None
        pl.mgr.grad_done(pl)
        return to_4d(gx, N, H, W, pl.inp), None, None


def run_block(pl, x, anchor):
    if pl.nb == 0:
        return x
    if torch.is_grad_enabled() and (x.requires_grad or anchor.requires_grad):
        return BlockFunction.apply(x, anchor, pl)
    T = pl.mgr.compute_dtype
    x2d, (N, H, W, C) = to_2d(x, T)
    out, _ = block_forward(pl, x2d, N, H, W, False)
    s = pl.stride
    return to_4d(out, N, (H - 1) // s + 1, (W - 1) // s + 1, pl.oup)


# ---------------------------------------------------------------------------------------------- stand-alone SqueezeAndExcitation
class SEFunction(torch.autograd.Function):
    """SqueezeAndExcitation.forward (models/mobilenet_base.py:109-112) as a module call of its own: sigmoid(W2 act(W1 mean(x) + b1) + b2) * x
    with the SE entry points the fused block uses (squeeze -> dense layers -> scale; backward: gate gradient + dense-layer gradients,
    then the gradient wrt x).  Inside InvertedResidualChannelsFused the gate runs in the block's executor on the raw depthwise
    output; this is the same arithmetic on an already activated tensor (identity BatchNorm coefficients, activation mode 0)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, se_act, dtype):
        x2d, (N, H, W, C) = to_2d(x, dtype)
        dev = x.device
        hid = w1.shape[0]
        HT = pads(C)
        M, HW = N * H * W, H * W
        D = ops.zeros(M, HT, dtype=dtype, device=dev)
        D[:, :C] = x2d
        cmap = torch.full((HT,), -1, dtype=torch.int32, device=dev)
        cmap[:C] = torch.arange(C, dtype=torch.int32, device=dev)
        w1m, w2m = w1.detach().reshape(hid, C).float(), w2.detach().reshape(C, hid).float()
        w1p, w2t, b2p = (ops.zeros(hid, HT, dtype=torch.float32, device=dev), ops.zeros(hid, HT, dtype=torch.float32, device=dev),
                         ops.zeros(HT, dtype=torch.float32, device=dev))
        w1p[:, :C] = w1m
        w2t[:, :C] = w2m.t()
        b2p[:C] = b2.detach().float()
        one, zero = torch.ones(HT, dtype=torch.float32, device=dev), ops.zeros(HT, dtype=torch.float32, device=dev)
        one[C:] = 0
        pooled, gate = _f32(N * HT, dev).view(N, HT), _f32(N * HT, dev).view(N, HT)
        hpre = _f32(N * hid, dev).view(N, hid)
        parts = ops.se_pool_parts(N, HW, HT)
        pparts = _f32(parts * N * HT, dev).view(parts, N, HT)
        ops.se_squeeze(D, one, zero, ACT_NONE, pparts, N, HW, HT)
        ops.se_mlp_fwd(pparts, pooled, cmap, w1p, b1.detach().float().contiguous(), w2t, b2p, se_act, hpre, gate, N, HT, hid)
        S = torch.empty(M, HT, dtype=dtype, device=dev)
        ops.se_scale(D, one, zero, ACT_NONE, gate, S, M, HW, HT)
        ctx.save_for_backward(D, one, zero, gate, pooled, cmap, w1p, w2t, hpre)
        ctx.dims = (N, H, W, C, HT, hid, se_act, dtype)
        return to_4d(S[:, :C], N, H, W, C)

    @staticmethod
    def backward(ctx, gout):
        D, one, zero, gate, pooled, cmap, w1p, w2t, hpre = ctx.saved_tensors
        N, H, W, C, HT, hid, se_act, dtype = ctx.dims
        dev = D.device
        M, HW = N * H * W, H * W
        g2d, _ = to_2d(gout, dtype)
        dS = ops.zeros(M, HT, dtype=dtype, device=dev)
        dS[:, :C] = g2d
        dz2, dpooled = _f32(N * HT, dev).view(N, HT), _f32(N * HT, dev).view(N, HT)
        parts = ops.se_pool_parts(N, HW, HT)
        dgate = _f32(parts * N * HT, dev).view(parts, N, HT)
        dz1 = _f32(N * hid, dev).view(N, hid)
        dw1, db1, dw2, db2 = (_f32(hid * C, dev, zero=True), _f32(hid, dev, zero=True), _f32(C * hid, dev, zero=True), _f32(C, dev, zero=True))
        ops.se_bwd_gate(dS, D, one, zero, ACT_NONE, gate, pooled, cmap, w1p, w2t, hpre, dgate, dz2, dz1, dpooled, dw1, db1, dw2, db2, N, HW, HT, C,
                        hid, se_act=se_act)
        g = torch.empty(M, HT, dtype=dtype, device=dev)
        st = _stats(HT, dev)
        ops.se_bwd_apply(dS, D, one, zero, ACT_NONE, gate, dpooled, g, st.t, M, HW, HT, stat_rows=st.rows)
        return (to_4d(g[:, :C], N, H, W, C), dw1.view(hid, C, 1, 1), db1, dw2.view(C, hid, 1, 1), db2, None, None)


def run_se(module, x):
    """stand-alone call of a SqueezeAndExcitation module (NCHW tensor on the GPU, C a multiple of 8)"""
    dtype = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
    return SEFunction.apply(x, module.se_reduce.weight, module.se_reduce.bias, module.se_expand.weight, module.se_expand.bias,
                            act_code(module.active_fn), dtype)


# ---------------------------------------------------------------------------------------------- ConvBNReLU (stem / 1x1 / depthwise)
def convbn_forward(pl, x, need_grad):
    """Stand-alone ConvBNReLU: stem 3x3/s2 on an NCHW fp32 image (im2col + GEMM), 1x1 conv, or depthwise conv."""
    mgr = pl.mgr
    T = mgr.compute_dtype
    dev = x.device
    act = pl.act
    sv = {}
    if pl.groups == 1 and pl.k == 3:
        if pl.cin != 3 or pl.stride != 2:
            raise NotImplementedError("dense 3x3 convolution other than the 3-channel stride-2 stem")
        N, _, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        M = N * Ho * Wo
        col = torch.empty(M, 32, dtype=T, device=dev)
        ops.im2col_stem(x.float().contiguous(), col, N, H, W)
        a2d, K = col, 27
        sv["kind"] = "stem"
    elif pl.groups == 1 and pl.k == 1:
        a2d, (N, H, W, C) = to_2d(x, T)
        Ho, Wo, M, K = H, W, N * H * W, pl.cin
        sv["kind"] = "pw"
    else:
        a2d, (N, H, W, C) = to_2d(x, T)
        s = pl.stride
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        M = N * Ho * Wo
        sv["kind"] = "dw"
    bs = bn_uses_batch_stats(pl.bn)
    Cp = pad8(pl.cout)
    Y = torch.empty(M, Cp, dtype=T, device=dev) if sv["kind"] != "dw" else ops.zeros(M, Cp, dtype=T, device=dev)
    st = _stats(pl.cout, dev, pl.bn["mgr"]) if bs else None
    if sv["kind"] == "dw":
        ops.dwconv_fwd(a2d, None, None, 0, pl.taps, Y, st.t if bs else None, pl.cout, N, H, W, pl.cout, pl.k, pl.stride,
                       stat_rows=st.rows if bs else None)
    else:
        ops.gemm_nt(a2d, pl.W_pack, Y, M, pl.cout, K, stats=st.t if bs else None, stat_mode=STAT_SQ if bs else 0,
                    stat_rows=st.rows if bs else None)
    b = bn_forward_coeffs(pl.bn, st, M, dev)
    if act:
        _tap("convbn", pl, Y, b)
    out = torch.empty(M, Cp, dtype=T, device=dev)
    ops.bn_apply(Y, b.scale, b.shift, int(act), None, out, M, pl.cout)
    _stap(pl, Y=Y, bn=b, out=out)
    if need_grad:
        sv.update(a=a2d, Y=Y, b=b, dims=(N, H, W, Ho, Wo), K=K if sv["kind"] != "dw" else 0)
    return out, (N, Ho, Wo), sv


def convbn_backward(pl, sv, G, need_input_grad):
    dev, T = G.device, G.dtype
    N, H, W, Ho, Wo = sv["dims"]
    M = N * Ho * Wo
    act = pl.act
    Y, b, a2d = sv["Y"], sv["b"], sv["a"]
    g = torch.empty_like(Y)
    st2 = _stats(pl.cout, dev, pl.bn["mgr"])
    ops.act_bwd_stats(G, Y, b.scale if act else None, b.shift if act else None, int(act), g, st2.t, M, pl.cout, stat_rows=st2.rows)
    c1, c2, c3 = bn_backward_coeffs(pl.bn, b, st2, M, dev)
    _stap(pl, g=g, c=(c1, c2, c3))
    if sv["kind"] == "dw":
        h = ops.zeros(N * H * W, pad8(pl.cout), dtype=T, device=dev)
        ops.dwconv_bwd(g, Y, c1, c2, c3, a2d, None, None, 0, pl.taps, h, pl.W_grad, None, 0, N, H, W, pl.cout, pl.k, pl.stride)
        return h
    K = sv["K"]
    # dW[n][k] = sum_m dY[m][n] * a[m][k]
    ops.gemm_tn(a2d, K, g, pl.cout, pl.W_grad, 1, K, M, v_mode=PRO_BNBWD, v2=Y, vc1=c1, vc2=c2, vc3=c3)
    if not need_input_grad or sv["kind"] == "stem":
        return None
    Gx = torch.empty(M, pad8(K), dtype=T, device=dev)
    ops.gemm_nt(g, pl.WT_pack, Gx, M, K, pl.cout, a_mode=PRO_BNBWD, a2=Y, ac1=c1, ac2=c2, ac3=c3)
    return Gx


class ConvBNFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, pl):
        out, (N, Ho, Wo), sv = convbn_forward(pl, x, True)
        ctx.pl, ctx.sv = pl, sv
        ctx.x_needs = x.requires_grad
        return to_4d(out[:, :pl.cout] if out.shape[1] != pl.cout else out, N, Ho, Wo, pl.cout)

    @staticmethod
    def backward(ctx, gout):
        pl, sv = ctx.pl, ctx.sv
        G, _ = to_2d(gout, pl.mgr.compute_dtype)
        gx = convbn_backward(pl, sv, G, ctx.x_needs)
        ctx.sv = None
        pl.mgr.grad_done(pl)
        N, H, W, _, _ = sv["dims"]
        if gx is None:
            return None, None, None
        cin = pl.cin
        return to_4d(gx[:, :cin] if gx.shape[1] != cin else gx, N, H, W, cin), None, None


def run_convbn(pl, x, anchor):
    if pl.cout % 8 != 0:
        raise ValueError("ConvBNReLU output channels must be a multiple of 8 on the HIP path, got %d" % pl.cout)
    if torch.is_grad_enabled() and (x.requires_grad or anchor.requires_grad):
        return ConvBNFunction.apply(x, anchor, pl)
    out, (N, Ho, Wo), _ = convbn_forward(pl, x, False)
    return to_4d(out, N, Ho, Wo, pl.cout)


# ---------------------------------------------------------------------------------------------- fused tail: last conv + pool + dropout + fc
def tail_forward(lp, fp, x, drop_p, training, seed, step_ptr, need_grad):
    """features[-2] (1x1 ConvBNReLU) -> AvgPool2d(H) -> squeeze -> Dropout -> Linear, without materialising the activated map."""
    mgr = lp.mgr
    T = mgr.compute_dtype
    dev = x.device
    a2d, (N, H, W, C) = to_2d(x, T)
    M = N * H * W
    act = lp.act
    bs = bn_uses_batch_stats(lp.bn)
    L = torch.empty(M, lp.cout, dtype=T, device=dev)
    st = _stats(lp.cout, dev, lp.bn["mgr"]) if bs else None
    ops.gemm_nt(a2d, lp.W_pack, L, M, lp.cout, lp.cin, stats=st.t if bs else None, stat_mode=STAT_SQ if bs else 0,
                stat_rows=st.rows if bs else None)
    b = bn_forward_coeffs(lp.bn, st, M, dev)
    if act:
        _tap("convbn", lp, L, b)
    pooled = torch.empty(N, lp.cout, dtype=T, device=dev)
    p = float(drop_p) if training else 0.0
    keep = torch.empty(N, lp.cout, dtype=torch.uint8, device=dev) if p > 0 else None
    if TAIL_TAP is not None:   # tests: the dropout keep mask this forward uses (and its backward re-uses)
        TAIL_TAP.append(keep)
    ops.bn_act_pool(L, b.scale, b.shift, int(act), pooled, keep, p, seed, step_ptr, N, H * W, lp.cout)
    Kc = fp.cout
    logits = torch.empty(N, pad8(Kc), dtype=torch.float32, device=dev)
    ops.gemm_nt(pooled, fp.W_pack, logits, N, Kc, fp.cin, bias=fp.bias)
    _stap(lp, L=L, bn=b, pooled=pooled, logits=logits[:, :Kc])
    sv = None
    if need_grad:
        sv = dict(a=a2d, L=L, b=b, pooled=pooled, keep=keep, p=p, dims=(N, H, W))
    return logits[:, :Kc], sv


def tail_backward(lp, fp, sv, dlogits, dl_padded=None):
    """dlogits [N, K] (any float dtype) -> gradient wrt the tail input [M, cin].  dl_padded: the same gradient already in the
    compute dtype with the channel padding zeroed (what atomnas_ce_smooth writes), used as is."""
    mgr = lp.mgr
    T = mgr.compute_dtype
    dev = sv["L"].device
    N, H, W = sv["dims"]
    M, HW = N * H * W, H * W
    Kc = fp.cout
    act = lp.act
    if dl_padded is not None:
        dl = dl_padded
    else:
        dl = ops.zeros(N, pad8(Kc), dtype=T, device=dev)
        dl[:, :Kc] = dlogits
    # classifier
    ops.gemm_tn(dl, Kc, sv["pooled"], fp.cin, fp.W_grad, fp.cin, 1, N)
    if fp.bias_grad is not None:
        ops.colsum(dl, fp.bias_grad, N, Kc)
    dpooled = torch.empty(N, fp.cin, dtype=T, device=dev)
    ops.gemm_nt(dl, fp.WT_pack, dpooled, N, fp.cin, Kc)
    # dropout + average pool + ReLU backward, with the last BN's backward statistics
    L, b = sv["L"], sv["b"]
    gL = torch.empty(M, lp.cout, dtype=T, device=dev)
    st2 = _stats(lp.cout, dev, lp.bn["mgr"])
    ops.pool_act_bwd(dpooled, sv["keep"], sv["p"], L, b.scale, b.shift, int(act), gL, st2.t, N, HW, lp.cout, stat_rows=st2.rows)
    c1, c2, c3 = bn_backward_coeffs(lp.bn, b, st2, M, dev)
    ops.gemm_tn(sv["a"], lp.cin, gL, lp.cout, lp.W_grad, 1, lp.cin, M, v_mode=PRO_BNBWD, v2=L, vc1=c1, vc2=c2, vc3=c3)
    Gx = torch.empty(M, lp.cin, dtype=T, device=dev)
    ops.gemm_nt(gL, lp.WT_pack, Gx, M, lp.cin, lp.cout, a_mode=PRO_BNBWD, a2=L, ac1=c1, ac2=c2, ac3=c3)
    _stap(lp, dpooled=dpooled, gL=gL, c=(c1, c2, c3), gx=Gx)
    return Gx


class TailFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, lp, fp, drop_p, training, seed, step_ptr):
        logits, sv = tail_forward(lp, fp, x, drop_p, training, seed, step_ptr, True)
        ctx.lp, ctx.fp, ctx.sv = lp, fp, sv
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        lp, fp, sv = ctx.lp, ctx.fp, ctx.sv
        gx = tail_backward(lp, fp, sv, dlogits)
        ctx.sv = None
        lp.mgr.grad_done(fp)
        lp.mgr.grad_done(lp)
        N, H, W = sv["dims"]
        return to_4d(gx, N, H, W, lp.cin), None, None, None, None, None, None, None


def run_tail(lp, fp, x, anchor, drop_p, training, seed, step_ptr):
    if torch.is_grad_enabled() and (x.requires_grad or anchor.requires_grad):
        return TailFunction.apply(x, anchor, lp, fp, drop_p, training, seed, step_ptr)
    logits, _ = tail_forward(lp, fp, x, drop_p, training, seed, step_ptr, False)
    return logits


class TailLossFunction(torch.autograd.Function):
    """Tail (last 1x1 ConvBNReLU -> pool -> dropout -> classifier) + label-smoothed cross entropy, mean over the batch, as ONE
    autograd node for engine.TrainStep: the loss kernel writes d(mean loss)/d(logits) directly in the compute dtype with zeroed
    padding, so no ATen op (mean, its backward, dtype / padding copies) is left between the HIP launches
    (train.py:176-180 `loss = forward_loss(...)`, utils/optim.py:199-207, common.py:67-80).
    Returns the scalar mean loss; loss_vec / topk receive the per-sample losses and the top-1 / top-5 hit counts."""

    @staticmethod
    def forward(ctx, x, anchor, lp, fp, drop_p, training, seed, step_ptr, target, eps, loss_vec, topk, loss_out):
        logits, sv = tail_forward(lp, fp, x, drop_p, training, seed, step_ptr, True)
        B, K = logits.shape
        T = lp.mgr.compute_dtype
        dl = torch.empty(B, pad8(K), dtype=T, device=x.device)
        ops.ce_smooth(logits, target, eps, B, K, loss_vec, dl, 1.0, topk)   # gscale 1: dl = d(mean loss)/d(logits)
        ops.vec_sum(loss_vec, B, 1.0 / B, loss_out)
        ctx.lp, ctx.fp, ctx.sv, ctx.dl = lp, fp, sv, dl
        ctx.logits = logits
        return loss_out[0]

    @staticmethod
    def backward(ctx, gout):
        # contract: the caller differentiates the loss itself (d total / d loss = 1, as train.py:181 `loss.backward()` does);
        # gout is therefore not multiplied in (that would be an elementwise launch per step for a factor of one).  A caller that
        # scales the returned loss would silently get unscaled gradients: ATOMNAS_CHECK_LOSS_SEED=1 verifies the seed (one host
        # synchronisation per step, so not in the default path and never under graph capture).
        if _CHECK_LOSS_SEED and not torch.cuda.is_current_stream_capturing() and float(gout) != 1.0:
            raise RuntimeError("TailLossFunction differentiates d(loss)/d(loss) = 1 only; got a seed of %r (scale the learning "
                               "rate, or use CrossEntropyLabelSmooth for a loss that is combined with other terms)" % float(gout))
        lp, fp, sv = ctx.lp, ctx.fp, ctx.sv
        gx = tail_backward(lp, fp, sv, None, dl_padded=ctx.dl)
        ctx.sv = ctx.dl = None
        lp.mgr.grad_done(fp)
        lp.mgr.grad_done(lp)
        N, H, W = sv["dims"]
        return (to_4d(gx, N, H, W, lp.cin),) + (None,) * 12


# ---------------------------------------------------------------------------------------------- loss
class CESmoothFunction(torch.autograd.Function):
    """Per-sample label-smoothed cross entropy; also leaves top-1 / top-5 hit counts in `topk` (int32[2], accumulated)."""

    @staticmethod
    def forward(ctx, logits, target, eps, topk):
        if not logits.is_cuda:
            raise ops._lib.AtomnasHipError("CrossEntropyLabelSmooth runs on the GPU only")
        B, K = logits.shape
        lg = logits.float()
        if lg.stride(1) != 1:
            lg = lg.contiguous()
        loss = torch.empty(B, dtype=torch.float32, device=logits.device)
        dl = torch.empty(B, K, dtype=torch.float32, device=logits.device)
        # gscale = B: dl holds d(loss_i)/d(logits_i) (the 1/B of a mean reduction comes in through grad_output)
        ops.ce_smooth(lg, target, eps, B, K, loss, dl, float(B), topk)
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (dl,) = ctx.saved_tensors
        return dl * gout.unsqueeze(1), None, None, None
