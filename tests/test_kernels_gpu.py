"""Kernel-level parity (GPU): every C-ABI entry point against a float64 torch restatement of the same op.

Inputs are first rounded to the storage dtype, the reference computes in float64, and the kernel result must agree to the
tolerance of one output rounding (bf16) or accumulation order (fp32).  Integer results (masks, indices, counts) are exact.
"""
import itertools
import os
import sys

import pytest
import torch
import torch.nn.functional as F

from kutil import assert_close, cvec, from_act, pad8, rounded, to_act, tol

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def _ops():
    from atomnas_amd import ops
    return ops


def fresh(M, C, dtype):
    """output buffer as the allocator hands it out: padding channels zero (kernels never write non-zero there); the valid
    region is poisoned so that unwritten elements are caught"""
    b = torch.zeros(M, pad8(C), dtype=dtype, device="cuda")
    b[:, :C] = 7.0
    return b


def poisoned_stats(rows, C):
    """Partial-row statistics buffer as the contract allows it to arrive: uninitialised (here NaN).  The producer must write
    every row of its channel range; the rows are summed by the consumer (here: the test) in any order."""
    return torch.full((rows, 2, C), float("nan"), dtype=torch.float32, device="cuda")


def taps(w):  # [C,1,k,k] -> [k*k][pad8(C)] fp32 on GPU
    C, _, k, _ = w.shape
    t = torch.zeros(k * k, pad8(C), dtype=torch.float32, device="cuda")
    t[:, :C] = w.reshape(C, k * k).t().float().cuda()
    return t


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,stride", list(itertools.product([3, 5, 7], [1, 2])))
@pytest.mark.parametrize("N,C,H,W", [(2, 1, 7, 7), (3, 13, 15, 15), (2, 32, 8, 14), (1, 70, 28, 28), (2, 24, 44, 37)])
def test_dwconv_fwd(gpu_lib, dtype, k, stride, N, C, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(1000 * k + 10 * stride + C)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    for fuse in (False, True):
        xb = to_act(x, dtype)
        xr = rounded(x, dtype)
        xa = torch.relu(xr * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)) if fuse else xr
        yref = F.conv2d(xa, w.double(), None, stride, (k - 1) // 2, 1, C)
        Ho, Wo = yref.shape[2:]
        yb = fresh(N * Ho * Wo, C, dtype)
        stats = poisoned_stats(48 if k == 5 else 64, C)
        ops.dwconv_fwd(xb, cvec(sc) if fuse else None, cvec(sh) if fuse else None, fuse, taps(w), yb, stats, C, N, H, W, C, k, stride)
        torch.cuda.synchronize()
        y = from_act(yb, N, Ho, Wo, C)
        assert_close("y", y, yref, **tol(dtype))
        assert float(yb[:, C:].abs().max() if pad8(C) > C else 0) == 0.0
        stats = stats.sum(0)  # partial rows
        assert_close("sum", stats[0], y.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
        assert_close("sumsq", stats[1], (y * y).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,stride", list(itertools.product([3, 5, 7], [1, 2])))
@pytest.mark.parametrize("N,C,H,W", [(2, 1, 7, 7), (3, 13, 15, 15), (2, 32, 8, 14), (1, 70, 28, 28), (2, 24, 44, 37)])
def test_dwconv_bwd(gpu_lib, dtype, k, stride, N, C, H, W):
    ops = _ops()
    g = torch.Generator().manual_seed(2000 * k + 10 * stride + C)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    P = (k - 1) // 2
    Ho, Wo = (H + 2 * P - k) // stride + 1, (W + 2 * P - k) // stride + 1
    gup = torch.randn(N, C, Ho, Wo, generator=g)
    yraw = torch.randn(N, C, Ho, Wo, generator=g)
    c1 = torch.rand(C, generator=g) + 0.5
    c2 = torch.randn(C, generator=g) * 0.1
    c3 = torch.randn(C, generator=g) * 0.1
    for fuse in (False, True):
        v = lambda t: t.double().view(1, -1, 1, 1)
        xr = rounded(x, dtype).requires_grad_(True)
        pre = xr * v(sc) + v(sh) if fuse else xr
        xa = torch.relu(pre) if fuse else pre
        wd = w.double().requires_grad_(True)
        y = F.conv2d(xa, wd, None, stride, P, 1, C)
        dy = v(c1) * rounded(gup, dtype) + v(c2) * rounded(yraw, dtype) + v(c3) if fuse else rounded(gup, dtype)
        xa.retain_grad()
        (y * dy).sum().backward()
        href = xa.grad * (pre > 0).double() if fuse else xa.grad
        hb = fresh(N * H * W, C, dtype)
        dw = torch.zeros(C, k * k, dtype=torch.float32, device="cuda")
        stats = poisoned_stats(48 if k == 5 else 64, C)
        ops.dwconv_bwd(to_act(gup, dtype), to_act(yraw, dtype) if fuse else None, cvec(c1) if fuse else None,
                       cvec(c2) if fuse else None, cvec(c3) if fuse else None, to_act(x, dtype), cvec(sc) if fuse else None,
                       cvec(sh) if fuse else None, fuse, taps(w), hb, dw, stats, C, N, H, W, C, k, stride)
        torch.cuda.synchronize()
        h = from_act(hb, N, H, W, C)
        t = tol(dtype)
        assert_close("h", h, href, t["rtol"], t["atol"] * 4)
        assert_close("dw", dw.reshape(C, 1, k, k), wd.grad, rtol=2e-3, atol=2e-3 * float(wd.grad.abs().max()))
        stats = stats.sum(0)
        assert_close("sum_h", stats[0], h.sum((0, 2, 3)), rtol=1e-4, atol=2e-3)
        assert_close("sum_hx", stats[1], (h * rounded(x, dtype)).sum((0, 2, 3)), rtol=1e-4, atol=2e-3)


def pack_w(w, dtype, transposed=False):
    """[N,K] fp32 -> packed [pad64(N)][pad32(K)] storage dtype (or the transpose)"""
    if transposed:
        w = w.t()
    n, k = w.shape
    buf = torch.zeros((n + 63) // 64 * 64, (k + 31) // 32 * 32, dtype=dtype, device="cuda")
    buf[:n, :k] = w.to(dtype).cuda()
    return buf


GEMM_SHAPES = [(100, 24, 16), (1000, 432, 24), (333, 40, 139), (64, 16, 432), (257, 100, 40), (50, 7, 3), (4096, 320, 1152),
               # column-stationary kernel (M >= 1024, K <= 192, N >= 2K): 1, 2, 3 and 6 k-steps, ragged N and M
               (2000, 432, 24), (1500, 288, 16), (1100, 720, 40), (1031, 203, 96), (1200, 400, 192), (2048, 139, 24),
               (20000, 1440, 80),
               # weight-shared kernel (bf16, K > 192, M >= 4096): one / two 64-channel chunks, channel groups, ragged M, K and N,
               # and enough rows for two subtiles per wave
               (4100, 24, 432), (5000, 80, 1440), (4097, 139, 203), (4200, 192, 3456), (140000, 40, 720), (70000, 96, 250),
               # narrow kernel (bf16, N <= 64, K <= 64, M >= 65536): the stem and the first block; one / two k-steps, ragged everything
               (70001, 32, 27), (66000, 16, 32), (65599, 32, 16), (80000, 24, 40), (65536, 64, 64)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
@pytest.mark.parametrize("variant", ["plain_stats", "bnrelu_stats", "bnbwd_mask_statz", "bnbwd_add_statz", "bias_f32", "plain_mask_statz"])
def test_gemm_nt(gpu_lib, dtype, M, N, K, variant):
    ops = _ops()
    g = torch.Generator().manual_seed(M + 7 * N + 13 * K)
    r = lambda *s: torch.randn(*s, generator=g)
    A, A2, W = r(M, K), r(M, K), r(N, K) / K ** 0.5
    c1, c2, c3 = torch.rand(K, generator=g) + 0.5, r(K) * 0.2, r(K) * 0.2
    Z, ADD = r(M, N), r(M, N)
    zs, zh = torch.rand(N, generator=g) + 0.5, r(N) * 0.3
    bias = r(N)
    rd = lambda t: t.to(dtype).double()
    act2d = lambda t, c: torch.cat([t.to(dtype), torch.zeros(t.shape[0], pad8(c) - c, dtype=dtype)], 1).cuda()
    Wd = rd(W)
    kw = {}
    out_dtype = dtype
    if variant == "plain_stats":
        Aeff = rd(A)
        kw = dict(stat_mode=ops.STAT_SQ)
    elif variant == "bnrelu_stats":
        Aeff = torch.relu(rd(A) * c1.double() + c2.double())
        kw = dict(a_mode=ops.PRO_BNRELU, ac1=cvec(c1), ac2=cvec(c2), a_relu=True, stat_mode=ops.STAT_SQ)
    elif variant == "plain_mask_statz":   # the streaming kernel's masked form: no prologue (the operand is atomnas_bnbwd_apply's output)
        Aeff = rd(A)
        kw = dict(stat_mode=ops.STAT_Z)
    else:
        Aeff = rd(A) if variant == "bias_f32" else c1.double() * rd(A) + c2.double() * rd(A2) + c3.double()
        if variant != "bias_f32":
            kw = dict(a_mode=ops.PRO_BNBWD, a2=act2d(A2, K), ac1=cvec(c1), ac2=cvec(c2), ac3=cvec(c3), stat_mode=ops.STAT_Z)
    if dtype == torch.bfloat16:
        Aeff = Aeff.to(torch.bfloat16).double()  # the prologue result is rounded to bf16 before the MFMA
    Cref = Aeff @ Wd.t()
    if variant in ("bnbwd_mask_statz", "plain_mask_statz"):
        Cref = Cref * ((rd(Z) * zs.double() + zh.double()) > 0)
        kw.update(z=act2d(Z, N), zscale=cvec(zs), zshift=cvec(zh), mask=True)
    elif variant == "bnbwd_add_statz":
        Cref = Cref + rd(ADD)
        kw.update(z=act2d(Z, N), add=act2d(ADD, N))
    elif variant == "bias_f32":
        Cref = Cref + bias.double()
        kw.update(bias=cvec(bias))
        out_dtype = torch.float32
    Cb = fresh(M, N, out_dtype)
    stats = poisoned_stats(40 if M % 2 else 128, N) if "stat_mode" in kw else None
    ops.gemm_nt(act2d(A, K), pack_w(W, dtype), Cb, M, N, K, stats=stats, **kw)
    torch.cuda.synchronize()
    Cg = Cb[:, :N].double().cpu()
    scale = float(Cref.abs().max())
    t = tol(out_dtype if variant == "bias_f32" and dtype == torch.float32 else dtype)
    assert_close("C", Cg, Cref, t["rtol"], t["atol"] * max(1.0, scale))
    if pad8(N) > N:
        assert float(Cb[:, N:].abs().max()) == 0.0
    if stats is not None:
        stats = stats.sum(0)
        s1 = Cg.sum(0)
        s2 = (Cg * Cg).sum(0) if kw["stat_mode"] == ops.STAT_SQ else (Cg * rd(Z)).sum(0)
        assert_close("s1", stats[0], s1, rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale))
        assert_close("s2", stats[1], s2, rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale) ** 2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,NU,NV", [(100, 24, 16), (1000, 24, 432), (3000, 40, 139), (700, 320, 1152), (257, 700, 40), (50, 3, 7),
                                     # 128-column V tiles: 4 / 6 accumulator tiles on 64-row slabs, two U tiles of 160 on 128-row slabs
                                     (2000, 96, 1728), (1500, 40, 720), (3100, 80, 300)])
@pytest.mark.parametrize("variant", ["none_none", "none_bnbwd", "bnbwd_bnrelu", "none_bnrelu"])
def test_gemm_tn(gpu_lib, dtype, M, NU, NV, variant):
    ops = _ops()
    g = torch.Generator().manual_seed(M + 7 * NU + 13 * NV)
    r = lambda *s: torch.randn(*s, generator=g)
    U, U2, V, V2 = r(M, NU), r(M, NU), r(M, NV), r(M, NV)
    uc = [torch.rand(NU, generator=g) + 0.5, r(NU) * 0.2, r(NU) * 0.2]
    vc = [torch.rand(NV, generator=g) + 0.5, r(NV) * 0.2, r(NV) * 0.2]
    rd = lambda t: t.to(dtype).double()
    act2d = lambda t, c: torch.cat([t.to(dtype), torch.zeros(t.shape[0], pad8(c) - c, dtype=dtype)], 1).cuda()
    kw = {}
    Ue, Ve = rd(U), rd(V)
    if variant == "none_bnbwd":
        Ve = vc[0].double() * rd(V) + vc[1].double() * rd(V2) + vc[2].double()
        kw = dict(v_mode=ops.PRO_BNBWD, v2=act2d(V2, NV), vc1=cvec(vc[0]), vc2=cvec(vc[1]), vc3=cvec(vc[2]))
    elif variant == "bnbwd_bnrelu":
        Ue = uc[0].double() * rd(U) + uc[1].double() * rd(U2) + uc[2].double()
        Ve = torch.relu(rd(V) * vc[0].double() + vc[1].double())
        kw = dict(u_mode=ops.PRO_BNBWD, u2=act2d(U2, NU), uc1=cvec(uc[0]), uc2=cvec(uc[1]), uc3=cvec(uc[2]),
                  v_mode=ops.PRO_BNRELU, vc1=cvec(vc[0]), vc2=cvec(vc[1]), v_relu=True)
    elif variant == "none_bnrelu":   # U = atomnas_bnbwd_apply's output
        Ve = torch.relu(rd(V) * vc[0].double() + vc[1].double())
        kw = dict(v_mode=ops.PRO_BNRELU, vc1=cvec(vc[0]), vc2=cvec(vc[1]), v_relu=True)
    if dtype == torch.bfloat16:
        Ue, Ve = Ue.to(torch.bfloat16).double(), Ve.to(torch.bfloat16).double()
    ref = Ue.t() @ Ve  # [NU, NV]
    # write the transposed layout out[j][i] (si = 1, sj = NU) as well as the natural one
    for si, sj, view in ((NV, 1, lambda o: o), (1, NU, lambda o: o.t())):
        out = torch.zeros(NU, NV, dtype=torch.float32, device="cuda") if si == NV else torch.zeros(NV, NU, dtype=torch.float32, device="cuda")
        ops.gemm_tn(act2d(U, NU), NU, act2d(V, NV), NV, out, si, sj, M, **kw)
        torch.cuda.synchronize()
        assert_close("out", view(out), ref, rtol=2e-3 if dtype == torch.bfloat16 else 2e-4, atol=2e-3 * float(ref.abs().max()))


@pytest.mark.parametrize("slab", [False, True])
@pytest.mark.parametrize("variant", ["none_none", "none_bnrelu"])
@pytest.mark.parametrize("M,NU,NV", [(12544, 192, 3456), (5001, 80, 1440), (6000, 320, 1280), (2500, 96, 1000), (1024, 40, 720), (3333, 36, 257)])
def test_gemm_tn_dma_form(gpu_lib, M, NU, NV, variant, slab):
    """k_gemm_tn3 (round 4): the single-stream weight gradients of the late stages -- LDS-DMA copies, transposing fragment reads, V's
    prologue on the fragments.  The supernet's 7 x 7 / 14 x 14 / 28 x 28 shapes, two U tiles (320), ragged NU / NV / M (a last stage of
    fewer than 32 rows, V tiles beyond NV), plain and slab-major V."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(M + 7 * NU + 13 * NV)
    r = lambda *s: torch.randn(*s, generator=g)
    U, V = r(M, NU), r(M, NV)
    vc = [torch.rand(NV, generator=g) + 0.5, r(NV) * 0.2]
    rd = lambda t: t.to(dtype).double()
    act2d = lambda t, c: torch.cat([t.to(dtype), torch.zeros(t.shape[0], pad8(c) - c, dtype=dtype)], 1).cuda()
    Ue, Ve = rd(U), rd(V)
    kw = {}
    if variant == "none_bnrelu":
        Ve = torch.relu(rd(V) * vc[0].double() + vc[1].double()).to(dtype).double()
        kw = dict(v_mode=ops.PRO_BNRELU, vc1=cvec(vc[0]), vc2=cvec(vc[1]), v_relu=True)
    ref = Ue.t() @ Ve
    Vb = ops.Slab.from_plain(act2d(V, NV), NV) if slab else act2d(V, NV)
    out = torch.zeros(NU, NV, dtype=torch.float32, device="cuda")
    ops.gemm_tn(act2d(U, NU), NU, Vb, NV, out, NV, 1, M, **kw)
    torch.cuda.synchronize()
    assert_close("out", out, ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    # bit-reproducible (fixed-order partial sums) and accumulating (+=)
    out2 = out.clone()
    ops.gemm_tn(act2d(U, NU), NU, Vb, NV, out2, NV, 1, M, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out2, 2 * out)


# ---------------------------------------------------------------------------------------------- channel-pair-per-wave depthwise kernels
# csrc/dwconv_cw.hip: stride 1, slab-major tensors, width a multiple of 7 -- the instances the hidden tensors of the expanding
# blocks take.  Cases: several whole images per tile (7x7, 14x14; the last tile partly empty), one whole image per tile (30x14),
# row-ring tiles (28x28, 56x56, a ragged last tile at 44x28), 3 strips per row (pitch rule for non-powers of two), channel
# counts that leave the last slab / the last 8-channel group partly or wholly empty.
CW_SHAPES = [(10, 16, 7, 7), (3, 48, 14, 14), (2, 24, 56, 56), (5, 40, 28, 28), (2, 13, 21, 21), (1, 70, 28, 28), (2, 32, 30, 14),
             (2, 16, 44, 28), (3, 5, 14, 14)]


def _cw_supported(N, H, W, C, k, dtype, direction, stride=1):
    from atomnas_amd import _lib
    return bool(_lib.load().atomnas_dwconv_cw_supported(N, H, W, C, k, stride, 0 if dtype == torch.float32 else 1, direction))


# stride 2 (csrc/dwconv_cw.hip k_dwb_cw2): lanes on the output grid -- whole output images per tile (14x14 -> 7x7, 28x28 -> 14x14),
# row-ring tiles (112, 56, a ragged last tile at 88x56), 3 strips per row, ragged channel counts
CW2_SHAPES = [(10, 16, 14, 14), (3, 48, 28, 28), (2, 24, 112, 112), (5, 40, 56, 56), (2, 13, 42, 42), (1, 70, 56, 56), (2, 16, 88, 56),
              (3, 5, 28, 28)]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [3, 5, 7])
@pytest.mark.parametrize("N,C,H,W", CW_SHAPES)
def test_dwconv_fwd_cw(gpu_lib, dtype, k, N, C, H, W):
    ops = _ops()
    from atomnas_amd.ops import Slab
    assert _cw_supported(N, H, W, C, k, dtype, 0) or os.environ.get("ATOMNAS_DW_CW") is not None
    g = torch.Generator().manual_seed(3000 * k + C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    for fuse in (False, True):
        xr = rounded(x, dtype)
        xa = torch.relu(xr * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)) if fuse else xr
        yref = F.conv2d(xa, w.double(), None, 1, (k - 1) // 2, 1, C)
        ys = Slab.from_plain(fresh(N * H * W, C, dtype), C)   # poisoned valid region: unwritten elements are caught
        stats = poisoned_stats(48 if k == 5 else 64, C)
        ops.dwconv_fwd(_slab(to_act(x, dtype), C), cvec(sc) if fuse else None, cvec(sh) if fuse else None, fuse, taps(w), ys, stats, C, N,
                       H, W, C, k, 1)
        torch.cuda.synchronize()
        yp = ys.to_plain()
        y = from_act(yp, N, H, W, C)
        assert_close("y", y, yref, **tol(dtype))
        assert float(yp[:, C:].float().abs().max() if yp.shape[1] > C else 0) == 0.0
        stats = stats.sum(0)
        assert_close("sum", stats[0], y.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
        assert_close("sumsq", stats[1], (y * y).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k", [3, 5, 7])
@pytest.mark.parametrize("stride,N,C,H,W", [(1,) + s for s in CW_SHAPES] + [(2,) + s for s in CW2_SHAPES])
def test_dwconv_bwd_cw(gpu_lib, dtype, k, stride, N, C, H, W):
    ops = _ops()
    from atomnas_amd.ops import Slab
    assert _cw_supported(N, H, W, C, k, dtype, 1, stride) or os.environ.get("ATOMNAS_DW_CW") is not None
    g = torch.Generator().manual_seed(4000 * k + C + H)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    P = (k - 1) // 2
    Ho, Wo = (H + 2 * P - k) // stride + 1, (W + 2 * P - k) // stride + 1
    gup = torch.randn(N, C, Ho, Wo, generator=g)
    yraw = torch.randn(N, C, Ho, Wo, generator=g)
    c1 = torch.rand(C, generator=g) + 0.5
    c2 = torch.randn(C, generator=g) * 0.1
    c3 = torch.randn(C, generator=g) * 0.1
    for fuse in (False, True):
        v = lambda t: t.double().view(1, -1, 1, 1)
        xr = rounded(x, dtype).requires_grad_(True)
        pre = xr * v(sc) + v(sh) if fuse else xr
        xa = torch.relu(pre) if fuse else pre
        wd = w.double().requires_grad_(True)
        y = F.conv2d(xa, wd, None, stride, P, 1, C)
        dy = v(c1) * rounded(gup, dtype) + v(c2) * rounded(yraw, dtype) + v(c3) if fuse else rounded(gup, dtype)
        xa.retain_grad()
        (y * dy).sum().backward()
        href = xa.grad * (pre > 0).double() if fuse else xa.grad
        hs = Slab.from_plain(fresh(N * H * W, C, dtype), C)
        dw = torch.zeros(C, k * k, dtype=torch.float32, device="cuda")
        stats = poisoned_stats(48 if k == 5 else 64, C)
        ops.dwconv_bwd(_slab(to_act(gup, dtype), C), _slab(to_act(yraw, dtype), C) if fuse else None, cvec(c1) if fuse else None,
                       cvec(c2) if fuse else None, cvec(c3) if fuse else None, _slab(to_act(x, dtype), C), cvec(sc) if fuse else None,
                       cvec(sh) if fuse else None, fuse, taps(w), hs, dw, stats, C, N, H, W, C, k, stride)
        torch.cuda.synchronize()
        hp = hs.to_plain()
        h = from_act(hp, N, H, W, C)
        t = tol(dtype)
        assert_close("h", h, href, t["rtol"], t["atol"] * 4)
        assert float(hp[:, C:].float().abs().max() if hp.shape[1] > C else 0) == 0.0
        assert_close("dw", dw.reshape(C, 1, k, k), wd.grad, rtol=2e-3, atol=2e-3 * float(wd.grad.abs().max()))
        stats = stats.sum(0)
        assert_close("sum_h", stats[0], h.sum((0, 2, 3)), rtol=1e-4, atol=2e-3)
        assert_close("sum_hx", stats[1], (h * rounded(x, dtype)).sum((0, 2, 3)), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("H,C,k", [(56, 144, 7), (28, 240, 5), (14, 480, 3), (14, 576, 7), (7, 1152, 3), (7, 1152, 5), (7, 1152, 7)])
def test_dwconv_bench_shapes_against_torch(gpu_lib, H, C, k):
    """The depthwise entry points at the bench's own sizes (batch 256: several tiles per worker, partial last tiles -- the small cases
    above give every worker one tile), bf16 slab-major tensors, against torch's convolution in fp32 on the GPU: forward output and
    statistics, input gradient, weight gradient, backward statistics.  bf16 shapes of stride 1 run csrc/dwconv_mm.hip (tap arithmetic
    on the matrix cores: fp16 operands forward, bf16 operands backward), so the bounds are those of one bf16 output rounding plus the
    operand roundings stated there."""
    ops = _ops()
    from atomnas_amd.ops import Slab
    N, s = 256, 1
    P = (k - 1) // 2
    g = torch.Generator(device="cuda").manual_seed(H * 1000 + C + k)
    rn = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
    x2 = rn(N * H * H, C).bfloat16()
    g2 = (rn(N * H * H, C) * 1e-3).bfloat16()
    w = rn(C, 1, k, k) * 0.3
    tp = w.reshape(C, k * k).t().contiguous()
    sc, sh = torch.rand(C, device="cuda", generator=g) + 0.5, rn(C) * 0.3
    c1, c2, c3 = torch.rand(C, device="cuda", generator=g) + 0.5, rn(C) * 0.1, rn(C) * 1e-4
    rows = ops.stat_rows_for(C)
    nchw = lambda sl: sl.to_plain()[:, :C].float().reshape(N, H, H, C).permute(0, 3, 1, 2)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    xs, gs = Slab.from_plain(x2), Slab.from_plain(g2)
    ys = Slab(N * H * H, C, torch.bfloat16, "cuda")
    ys.t.fill_(float("nan"))
    st = torch.full((rows, 2, C), float("nan"), device="cuda")
    ops.dwconv_fwd(xs, sc, sh, True, tp, ys, st, C, N, H, H, C, k, s, stat_rows=rows)
    x4 = x2.float().reshape(N, H, H, C).permute(0, 3, 1, 2).requires_grad_(True)
    xa = torch.relu(x4 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    xa.retain_grad()
    wr = w.clone().requires_grad_(True)
    yref = F.conv2d(xa, wr, None, s, P, 1, C)
    y = nchw(ys)
    assert torch.isfinite(y).all()
    assert rel(y, yref.detach()) < 2.5e-3          # one bf16 rounding is 1.7e-3
    assert float((y - yref.detach()).abs().max()) < 2e-2 * float(yref.abs().max())
    ssum = st.sum(0)
    assert torch.allclose(ssum[0], y.sum((0, 2, 3)), rtol=1e-4, atol=1.0) and torch.allclose(ssum[1], (y * y).sum((0, 2, 3)), rtol=1e-4, atol=1.0)
    dy = c1.view(1, -1, 1, 1) * g2.float().reshape(N, H, H, C).permute(0, 3, 1, 2) + c2.view(1, -1, 1, 1) * y + c3.view(1, -1, 1, 1)
    (yref * dy).sum().backward()
    href = xa.grad * (xa.detach() > 0).float()
    hs = Slab(N * H * H, C, torch.bfloat16, "cuda")
    hs.t.fill_(float("nan"))
    dw = torch.zeros(C * k * k, device="cuda")
    st2 = torch.full((rows, 2, C), float("nan"), device="cuda")
    ops.dwconv_bwd(gs, ys, c1, c2, c3, xs, sc, sh, True, tp, hs, dw, st2, C, N, H, H, C, k, s, stat_rows=rows)
    h = nchw(hs)
    assert torch.isfinite(h).all()
    assert rel(h, href) < 3.5e-3                   # bf16 output rounding + bf16 rounding of dY (matrix-core operand)
    assert rel(dw.view(C, k * k), wr.grad.view(C, k * k)) < 1e-3
    s2 = st2.sum(0)
    assert rel(s2[0], h.sum((0, 2, 3))) < 1e-4 and rel(s2[1], (h * x4.detach()).sum((0, 2, 3))) < 1e-4


@pytest.mark.parametrize("N,H,C,k", [(256, 112, 96, 7), (256, 112, 96, 3), (256, 56, 144, 5), (256, 28, 240, 7), (256, 14, 576, 5),
                                     (3, 112, 16, 5), (5, 56, 24, 7), (5, 112, 40, 3), (7, 28, 20, 5), (11, 14, 40, 7), (1, 14, 8, 5)])
def test_dwconv_stride2_forward_on_the_matrix_cores(gpu_lib, N, H, C, k):
    """csrc/dwconv_mm2.hip: the stride-2 depthwise forward (bf16 slab-major) as a Toeplitz product on the matrix cores, at the bench's
    sizes (many bands per worker: row ring, image starts inside a worker's walk) and at small batches (one band per worker: every
    worker but the first of an image starts inside it; partial last image groups; channel counts that are not multiples of 16),
    against torch's convolution in fp32: output (one bf16 rounding + the fp16 operand roundings) and the statistics of the stored
    values.  The library must say that it takes that kernel for the shape (otherwise this test checks nothing new)."""
    ops = _ops()
    from atomnas_amd.ops import Slab
    s, P, Ho = 2, (k - 1) // 2, H // 2
    assert gpu_lib.atomnas_dwconv_mm_supported(N, H, H, C, k, 2, 1, 0) == 1
    g = torch.Generator(device="cuda").manual_seed(N * 7 + H * 1000 + C + k)
    rn = lambda *sh: torch.randn(*sh, device="cuda", generator=g)
    x2 = rn(N * H * H, C).bfloat16()
    w = rn(C, 1, k, k) * 0.3
    tp = w.reshape(C, k * k).t().contiguous()
    sc, sh = torch.rand(C, device="cuda", generator=g) + 0.5, rn(C) * 0.3
    rows = ops.stat_rows_for(C)
    acts = {0: lambda t: t, 1: torch.relu, 2: lambda t: t.clamp(0, 6), 3: lambda t: t * torch.sigmoid(t)}   # the C ABI's activation codes
    for act in ((1, 2, 3, 0) if N <= 11 or k == 5 else (2,)):
        xs = Slab.from_plain(x2)
        ys = Slab(N * Ho * Ho, C, torch.bfloat16, "cuda")
        ys.t.fill_(float("nan"))
        st = torch.full((rows, 2, C), float("nan"), device="cuda")
        ops.dwconv_fwd(xs, sc, sh, act, tp, ys, st, C, N, H, H, C, k, s, stat_rows=rows)
        x4 = x2.float().reshape(N, H, H, C).permute(0, 3, 1, 2)
        xa = x4 * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
        yref = F.conv2d(acts[act](xa), w, None, s, P, 1, C)
        yp = ys.to_plain()
        cpad = (C + 7) // 8 * 8
        assert float(yp[:, C:cpad].float().abs().max() if cpad > C else 0) == 0.0      # padding channels of the last 8-channel group are zero
        y = yp[:, :C].float().reshape(N, Ho, Ho, C).permute(0, 3, 1, 2)
        assert torch.isfinite(y).all()
        assert float((y - yref).norm() / yref.norm()) < 2.5e-3          # one bf16 rounding is 1.7e-3
        assert float((y - yref).abs().max()) < 2e-2 * float(yref.abs().max())
        ssum = st.sum(0)
        assert torch.allclose(ssum[0], y.sum((0, 2, 3)), rtol=1e-4, atol=1.0) and torch.allclose(ssum[1], (y * y).sum((0, 2, 3)), rtol=1e-4, atol=1.0)


def test_dwconv_long_tile_walks():
    """The depthwise kernels keep the halo rows of the tile above in their LDS ring when a workgroup walks down a column of
    tiles.  With the small tensors of the tests every workgroup normally gets a single tile, so the same cases are re-run
    with the number of workgroups per slab capped (read once per process -> separate interpreter)."""
    import subprocess
    env = dict(os.environ, ATOMNAS_DW_MAX_WORKERS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-k", "test_dwconv_fwd or test_dwconv_bwd"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]


# ---------------------------------------------------------------------------------------------- slab-major layout
def _slab(t2d, C):
    from atomnas_amd.ops import Slab
    return Slab.from_plain(t2d, C)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32], ids=["bf16", "fp32"])
@pytest.mark.parametrize("k,stride", [(3, 1), (5, 2), (7, 1), (7, 2)])
@pytest.mark.parametrize("N,C,H,W", [(3, 48, 15, 15), (2, 144, 28, 28), (2, 16, 44, 37)])
def test_dwconv_slab_layout_is_bit_identical_to_plain(gpu_lib, dtype, k, stride, N, C, H, W):
    """The hidden tensors of a block are slab-major ([C/16][M][16], include/atomnas_hip.h).  The layout changes addresses only:
    same arithmetic per element -> every output bit equals the plain-layout result.  The per-channel sums (statistics, weight
    gradient) group their partials by worker, and the number of workers depends on the layout (plain: whole groups per XCD), so
    those agree to summation-order rounding."""
    ops = _ops()
    from atomnas_amd.ops import Slab
    if stride == 1 and _cw_supported(N, H, W, C, k, dtype, 0):
        pytest.skip("slab-major stride-1 tensors of this shape run csrc/dwconv_cw.hip (other summation order): test_dwconv_*_cw")
    g = torch.Generator().manual_seed(k * 100 + C)
    P = (k - 1) // 2
    Ho, Wo = (H + 2 * P - k) // stride + 1, (W + 2 * P - k) // stride + 1
    x, gup, yraw = torch.randn(N, C, H, W, generator=g), torch.randn(N, C, Ho, Wo, generator=g), torch.randn(N, C, Ho, Wo, generator=g)
    w = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc, sh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    c1, c2, c3 = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    xb, gb, yb_ = to_act(x, dtype), to_act(gup, dtype), to_act(yraw, dtype)
    res = []
    for slab in (False, True):
        wrap = (lambda t: _slab(t, C)) if slab else (lambda t: t)
        y = Slab(N * Ho * Wo, C, dtype, "cuda", zero=True) if slab else fresh(N * Ho * Wo, C, dtype)
        st = poisoned_stats(64, C)
        ops.dwconv_fwd(wrap(xb), cvec(sc), cvec(sh), True, taps(w), y, st, C, N, H, W, C, k, stride)
        h = Slab(N * H * W, C, dtype, "cuda", zero=True) if slab else fresh(N * H * W, C, dtype)
        dw = torch.zeros(C, k * k, dtype=torch.float32, device="cuda")
        st2 = poisoned_stats(64, C)
        ops.dwconv_bwd(wrap(gb), wrap(yb_), cvec(c1), cvec(c2), cvec(c3), wrap(xb), cvec(sc), cvec(sh), True, taps(w), h, dw, st2, C,
                       N, H, W, C, k, stride)
        torch.cuda.synchronize()
        res.append((y.to_plain()[:, :C] if slab else y[:, :C], h.to_plain()[:, :C] if slab else h[:, :C],
                    st.view(64, 2, -1).sum(0), dw, st2.view(64, 2, -1).sum(0)))
    # stride 2: the slab-major backward of a supported shape runs the channel-pair kernel (csrc/dwconv_cw.hip, other FMA order), the
    # plain one the tile kernel: the input gradient then agrees to rounding of the bf16 result, not bit for bit
    cw_bwd = stride == 2 and _cw_supported(N, H, W, C, k, dtype, 1, stride=2)
    # ... and the slab-major bf16 forward of a supported stride-2 shape runs on the matrix cores (csrc/dwconv_mm2.hip: fp16 operands):
    # output and its statistics agree to the operand roundings
    mm_fwd = stride == 2 and dtype == torch.bfloat16 and gpu_lib.atomnas_dwconv_mm_supported(N, H, W, C, k, 2, 1, 0) == 1
    for i, (a, b) in enumerate(zip(*res)):
        if i == 0 and mm_fwd:
            assert torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2 * float(b.float().abs().max()))
            assert float((a != b).float().mean()) < 0.2
        elif i == 2 and mm_fwd:
            assert torch.allclose(a, b, rtol=2e-3, atol=2e-3 * float(b.abs().max())), (i, float((a - b).abs().max()))
        elif i == 1 and cw_bwd:
            if dtype == torch.bfloat16:
                assert torch.allclose(a.float(), b.float(), rtol=2e-2, atol=2e-2 * float(b.float().abs().max()))
                assert float((a != b).float().mean()) < 0.2
            else:   # fp32: the other FMA order shows in the last bits of many elements
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))
        elif i < 2:
            assert torch.equal(a, b)
        else:
            assert torch.allclose(a, b, rtol=2e-5, atol=2e-5 * float(b.abs().max())), (i, float((a - b).abs().max()))


@pytest.mark.parametrize("M,N,K", [(2000, 432, 24), (5000, 80, 1440), (333, 40, 139), (4100, 96, 576)])
def test_gemm_slab_layout_is_bit_identical_to_plain(gpu_lib, M, N, K):
    ops = _ops()
    from atomnas_amd.ops import Slab
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    r = lambda *s: torch.randn(*s, generator=g)
    act2d = lambda t, c: torch.cat([t.to(dtype), torch.zeros(t.shape[0], pad8(c) - c, dtype=dtype)], 1).cuda()
    A, A2, Z, W = act2d(r(M, K), K), act2d(r(M, K), K), act2d(r(M, N), N), r(N, K) / K ** 0.5
    c1, c2, c3 = cvec(torch.rand(K, generator=g) + 0.5), cvec(r(K) * 0.2), cvec(r(K) * 0.2)
    zs, zh = cvec(torch.rand(N, generator=g) + 0.5), cvec(r(N) * 0.3)
    res = []
    for slab in (False, True):
        wa = (lambda t: _slab(t, K)) if slab else (lambda t: t)
        wz = (lambda t: _slab(t, N)) if slab else (lambda t: t)
        C = Slab(M, N, dtype, "cuda", zero=True) if slab else fresh(M, N, dtype)
        st = poisoned_stats(64, N)
        ops.gemm_nt(wa(A), pack_w(W, dtype), C, M, N, K, a_mode=ops.PRO_BNBWD, a2=wa(A2), ac1=c1, ac2=c2, ac3=c3, z=wz(Z), zscale=zs,
                    zshift=zh, mask=True, stats=st, stat_mode=ops.STAT_Z)
        out = torch.zeros(N, K, dtype=torch.float32, device="cuda")
        ops.gemm_tn(wz(Z), N, wa(A), K, out, K, 1, M, v_mode=ops.PRO_BNBWD, v2=wa(A2), vc1=c1, vc2=c2, vc3=c3)
        out2 = torch.zeros(K, N, dtype=torch.float32, device="cuda")
        ops.gemm_tn(wa(A), K, wz(Z), N, out2, N, 1, M, u_mode=ops.PRO_BNBWD, u2=wa(A2), uc1=c1, uc2=c2, uc3=c3, v_mode=ops.PRO_BNRELU,
                    vc1=zs, vc2=zh, v_relu=True)
        torch.cuda.synchronize()
        res.append((C.to_plain()[:, :N] if slab else C[:, :N], st, out, out2))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(20000, 24, 432), (70001, 24, 432), (66000, 16, 288), (40000, 40, 432), (36000, 32, 304),
                                   (17000, 48, 120)])
def test_gemm_nt_streaming_wide_input(gpu_lib, M, N, K, act):
    """k_gemm_nt_sw (bf16, BNRELU prologue, slab-major A with K <= 448, N <= 48, M >= 16384): the projection of the early stages with
    and without the output statistics; several row blocks per workgroup, ragged M, K not a multiple of 64, all three activations;
    against fp64 and against the LDS-weights kernel on the plain layout."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K + act)
    r = lambda *s: torch.randn(*s, generator=g)
    A, W = r(M, K), r(N, K) / K ** 0.5
    c1, c2 = torch.rand(K, generator=g) + 0.5, r(K) * 0.3
    rd = lambda t: t.to(dtype).double()
    pre = rd(A) * c1.double() + c2.double()
    if act == 3:
        Aeff = pre * torch.sigmoid(pre)
    else:
        Aeff = pre.clamp(min=0.0, max=6.0 if act == 2 else float("inf"))
    Cref = Aeff.to(dtype).double() @ rd(W).t()
    scale = float(Cref.abs().max())
    Ap = torch.cat([A.to(dtype), torch.zeros(M, pad8(K) - K, dtype=dtype)], 1).cuda()
    Wp = pack_w(W, dtype)
    outs = []
    for slab, with_stats in ((True, True), (True, False), (False, True)):
        C = fresh(M, N, dtype)
        st = poisoned_stats(512, N) if with_stats else None
        ops.gemm_nt(_slab(Ap, K) if slab else Ap, Wp, C, M, N, K, a_mode=ops.PRO_BNRELU, ac1=cvec(c1), ac2=cvec(c2), a_relu=act, stats=st,
                    stat_mode=ops.STAT_SQ if with_stats else 0)
        torch.cuda.synchronize()
        Cg = C[:, :N].double().cpu()
        assert_close("C", Cg, Cref, tol(dtype)["rtol"], tol(dtype)["atol"] * max(1.0, scale))
        if with_stats:
            assert not torch.isnan(st).any()
            sm = st.sum(0)
            assert_close("s1", sm[0], Cg.sum(0), rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale))
            assert_close("s2", sm[1], (Cg * Cg).sum(0), rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale) ** 2)
        outs.append(C[:, :N].float())
    assert torch.equal(outs[0], outs[1])   # the statistics do not change the output
    # other kernel, other summation order: agreement to the rounding of the bf16 result
    assert torch.allclose(outs[0], outs[2], rtol=2e-2, atol=2e-2 * max(1.0, scale))


@pytest.mark.parametrize("act", [1, 3])
@pytest.mark.parametrize("M,N,K", [(12544, 192, 3456), (12544, 320, 3456), (50176, 80, 1440), (50176, 96, 1728),
                                   # ragged post-shrink widths (hidden width = sum of branch segments padded to 16), ragged M
                                   (12001, 192, 2608), (33333, 96, 1104), (9000, 40, 272), (50000, 88, 656)])
def test_gemm_nt_streaming_wide_input_late_stages(gpu_lib, M, N, K, act):
    """k_gemm_nt_swg (bf16, BNRELU prologue, slab-major A, 8192 <= M <= 100000, K >= 256: the projection forward of the 14x14 / 7x7
    stages with the weight chunks in the LDS-DMA queue) at the bench's shapes and at post-shrink widths: against fp64, with and without
    the statistics, and against the LDS-weights kernel on the plain layout -- the same k order of the MFMA accumulation, so the outputs
    are BIT-IDENTICAL (the slab / plain layouts of a hidden tensor must not change a result bit)."""
    ops = _ops()
    dtype = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(M + N + K + act)
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    A, W = r(M, K).to(dtype), r(N, K) / K ** 0.5
    c1, c2 = torch.rand(K, device="cuda", generator=g) + 0.5, r(K) * 0.3
    pre = A.double() * c1.double() + c2.double()
    Aeff = pre * torch.sigmoid(pre) if act == 3 else pre.clamp(min=0.0)
    Wp = pack_w(W.cpu(), dtype)
    Cref = Aeff.to(dtype).double() @ Wp[:N, :K].double().t()
    scale = float(Cref.abs().max())
    Ap = torch.cat([A, torch.zeros(M, pad8(K) - K, dtype=dtype, device="cuda")], 1)
    outs = []
    for slab, with_stats in ((True, True), (True, False), (False, True)):
        C = fresh(M, N, dtype)
        st = poisoned_stats(512, N) if with_stats else None
        ops.gemm_nt(_slab(Ap, K) if slab else Ap, Wp, C, M, N, K, a_mode=ops.PRO_BNRELU, ac1=cvec(c1), ac2=cvec(c2), a_relu=act, stats=st,
                    stat_mode=ops.STAT_SQ if with_stats else 0)
        torch.cuda.synchronize()
        Cg = C[:, :N].double()
        bad = (Cg - Cref).abs() > tol(dtype)["atol"] * max(1.0, scale) + tol(dtype)["rtol"] * Cref.abs()
        assert int(bad.sum()) == 0, (slab, with_stats, int(bad.sum()))
        if with_stats:
            assert not torch.isnan(st).any()
            sm = st.sum(0).double()
            assert torch.allclose(sm[0], Cg.sum(0), rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale))
            assert torch.allclose(sm[1], (Cg * Cg).sum(0), rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale) ** 2)
        outs.append(C[:, :N].clone())
    assert torch.equal(outs[0], outs[1])   # the statistics do not change the output
    assert torch.equal(outs[0], outs[2])   # slab-major (weights in the queue) == plain (LDS-weights kernel), bit for bit


@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(2000, 432, 24), (1500, 288, 16), (4111, 720, 40), (3000, 203, 80), (2500, 1440, 96), (1029, 400, 192),
                                   (30000, 432, 24)])
def test_gemm_nt_streaming_kernel(gpu_lib, M, N, K, act):
    """k_gemm_nt_st (bf16, no prologue, K % 8 == 0): expand-forward form and masked input-gradient form, slab-major and plain
    outputs, ragged M / N, all three activations; against fp64 and the two layouts against each other (bit-identical)."""
    ops = _ops()
    from atomnas_amd.ops import Slab
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K + act)
    r = lambda *s: torch.randn(*s, generator=g)
    A, Z, W = r(M, K), r(M, N), r(N, K) / K ** 0.5
    zs, zh = torch.rand(N, generator=g) + 0.5, r(N) * 0.3
    rd = lambda t: t.to(dtype).double()
    act2d = lambda t, c: torch.cat([t.to(dtype), torch.zeros(t.shape[0], pad8(c) - c, dtype=dtype)], 1).cuda()
    Ad, Zd, Wp = act2d(A, K), act2d(Z, N), pack_w(W, dtype)
    Cref = rd(A) @ rd(W).t()
    pre = rd(Z) * zs.double() + zh.double()
    if act == 3:
        sg = torch.sigmoid(pre)
        Mref = Cref * (sg * (1 + pre * (1 - sg)))
    else:
        Mref = Cref * ((pre > 0) & (pre < (6.0 if act == 2 else float("inf")))).double()
    scale = float(Cref.abs().max())
    outs = {}
    for slab in ([False, True] if N % 8 == 0 else [True]):
        for form in ("fwd", "mask"):
            C = Slab(M, N, dtype, "cuda", zero=True) if slab else fresh(M, N, dtype)
            st = poisoned_stats(64, N)
            if form == "fwd":
                ops.gemm_nt(Ad, Wp, C, M, N, K, stats=st, stat_mode=ops.STAT_SQ)
            else:
                ops.gemm_nt(Ad, Wp, C, M, N, K, z=_slab(Zd, N) if slab else Zd, zscale=cvec(zs), zshift=cvec(zh), mask=act, stats=st,
                            stat_mode=ops.STAT_Z)
            torch.cuda.synchronize()
            Cp = C.to_plain() if slab else C
            Cg = Cp[:, :N].double().cpu()
            ref = Cref if form == "fwd" else Mref
            assert_close("C " + form, Cg, ref, tol(dtype)["rtol"], tol(dtype)["atol"] * max(1.0, scale), outlier_frac=1e-4 if form == "mask" else 0.0)
            assert float(Cp[:, N:].float().abs().max()) == 0.0 if Cp.shape[1] > N else True
            s = st.sum(0)
            assert_close("s1 " + form, s[0], Cg.sum(0), rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale))
            s2 = (Cg * Cg).sum(0) if form == "fwd" else (Cg * rd(Z)).sum(0)
            assert_close("s2 " + form, s[1], s2, rtol=1e-4, atol=1e-3 * M ** 0.5 * max(1.0, scale) ** 2)
            outs[(slab, form)] = (Cp[:, :N].clone(), st.clone())
    if N % 8 == 0:
        for form in ("fwd", "mask"):
            assert torch.equal(outs[(False, form)][0], outs[(True, form)][0])
            assert torch.equal(outs[(False, form)][1], outs[(True, form)][1])


@pytest.mark.parametrize("M,N,K", [(12544, 3456, 192), (12544, 1728, 192), (50176, 1728, 96)])
def test_gemm_nt_streaming_kernel_bench_shapes_repeated(gpu_lib, M, N, K):
    """The expand forward of the late stages at the bench's own sizes (bs 256: 7x7 -> M = 12544, K = 192; 14x14 -> M = 50176), every
    element against an fp32 matmul on the GPU, output pre-filled with NaN, repeated launches.  Round 5 found the 7x7 instance writing the
    NEXT store's byte offset into the first dword of ~4400 of 43 M outputs per launch (a store-data hazard of `buffer_store ... sN offen`
    under a backed-up store queue, csrc/pwconv.hip k_gemm_nt_st::finish; profiles/r05_st_store_hazard.txt): different elements in every
    launch, nothing at the small shapes of the other tests."""
    ops = _ops()
    from atomnas_amd.ops import Slab
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda", generator=g).bfloat16()
    W = torch.randn(N, K, device="cuda", generator=g) / K ** 0.5
    Wp = torch.zeros((N + 63) // 64 * 64, (K + 31) // 32 * 32, dtype=torch.bfloat16, device="cuda")
    Wp[:N, :K] = W.bfloat16()
    ref = A.float() @ Wp[:N, :K].float().t()
    rows = ops.stat_rows_for(N)
    for rep in range(4):
        for slab in (True, False):
            C = Slab(M, N, torch.bfloat16, "cuda") if slab else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            (C.t if slab else C).fill_(float("nan"))
            st = torch.full((rows, 2, N), float("nan"), device="cuda")
            ops.gemm_nt(A, Wp, C, M, N, K, stats=st, stat_mode=ops.STAT_SQ, stat_rows=rows)
            torch.cuda.synchronize()
            Cp = (C.to_plain() if slab else C)[:, :N].float()
            bad = ~((Cp - ref).abs() <= 1.2e-2 * ref.abs() + 2e-2)
            assert int(bad.sum()) == 0, "rep %d %s: %d wrong outputs, first rows %s" % (
                rep, "slab" if slab else "plain", int(bad.sum()), torch.nonzero(bad.any(1)).flatten()[:8].tolist())
            s = st.sum(0)
            assert torch.allclose(s[0], Cp.sum(0), rtol=1e-4, atol=1e-2 * M ** 0.5)
            assert torch.allclose(s[1], (Cp * Cp).sum(0), rtol=1e-4, atol=1e-2 * M ** 0.5)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,C", [(1000, 80), (777, 96), (50, 13), (4096, 192)])
def test_bnbwd_apply(gpu_lib, dtype, M, C):
    ops = _ops()
    g = torch.Generator().manual_seed(M + C)
    r = lambda *s: torch.randn(*s, generator=g)
    G, X = r(M, C), r(M, C)
    c1, c2, c3 = torch.rand(C, generator=g) + 0.5, r(C) * 0.2, r(C) * 0.2
    act2d = lambda t: torch.cat([t.to(dtype), torch.zeros(M, pad8(C) - C, dtype=dtype)], 1).cuda()
    Y = fresh(M, C, dtype)
    ops.bnbwd_apply(act2d(G), act2d(X), cvec(c1), cvec(c2), cvec(c3), Y, M, C)
    torch.cuda.synchronize()
    ref = c1.double() * G.to(dtype).double() + c2.double() * X.to(dtype).double() + c3.double()
    t = tol(dtype)
    assert_close("dP", Y[:, :C].double().cpu(), ref, t["rtol"], t["atol"] * 4)
    if pad8(C) > C:
        assert float(Y[:, C:].float().abs().max()) == 0.0


# ---------------------------------------------------------------------------------------------- fused project backward
@pytest.mark.parametrize("slab", [True, False])
@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("M,oup,hid", [(5000, 24, 432), (1500, 16, 288), (4100, 40, 720), (1031, 8, 96), (3000, 48, 203 + 5), (20000, 32, 336),
                                       (2500, 56, 304), (2048, 64, 720)])
def test_project_bwd_fused_matches_the_two_gemm_form(gpu_lib, M, oup, hid, act, slab):
    """atomnas_project_bwd = atomnas_pw_gemm_nt(activation mask, STAT_Z) + atomnas_pw_gemm_tn of the projection's backward
    (models/mobilenet_base.py:338) on the differentiated BatchNorm output dP (atomnas_bnbwd_apply), with the raw depthwise output read
    once (k_gemm_nt_st, ST_PBWD).  Same MFMA sequence for the input gradient; the weight gradient groups its partials by row range.
    Both also against fp64."""
    ops = _ops()
    from atomnas_amd.ops import Slab
    dtype = torch.bfloat16
    assert ops.project_bwd_supported(oup, hid, dtype) and not ops.project_bwd_supported(80, 1440, dtype)   # oup <= 64 since ABI 9
    g = torch.Generator().manual_seed(M + oup + hid + act)
    r = lambda *s: torch.randn(*s, generator=g)
    act2d = lambda t, c: torch.cat([t.to(dtype), torch.zeros(t.shape[0], pad8(c) - c, dtype=dtype)], 1).cuda()
    G, P, Z = act2d(r(M, oup), oup), act2d(r(M, oup), oup), act2d(r(M, hid) * 2, hid)
    Wp = r(oup, hid) / hid ** 0.5                       # projection weight [oup, hid]
    c1, c2, c3 = cvec(torch.rand(oup, generator=g) + 0.5), cvec(r(oup) * 0.2), cvec(r(oup) * 0.2)
    zs, zh = cvec(torch.rand(hid, generator=g) + 0.5), cvec(r(hid) * 0.5)
    wpt = pack_w(Wp, dtype, transposed=True)            # Wp^T packed: [pad64(hid)][pad32(oup)]
    wz = (lambda t: Slab.from_plain(t, hid)) if slab else (lambda t: t)
    Zs = wz(Z)
    mk = lambda: (Slab(M, hid, dtype, "cuda", zero=True) if slab else fresh(M, hid, dtype))
    dPt = fresh(M, oup, dtype)
    ops.bnbwd_apply(G, P, c1, c2, c3, dPt, M, oup)
    gh1, gh2 = mk(), mk()
    st1, st2 = poisoned_stats(128, hid), poisoned_stats(128, hid)
    dw1 = torch.full((oup, hid), 0.5, dtype=torch.float32, device="cuda")
    dw2 = dw1.clone()
    assert ops.project_bwd_dp_supported(M, oup, hid, dPt, Zs, gh1, 128)
    ops.project_bwd(dPt, wpt, Zs, zs, zh, act, gh1, st1, dw1.view(-1), hid, 1, M, oup, hid)
    ops.gemm_tn(dPt, oup, Zs, hid, dw2.view(-1), hid, 1, M, v_mode=ops.PRO_BNRELU, vc1=zs, vc2=zh, v_relu=act)
    ops.gemm_nt(dPt, wpt, gh2, M, hid, oup, z=Zs, zscale=zs, zshift=zh, mask=act, stats=st2, stat_mode=ops.STAT_Z)
    torch.cuda.synchronize()
    a, b = (gh1.to_plain() if slab else gh1)[:, :hid].float(), (gh2.to_plain() if slab else gh2)[:, :hid].float()
    assert float((a != b).float().mean()) < 0.02
    assert torch.allclose(a, b, rtol=2e-2, atol=2e-2 * float(b.abs().max()))
    assert not torch.isnan(st1).any()
    s1, s2 = st1.sum(0), st2.sum(0)
    assert torch.allclose(s1, s2, rtol=1e-3, atol=2e-3 * float(s2.abs().max())), float((s1 - s2).abs().max())
    assert torch.allclose(dw1, dw2, rtol=1e-3, atol=2e-3 * float(dw2.abs().max())), float((dw1 - dw2).abs().max())
    # fp64 reference of the weight gradient (operands rounded to bf16 as the MFMAs see them)
    dP = dPt[:, :oup].double().cpu()
    pre = Z[:, :hid].double().cpu() * zs[:hid].double().cpu() + zh[:hid].double().cpu()
    A = {1: torch.relu(pre), 2: pre.clamp(0, 6), 3: pre * torch.sigmoid(pre)}[act].float().to(dtype).double()
    ref_dw = dP.t() @ A + 0.5
    assert_close("dwp", dw1, ref_dw, rtol=2e-3, atol=3e-3 * float(ref_dw.abs().max()))
