"""Kernel-level parity (GPU) of the expand backward without E (csrc/xbwd.hip, include/atomnas_hip.h): the Gram matrix of the block
input, the inp x inp corrections, and atomnas_expand_bwd -- against float64 torch restatements of
models/mobilenet_base.py:316-320 (backward) on inputs rounded to bf16.
"""
import pytest
import torch

from kutil import assert_close, cvec

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _ops():
    from atomnas_amd import ops
    return ops


def pad(n, m):
    return (n + m - 1) // m * m


def pack_we(w):
    """[C, inp] -> packed expand weight [pad64(C)][pad32(inp)] bf16 (atomnas_pack_weights mode 0)"""
    C, inp = w.shape
    buf = torch.zeros(pad(C, 64), pad(inp, 32), dtype=BF, device="cuda")
    buf[:C, :inp] = w.to(BF).cuda()
    return buf


@pytest.mark.parametrize("M,inp,C", [(5000, 24, 432), (3000, 40, 720), (1234, 16, 96), (2000, 64, 128)])
def test_gram_and_coeffs(gpu_lib, M, inp, C):
    ops = _ops()
    g = torch.Generator().manual_seed(M + inp)
    x = torch.randn(M, inp, generator=g) + 0.3
    we = torch.randn(C, inp, generator=g) / inp ** 0.5
    xb = x.to(BF).cuda().contiguous()
    wp = pack_we(we)
    xr, wr = x.to(BF).double(), we.to(BF).double()
    E = xr @ wr.t()
    gram = torch.full((inp, inp), float("nan"), dtype=torch.float32, device="cuda")
    sx = torch.full((inp,), float("nan"), dtype=torch.float32, device="cuda")
    ops.gram(xb, M, inp, gram, sx)
    torch.cuda.synchronize()
    assert_close("gram", gram, xr.t() @ xr, rtol=1e-5, atol=1e-3)
    assert_close("sx", sx, xr.sum(0), rtol=1e-5, atol=1e-2)
    # corrections of the expand backward
    c2 = torch.randn(C, generator=g) * 0.1
    c3 = torch.randn(C, generator=g) * 0.1
    mp = torch.zeros(pad(inp, 64), pad(inp, 32), dtype=BF, device="cuda")
    vb = torch.empty(inp, dtype=torch.float32, device="cuda")
    dwe0 = torch.randn(C, inp, generator=g)
    dwe = dwe0.float().cuda().contiguous()
    ops.xb_coeffs(cvec(c2), cvec(c3), wp, gram, sx, inp, C, mp, vb, dwe)
    torch.cuda.synchronize()
    Mref = wr.t() @ (c2.double().view(-1, 1) * wr)
    assert_close("M", mp[:inp, :inp], Mref, rtol=1e-2, atol=1e-2 * float(Mref.abs().max()))
    assert float(mp[inp:].abs().max() if mp.shape[0] > inp else 0) == 0.0
    assert_close("v", vb, c3.double() @ wr, rtol=1e-4, atol=1e-4)
    dref = dwe0.double() + c2.double().view(-1, 1) * (wr @ (xr.t() @ xr)) + c3.double().view(-1, 1) * xr.sum(0).view(1, -1)
    assert_close("dwe", dwe, dref, rtol=1e-4, atol=1e-3 * float(dref.abs().max()))


@pytest.mark.parametrize("M,inp,hid", [(3000, 24, 432),      # ragged last row block
                                       (70001, 24, 432),     # several row blocks per workgroup, one row in the last block
                                       (66000, 16, 288),     # the 112 x 112 stage's shape (one x-channel tile, 5 chunks)
                                       (40000, 16, 720),     # 12 chunks
                                       (36000, 32, 304),     # hidden width not a multiple of 64: the last chunk ends inside the tensor
                                       (35000, 40, 240),     # three x-channel tiles
                                       (100000, 16, 96),     # two chunks only, many row blocks per workgroup: k_expand_bwd (the streaming
                                       (90000, 24, 64)])     # kernel's tile double-buffering needs three stages per row block)
def test_expand_bwd_without_e(gpu_lib, M, inp, hid):
    """atomnas_expand_bwd: gx = (c1*h) We + add (+ x M + v), dwe += (c1*h)^T x (slab-major h: the streaming kernel
    k_expand_bwd_s; the same cases through k_expand_bwd with ATOMNAS_XB_STREAM=0)"""
    ops = _ops()
    if not ops.expand_bwd_supported(inp, hid, BF):
        pytest.skip("no instance")
    g = torch.Generator().manual_seed(5)
    h = torch.randn(M, hid, generator=g)
    x = torch.randn(M, inp, generator=g)
    we = torch.randn(hid, inp, generator=g) / inp ** 0.5
    add = torch.randn(M, inp, generator=g)
    c1 = torch.rand(hid, generator=g) + 0.5
    mm = torch.randn(inp, inp, generator=g) * 0.2
    mm = mm + mm.t()
    vb = torch.randn(inp, generator=g)
    hb = ops.Slab.from_plain(h.to(BF).cuda().contiguous(), hid)
    wt = torch.zeros(pad(inp, 64), pad(hid, 32), dtype=BF, device="cuda")
    wt[:inp, :hid] = we.t().to(BF).cuda()
    mp = torch.zeros(pad(inp, 64), pad(inp, 32), dtype=BF, device="cuda")
    mp[:inp, :inp] = mm.to(BF).cuda()
    xb = x.to(BF).cuda().contiguous()
    dE = (c1.double().view(1, -1) * h.to(BF).double()).to(BF).double()   # the kernel rounds dE to the MFMA input type
    dref = dE.t() @ x.to(BF).double()
    for with_m, with_add in ((False, True), (True, True), (True, False)):
        gx = torch.full((M, inp), 7.0, dtype=BF, device="cuda")
        dwe = torch.zeros(hid, inp, dtype=torch.float32, device="cuda")
        ops.expand_bwd(hb, cvec(c1), xb, wt, add.to(BF).cuda().contiguous() if with_add else None, gx, dwe, M, inp, hid,
                       mp=mp if with_m else None, vb=cvec(vb) if with_m else None)
        torch.cuda.synchronize()
        gref = dE @ we.to(BF).double() + (add.to(BF).double() if with_add else 0.0)
        if with_m:
            gref = gref + x.to(BF).double() @ mm.to(BF).double() + vb.double().view(1, -1)
        assert_close("gx", gx, gref, rtol=1.2e-2, atol=2e-2 * float(gref.abs().max()))
        assert_close("dwe", dwe, dref, rtol=2e-3, atol=2e-3 * float(dref.abs().max()))


@pytest.mark.parametrize("slab", [True, False])
@pytest.mark.parametrize("M,inp,hid,res", [(5000, 24, 432, True), (777, 16, 288, False), (4100, 40, 304, True), (130, 8, 48, False),
                                           (9000, 16, 768, True), (2500, 32, 3 * 112, False)])
def test_expand_bwd_fused_matches_the_two_gemm_form(gpu_lib, M, inp, hid, res, slab):
    """atomnas_expand_bwd = atomnas_pw_gemm_nt + atomnas_pw_gemm_tn with c1 as their BNRELU scale (what functional._expand_backward_noe
    issues for the layers too wide for the fused kernel), with h read once: same products, the weight gradient grouped by workgroup
    (rounding-level difference).  Slab-major h takes the streaming kernel (k_expand_bwd_s), plain h the register-prefetch kernel."""
    ops = _ops()
    assert ops.expand_bwd_supported(inp, hid, BF) and not ops.expand_bwd_supported(40, 720, BF)   # see pwconv.hip
    g = torch.Generator().manual_seed(M + inp + hid)
    r = lambda *s: torch.randn(*s, generator=g)
    H = r(M, hid).to(BF).cuda()
    Hs = ops.Slab.from_plain(H, hid) if slab else H
    X, Gres = r(M, inp).to(BF).cuda(), r(M, inp).to(BF).cuda()
    We = r(hid, inp) / inp ** 0.5
    c1 = cvec(torch.rand(hid, generator=g) + 0.5)
    zeros = torch.zeros_like(c1)
    wt = torch.zeros(pad(inp, 64), pad(hid, 32), dtype=BF, device="cuda")
    wt[:inp, :hid] = We.t().to(BF).cuda()
    gx1 = torch.full((M, inp), 7.0, dtype=BF, device="cuda")
    gx2 = gx1.clone()
    dw1 = torch.full((hid, inp), 0.25, dtype=torch.float32, device="cuda")   # gradients ACCUMULATE into the arena
    dw2 = dw1.clone()
    ops.expand_bwd(Hs, c1, X, wt, Gres if res else None, gx1, dw1.view(-1), M, inp, hid)
    ops.gemm_tn(X, inp, Hs, hid, dw2.view(-1), 1, inp, M, v_mode=ops.PRO_BNRELU, vc1=c1, vc2=zeros, v_relu=0)
    ops.gemm_nt(Hs, wt, gx2, M, inp, hid, a_mode=ops.PRO_BNRELU, ac1=c1, ac2=zeros, a_relu=0, add=Gres if res else None)
    torch.cuda.synchronize()
    # the fused kernel folds c1 into the bf16 weights (input gradient) / applies it to the fp32 accumulators (weight gradient), the GEMMs
    # round c1*h to bf16: agreement to the operand roundings
    assert torch.allclose(gx1.float(), gx2.float(), rtol=2e-2, atol=2e-2 * float(gx2.float().abs().max()))
    assert torch.allclose(dw1, dw2, rtol=5e-3, atol=5e-3 * float(dw2.abs().max())), float((dw1 - dw2).abs().max())
