"""Kernel-level parity (GPU) of the Squeeze-and-Excitation entry points against a float64 torch restatement of
SqueezeAndExcitation.forward (models/mobilenet_base.py:109-112) and its autograd backward, on padded multi-segment channel
layouts as the fused block (mobilenet_base.py:256-267) lays its branches out."""
import pytest
import torch
import torch.nn.functional as F

from kutil import assert_close

pytestmark = pytest.mark.gpu

ACT_RELU, ACT_SWISH = 1, 3


def _layout(hid_list):
    """branch widths -> (HT, total, cmap, segments[(padded offset, real offset, width)])"""
    segs, o, st = [], 0, 0
    for h in hid_list:
        segs.append((o, st, h))
        o += (h + 15) // 16 * 16
        st += h
    cmap = torch.full((o,), -1, dtype=torch.int32)
    for sg, s0, h in segs:
        cmap[sg:sg + h] = torch.arange(s0, s0 + h, dtype=torch.int32)
    return o, st, cmap, segs


def _pack(w1, w2, b2, cmap, HT):
    hid = w1.shape[0]
    valid = cmap >= 0
    idx = cmap[valid].long()
    w1p = torch.zeros(hid, HT, dtype=torch.float32)
    w2t = torch.zeros(hid, HT, dtype=torch.float32)
    b2p = torch.zeros(HT, dtype=torch.float32)
    w1p[:, valid] = w1[:, idx]
    w2t[:, valid] = w2[idx, :].t()
    b2p[valid] = b2[idx]
    return w1p, w2t, b2p


def _act(x, code):
    return F.relu(x) if code == ACT_RELU else x * torch.sigmoid(x)


CASES = [
    # N, HW, branch widths, hidden units, SE activation
    (5, 49, [23, 7, 40], 12, ACT_SWISH),
    (16, 9, [188, 137, 306], 96, ACT_SWISH),       # wide block: the 1024-thread form
    (3, 196, [32], 16, ACT_RELU),
    (4, 3136, [15, 23, 13], 8, ACT_SWISH),         # many pixels, few channels: the pooled sums arrive in several planes
    (9, 4, [823, 738, 749], 96, ACT_SWISH),        # the last block of AtomNAS-C+: LDS above 64 KiB
    (130, 1, [15, 23, 13], 8, ACT_SWISH),          # batch beyond two image chunks of the weight-gradient kernel
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CASES)
def test_se_forward_backward(gpu_lib, case, dtype):
    from atomnas_amd import ops
    from atomnas_amd.ops import Slab
    N, HW, hid_list, hid, se_act = case
    HT, total, cmap, segs = _layout(hid_list)
    M = N * HW
    g = torch.Generator().manual_seed(N * 131 + HT)
    act = ACT_SWISH
    valid = cmap >= 0
    Dv = torch.randn(M, HT, generator=g)
    Dv[:, ~valid] = 0
    Dv = Dv.to(dtype)
    scale = torch.rand(HT, generator=g) + 0.5
    shift = torch.randn(HT, generator=g) * 0.3
    scale[~valid] = 0
    shift[~valid] = 0
    w1 = torch.randn(hid, total, generator=g) / total ** 0.5
    b1 = torch.randn(hid, generator=g) * 0.1
    w2 = torch.randn(total, hid, generator=g) / hid ** 0.5
    b2 = torch.randn(total, generator=g) * 0.1
    dSv = torch.randn(M, HT, generator=g)
    dSv[:, ~valid] = 0
    dSv = dSv.to(dtype)

    # ---- float64 reference on the real channels
    idx = valid.nonzero().flatten()
    D64 = Dv.double()[:, idx].requires_grad_(True)
    p64 = [t.double().requires_grad_(True) for t in (w1, b1, w2, b2)]
    A = _act(D64 * scale.double()[idx] + shift.double()[idx], act)
    pooled_r = A.view(N, HW, -1).mean(1)
    hpre_r = pooled_r @ p64[0].t() + p64[1]
    gate_r = torch.sigmoid(_act(hpre_r, se_act) @ p64[2].t() + p64[3])
    S_r = A * gate_r.repeat_interleave(HW, 0)
    # the gradient wrt the pre-activation a = D*scale+shift is what the kernel emits: differentiate wrt a
    a_leaf = (D64 * scale.double()[idx] + shift.double()[idx]).detach().requires_grad_(True)
    A2 = _act(a_leaf, act)
    pooled2 = A2.view(N, HW, -1).mean(1)
    hpre2 = pooled2 @ p64[0].t() + p64[1]
    gate2 = torch.sigmoid(_act(hpre2, se_act) @ p64[2].t() + p64[3])
    S2 = A2 * gate2.repeat_interleave(HW, 0)
    S2.backward(dSv.double()[:, idx])
    g_r = a_leaf.grad

    # ---- the kernels
    dev = "cuda"
    idx_d, pad_d = idx.to(dev), (~valid).to(dev)
    Dd = Slab.from_plain(Dv.to(dev)) if dtype == torch.bfloat16 else Dv.to(dev)
    dSd = Slab.from_plain(dSv.to(dev)) if dtype == torch.bfloat16 else dSv.to(dev)
    sc, sh, cm = scale.to(dev), shift.to(dev), cmap.to(dev)
    w1p, w2t, b2p = (t.to(dev) for t in _pack(w1, w2, b2, cmap, HT))
    pooled = torch.full((N, HT), 7.0, device=dev)
    gate = torch.full((N, HT), 7.0, device=dev)
    hpre = torch.full((N, hid), 7.0, device=dev)
    parts = ops.se_pool_parts(N, HW, HT)
    pparts = torch.full((parts, N, HT), 7.0, device=dev)
    ops.se_squeeze(Dd, sc, sh, act, pparts, N, HW, HT)
    ops.se_mlp_fwd(pparts, pooled, cm, w1p, b1.to(dev), w2t, b2p, se_act, hpre, gate, N, HT, hid)
    if dtype == torch.bfloat16:
        Sd = Slab(M, HT, dtype, dev, zero=True)
    else:
        Sd = torch.zeros(M, HT, dtype=dtype, device=dev)
    ops.se_scale(Dd, sc, sh, act, gate, Sd, M, HW, HT)
    rt = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert_close("pooled", pooled[:, idx_d], pooled_r.detach(), 1e-4, 1e-5)
    assert float(pooled[:, pad_d].abs().max() if (~valid).any() else 0.0) == 0.0
    assert_close("hpre", hpre, hpre_r.detach(), 1e-4, 1e-5)
    assert_close("gate", gate[:, idx_d], gate_r.detach(), 1e-4, 1e-5)
    assert float(gate[:, pad_d].abs().max() if (~valid).any() else 0.0) == 0.0
    Sp = Sd.to_plain() if dtype == torch.bfloat16 else Sd
    assert_close("S", Sp[:, idx_d], S_r.detach(), rt, 1e-3 if dtype == torch.bfloat16 else 1e-5)

    dz2, dpooled = (torch.full((N, HT), 7.0, device=dev) for _ in range(2))
    dgate = torch.full((parts, N, HT), 7.0, device=dev)
    dz1 = torch.full((N, hid), 7.0, device=dev)
    base = [torch.randn(hid * total, generator=g), torch.randn(hid, generator=g), torch.randn(total * hid, generator=g),
            torch.randn(total, generator=g)]
    dw1, db1, dw2, db2 = (t.clone().to(dev) for t in base)   # the kernels accumulate into the gradient arena
    ops.se_bwd_gate(dSd, Dd, sc, sh, act, gate, pooled, cm, w1p, w2t, hpre, dgate, dz2, dz1, dpooled, dw1, db1, dw2, db2, N, HW, HT, total,
                    hid, se_act=se_act)
    rows = 64
    st2 = torch.full((rows, 2, HT), float("nan"), device=dev)
    gd = Slab(M, HT, dtype, dev, zero=True) if dtype == torch.bfloat16 else torch.zeros(M, HT, dtype=dtype, device=dev)
    ops.se_bwd_apply(dSd, Dd, sc, sh, act, gate, dpooled, gd, st2, M, HW, HT, stat_rows=rows)
    gp = (gd.to_plain() if dtype == torch.bfloat16 else gd)
    assert_close("g", gp[:, idx_d], g_r, rt, 2e-3 if dtype == torch.bfloat16 else 2e-5)
    gq = gp.double()
    assert_close("stats2 sum g", st2[:, 0].sum(0).double(), gq.sum(0), 1e-4, 1e-4 * max(1.0, float(gq.abs().sum(0).max())))
    assert_close("stats2 sum g*D", st2[:, 1].sum(0).double(), (gq * Dv.to(dev).double()).sum(0), 1e-4,
                 1e-4 * max(1.0, float((gq * Dv.to(dev).double()).abs().sum(0).max())))
    # weight gradients: differentiate the float64 graph wrt the dense layers
    for p in p64:
        p.grad = None
    S_r.backward(dSv.double()[:, idx])
    for name, got, b, p in (("dw1", dw1, base[0], p64[0]), ("db1", db1, base[1], p64[1]), ("dw2", dw2, base[2], p64[2]),
                            ("db2", db2, base[3], p64[3])):
        ref = p.grad.reshape(-1)
        scale_ = max(1.0, float(ref.abs().max()))
        assert_close(name, got.double().cpu() - b.double(), ref, 1e-3, 2e-5 * scale_ * (N * HW) ** 0.5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,C,H,hid,act", [(3, 40, 7, 12, "Swish"), (2, 24, 14, 8, "ReLU")])
def test_se_module_stand_alone(gpu_lib, dtype, N, C, H, hid, act):
    """SqueezeAndExcitation called as a module of its own (the reference's API, models/mobilenet_base.py:109-112): output, input
    gradient and the gradients of both dense layers against the reference's formula in float64."""
    from atomnas_amd.models import mobilenet_base as mb
    fn = mb.Swish if act == "Swish" else torch.nn.ReLU
    se = mb.SqueezeAndExcitation(C, hid, active_fn=fn)
    g = torch.Generator().manual_seed(C + H)
    with torch.no_grad():
        for p in se.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * (0.3 if p.dim() > 1 else 0.1))
    x = torch.randn(N, C, H, H, generator=g).to(dtype)
    gout = torch.randn(N, C, H, H, generator=g).to(dtype)
    # float64 reference with the module's own parameters
    xr = x.double().requires_grad_(True)
    pr = [p.detach().double().requires_grad_(True) for p in (se.se_reduce.weight, se.se_reduce.bias, se.se_expand.weight, se.se_expand.bias)]
    t = xr.mean([2, 3], keepdim=True)
    t = F.conv2d(t, pr[0], pr[1])
    t = t * torch.sigmoid(t) if act == "Swish" else F.relu(t)
    ref = torch.sigmoid(F.conv2d(t, pr[2], pr[3])) * xr
    ref.backward(gout.double())
    se.cuda()
    xg = x.cuda().requires_grad_(True)
    out = se(xg)
    assert out.shape == x.shape and out.dtype == dtype
    out.backward(gout.cuda())
    torch.cuda.synchronize()
    rt, at = (2e-2, 2e-2) if dtype == torch.bfloat16 else (1e-4, 1e-5)
    assert_close("out", out, ref.detach(), rt, at)
    assert_close("dx", xg.grad, xr.grad, rt, at * max(1.0, float(xr.grad.abs().max())))
    for name, p, r in zip(("dw1", "db1", "dw2", "db2"), (se.se_reduce.weight, se.se_reduce.bias, se.se_expand.weight, se.se_expand.bias), pr):
        assert_close(name, p.grad, r.grad, 2e-2 if dtype == torch.bfloat16 else 1e-3, (3e-2 if dtype == torch.bfloat16 else 1e-4) * max(1.0, float(r.grad.abs().max())))
