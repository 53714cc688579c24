"""Kernel-level parity (GPU) of the round-4 E-elimination EXPERIMENT (atomnas_amd/csrc/experimental/: the expand 1x1 convolution +
BatchNorm + activation computed on chip in front of the depthwise convolution, forward and backward; not part of the product
library) against a float64 torch restatement of models/mobilenet_base.py:316-336 on inputs rounded to bf16.  Runs only when the
experiment library is loaded:
    tools/build_xdw_experiment.sh && ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libxdw.so \
        python -m pytest tools/experiments/test_xdw_experimental_gpu.py -m gpu -p no:cacheprovider
(kept out of tests/: the product suite has no test that can only skip)
"""
import itertools
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, "tests"))   # kutil, conftest fixtures

import pytest
import torch
import torch.nn.functional as F

from kutil import assert_close, cvec, pad8, rounded

pytestmark = pytest.mark.gpu
BF = torch.bfloat16

# (N, H, W, inp, C): whole-image tiles with several images per tile (ragged batch), row tiles, one / two 32-channel chunks of x
SHAPES = [(5, 7, 7, 16, 16), (3, 14, 14, 24, 32), (2, 28, 28, 40, 48), (1, 56, 56, 24, 16), (2, 35, 28, 40, 32), (3, 21, 14, 32, 48),
          (2, 14, 14, 64, 16)]
ACTS = {1: torch.relu, 2: lambda t: torch.clamp(t, 0.0, 6.0), 3: lambda t: t * torch.sigmoid(t)}


class _Ops:
    """product ops + the experiment's wrappers"""
    def __getattr__(self, name):
        import os
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import xdw_ops
        from atomnas_amd import ops
        if not xdw_ops.available():
            pytest.skip("the E-elimination experiment library is not loaded (tools/build_xdw_experiment.sh, ATOMNAS_HIP_LIB)")
        return getattr(xdw_ops, name) if hasattr(xdw_ops, name) and name.startswith(("xdw_", "gram_stats")) else getattr(ops, name)


def _ops():
    return _Ops()


def pad(n, m):
    return (n + m - 1) // m * m


def pack_we(w):
    """[C, inp] -> packed expand weight [pad64(C)][pad32(inp)] bf16 (atomnas_pack_weights mode 0)"""
    C, inp = w.shape
    buf = torch.zeros(pad(C, 64), pad(inp, 32), dtype=BF, device="cuda")
    buf[:C, :inp] = w.to(BF).cuda()
    return buf


def taps(w):
    C, _, k, _ = w.shape
    t = torch.zeros(k * k, pad8(C), dtype=torch.float32, device="cuda")
    t[:, :C] = w.reshape(C, k * k).t().float().cuda()
    return t


def slab_to_nchw(s, N, H, W, C):
    return s.to_plain()[:, :C].double().cpu().reshape(N, H, W, C).permute(0, 3, 1, 2).contiguous()


def nchw_to_slab(t, ops):
    N, C, H, W = t.shape
    return ops.Slab.from_plain(t.permute(0, 2, 3, 1).reshape(-1, C).to(BF).cuda().contiguous(), C)


def make(N, H, W, inp, C, k, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, inp, H, W, generator=g)
    we = torch.randn(C, inp, generator=g) / inp ** 0.5
    wd = torch.randn(C, 1, k, k, generator=g) * 0.3
    sc = torch.rand(C, generator=g) + 0.5
    sh = torch.randn(C, generator=g) * 0.3
    return g, x, we, wd, sc, sh


def expand_ref(x, we):
    """E[n, c, h, w] in float64 from the bf16-rounded operands"""
    return torch.einsum("nihw,ci->nchw", rounded(x, BF), rounded(we, BF))


@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("k", [3, 5, 7])
@pytest.mark.parametrize("N,H,W,inp,C", SHAPES)
def test_xdw_fwd(gpu_lib, N, H, W, inp, C, k, act):
    ops = _ops()
    if not ops.xdw_supported(N, H, W, inp, C, k, 1, BF):
        pytest.skip("no instance")
    g, x, we, wd, sc, sh = make(N, H, W, inp, C, k, 31 * k + inp + C + act)
    v = lambda t: t.double().view(1, -1, 1, 1)
    a = ACTS[act](expand_ref(x, we) * v(sc) + v(sh))
    yref = F.conv2d(a, wd.double(), None, 1, (k - 1) // 2, 1, C)
    M = N * H * W
    xb = x.permute(0, 2, 3, 1).reshape(M, inp).to(BF).cuda().contiguous()
    y = ops.Slab(M, C, BF, "cuda")
    y.t.fill_(7.0)
    rows = 64
    stats = torch.full((rows, 2, C), float("nan"), dtype=torch.float32, device="cuda")
    ops.xdw_fwd(xb, inp, pack_we(we), cvec(sc), cvec(sh), act, taps(wd), y, stats, C, N, H, W, C, k)
    torch.cuda.synchronize()
    yk = slab_to_nchw(y, N, H, W, C)
    assert_close("y", yk, yref, rtol=1.2e-2, atol=1e-2)
    st = stats.sum(0)
    assert_close("sum", st[0], yk.sum((0, 2, 3)), rtol=1e-4, atol=1e-3)
    assert_close("sumsq", st[1], (yk * yk).sum((0, 2, 3)), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("workers", [0, 3])
def test_xdw_fwd_long_tile_walk(gpu_lib, workers, monkeypatch):
    """few workers: every workgroup walks many tiles (window rewrite, next-tile fragment prefetch, image borders inside a walk)"""
    import os
    import subprocess
    import sys
    _ops().xdw_fwd   # skips when the experiment library is not loaded
    env = dict(os.environ)
    if workers:
        env["ATOMNAS_DW_MAX_WORKERS"] = str(workers)
    code = ("import sys; sys.path.insert(0, 'tests'); sys.path.insert(0, 'tools/experiments'); import torch, test_xdw_experimental_gpu as t; "
            "[t.test_xdw_fwd(None, 9, 28, 28, 40, 48, k, 1) for k in (3, 7)]; [t.test_xdw_bwd(None, 9, 28, 28, 40, 48, k, 1) for k in (3, 7)]; print('ok')")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=_ROOT)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("act", [1, 2, 3])
@pytest.mark.parametrize("k", [3, 5, 7])
@pytest.mark.parametrize("N,H,W,inp,C", SHAPES)
def test_xdw_bwd(gpu_lib, N, H, W, inp, C, k, act):
    ops = _ops()
    if not ops.xdw_supported(N, H, W, inp, C, k, 1, BF):
        pytest.skip("no instance")
    g, x, we, wd, sc, sh = make(N, H, W, inp, C, k, 77 * k + inp + C + act)
    gup = torch.randn(N, C, H, W, generator=g)
    yraw = torch.randn(N, C, H, W, generator=g)
    c1 = torch.rand(C, generator=g) + 0.5
    c2 = torch.randn(C, generator=g) * 0.1
    c3 = torch.randn(C, generator=g) * 0.1
    v = lambda t: t.double().view(1, -1, 1, 1)
    E = expand_ref(x, we)
    pre = (E * v(sc) + v(sh)).requires_grad_(True)
    wdd = wd.double().requires_grad_(True)
    y = F.conv2d(ACTS[act](pre), wdd, None, 1, (k - 1) // 2, 1, C)
    dy = v(c1) * rounded(gup, BF) + v(c2) * rounded(yraw, BF) + v(c3)
    (y * dy).sum().backward()
    href = pre.grad
    M = N * H * W
    xb = x.permute(0, 2, 3, 1).reshape(M, inp).to(BF).cuda().contiguous()
    h = ops.Slab(M, C, BF, "cuda")
    h.t.fill_(7.0)
    dw = torch.zeros(C, k * k, dtype=torch.float32, device="cuda")
    rows = 64
    stats = torch.full((rows, 2, C), float("nan"), dtype=torch.float32, device="cuda")
    ops.xdw_bwd(nchw_to_slab(gup, ops), nchw_to_slab(yraw, ops), cvec(c1), cvec(c2), cvec(c3), xb, inp, pack_we(we), cvec(sc), cvec(sh), act,
                taps(wd), h, dw, stats, C, N, H, W, C, k)
    torch.cuda.synchronize()
    hk = slab_to_nchw(h, N, H, W, C)
    # a ReLU pre-activation within fp32 rounding of zero may land on the other side than in float64: rare single elements
    assert_close("h", hk, href, rtol=1.2e-2, atol=4e-2, outlier_frac=1e-4)
    assert_close("dw", dw.reshape(C, 1, k, k), wdd.grad, rtol=2e-3, atol=2e-3 * float(wdd.grad.abs().max()))
    st = stats.sum(0)
    assert_close("sum_h", st[0], hk.sum((0, 2, 3)), rtol=1e-4, atol=2e-3)
    assert_close("sum_he", st[1], (hk * E).sum((0, 2, 3)), rtol=2e-4, atol=5e-3)


