"""Bit-reproducibility (GPU): no entry point of the library accumulates floating-point values with atomics -- statistics leave
the kernels as partial rows (plain stores, fixed-order finalize), weight gradients as per-workgroup partials summed in a fixed
order -- so two runs of the same kernel, and two runs of the same training iterations, must agree bit for bit
(the reference's own loop is deterministic on one device: train.py:165-236)."""
import collections
import os
import sys

import pytest
import torch

from kutil import pad8

pytestmark = pytest.mark.gpu


def _act(M, C, dtype, seed):
    g = torch.Generator().manual_seed(seed)
    b = torch.zeros(M, pad8(C), dtype=dtype, device="cuda")
    b[:, :C] = torch.randn(M, C, generator=g).to(dtype).cuda()
    return b


@pytest.mark.parametrize("k,stride", [(3, 1), (5, 2), (7, 1)])
def test_dwconv_twice_bit_identical(gpu_lib, k, stride):
    from atomnas_amd import ops
    N, C, H = 16, 70, 56
    Ho = (H - 1) // stride + 1
    dt = torch.bfloat16
    x, gup, yraw = _act(N * H * H, C, dt, 1), _act(N * Ho * Ho, C, dt, 2), _act(N * Ho * Ho, C, dt, 3)
    g = torch.Generator().manual_seed(4)
    taps = torch.zeros(k * k, pad8(C), device="cuda")
    taps[:, :C] = torch.randn(k * k, C, generator=g).cuda() * 0.3
    vec = lambda s: torch.cat([torch.rand(C, generator=torch.Generator().manual_seed(s)) + 0.5, torch.zeros(pad8(C) - C)]).cuda()
    outs = []
    for rep in range(3):
        y = torch.zeros(N * Ho * Ho, pad8(C), dtype=dt, device="cuda")
        st = torch.full((300, 2, C), float("nan"), device="cuda")
        ops.dwconv_fwd(x, vec(5), vec(6), 1, taps, y, st, C, N, H, H, C, k, stride)
        h = torch.zeros(N * H * H, pad8(C), dtype=dt, device="cuda")
        dw = torch.zeros(C, k * k, device="cuda")
        st2 = torch.full((300, 2, C), float("nan"), device="cuda")
        ops.dwconv_bwd(gup, yraw, vec(7), vec(8), vec(9), x, vec(5), vec(6), 1, taps, h, dw, st2, C, N, H, H, C, k, stride)
        torch.cuda.synchronize()
        assert not torch.isnan(st).any() and not torch.isnan(st2).any()
        outs.append((y, st, h, dw, st2))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K", [(50000, 432, 24), (50000, 24, 432), (3000, 40, 139)])
def test_gemm_twice_bit_identical(gpu_lib, M, N, K):
    from atomnas_amd import ops
    dt = torch.bfloat16
    A, Z = _act(M, K, dt, 1), _act(M, N, dt, 2)
    g = torch.Generator().manual_seed(3)
    W = torch.zeros((N + 63) // 64 * 64, (K + 31) // 32 * 32, dtype=dt, device="cuda")
    W[:N, :K] = (torch.randn(N, K, generator=g) / K ** 0.5).to(dt).cuda()
    outs = []
    for rep in range(3):
        C = torch.zeros(M, pad8(N), dtype=dt, device="cuda")
        st = torch.full((200, 2, N), float("nan"), device="cuda")
        ops.gemm_nt(A, W, C, M, N, K, z=Z, stats=st, stat_mode=ops.STAT_Z)
        out = torch.zeros(K, N, device="cuda")
        ops.gemm_tn(A, K, Z, N, out, N, 1, M)
        torch.cuda.synchronize()
        assert not torch.isnan(st).any()
        outs.append((C, st, out))
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert torch.equal(a, b)


def _run_steps(dtype, use_graph, steps=3):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from trainstep_diag import setup
    model, sd, spec, pinfo, opt, ema, engine = setup(dtype, 64)
    N = 8
    ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-3, label_smoothing=0.1, batch_size=N, image_size=64, use_graph=use_graph)
    g = torch.Generator().manual_seed(5)
    for step in range(steps):
        x = torch.randn(N, 3, 64, 64, generator=g)
        y = torch.randint(0, 10, (N,), generator=g)
        ts.set_batch(x.cuda(), y.cuda())
        ts.step(lr=0.002 * (1 + step), rho=1e-3 * (1 + step))
    torch.cuda.synchronize()
    mgr = ts.mgr
    return collections.OrderedDict((k, getattr(mgr, k).detach().clone()) for k in ("P", "G", "SQ", "BUF", "EMA", "S", "SEMA")), ts.loss.clone()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_training_iterations_bit_identical(gpu_lib, dtype):
    """Three full iterations (forward, losses, backward, RMSprop, EMA) twice eagerly and twice through the captured graphs:
    parameters, gradients, optimizer state, EMA shadows, BN statistics and the logged losses are bit-identical across ALL runs."""
    runs = [_run_steps(dtype, g) for g in (False, False, True, True)]
    a0, l0 = runs[0]
    for arenas, loss in runs[1:]:
        assert torch.equal(loss, l0), (loss, l0)
        for k in a0:
            assert torch.equal(arenas[k], a0[k]), k


def test_deferred_reductions_bit_identical_to_immediate(gpu_lib, monkeypatch):
    """The batched form of the weight-gradient reductions (csrc/reduce.hip atomnas_reduce_defer / _flush, one launch per 56
    recorded jobs at the end of backward) adds every element's partials in the same order as the per-call launches: three
    iterations are bit-identical with the switch on and off, eagerly and through the graphs."""
    from atomnas_amd import engine
    runs = []
    for defer in (False, True):
        monkeypatch.setattr(engine, "_DEFER_REDUCE", defer)
        runs += [_run_steps(torch.bfloat16, False), _run_steps(torch.bfloat16, True)]
    a0, l0 = runs[0]
    for arenas, loss in runs[1:]:
        assert torch.equal(loss, l0), (loss, l0)
        for k in a0:
            assert torch.equal(arenas[k], a0[k]), k


def _run_full_size(use_graph, steps=3):
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import bench
    model, ts, hp = bench.build("atomnas_c_supernet", torch.bfloat16, 4, seed=11)[:3]
    ts.use_graph = use_graph
    g = torch.Generator().manual_seed(6)
    for step in range(steps):
        ts.set_batch(torch.randn(4, 3, 224, 224, generator=g).cuda(), torch.randint(0, 1000, (4,), generator=g).cuda())
        ts.step(rho=1e-4)
    torch.cuda.synchronize()
    return {k: getattr(ts.mgr, k).detach().clone() for k in ("P", "G", "SQ", "S")}, ts.loss.clone(), ts.topk.clone()


def test_full_size_supernet_graph_replay_equals_eager(gpu_lib):
    """The bench configuration itself (AtomNAS-C supernet, 44 MB gradient arena, 224x224; batch 4): the captured graphs replay the
    SAME iterations as the eager launches, bit for bit.  (Small networks did not show it when a captured memset node failed to clear
    the full-size gradient arena on replay: every per-step clear is a kernel of the library since.)"""
    (a0, l0, t0), (a1, l1, t1) = _run_full_size(False), _run_full_size(True)
    assert torch.equal(l0, l1), (l0, l1)
    assert torch.equal(t0, t1) and 0 <= int(t1[0]) <= 4 and int(t1[0]) <= int(t1[1]) <= 4
    for k in a0:
        assert torch.equal(a0[k], a1[k]), k
