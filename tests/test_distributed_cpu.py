"""Multi-process data-parallel path on CPU (gloo, world size 2): the reference-compatible wrapper / all-reduce helpers and the
host-side invariants the RCCL path relies on (rank-identical results, averaged gradients and BN statistics)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from atomnas_amd.utils import distributed as udist
    udist.init_dist(backend="gloo")
    try:
        torch.manual_seed(100 + rank)   # ranks start different on purpose
        net = nn.Sequential(nn.Conv2d(3, 4, 3, bias=False), nn.BatchNorm2d(4), nn.ReLU(), nn.Flatten(), nn.Linear(4 * 6 * 6, 5))
        wrapped = udist.AllReduceDistributedDataParallel(net)     # broadcast from rank 0
        ref = [p.detach().clone() for p in net.parameters()]
        gathered = [torch.zeros_like(ref[0]) for _ in range(world)]
        dist.all_gather(gathered, ref[0])
        assert all(torch.equal(g, gathered[0]) for g in gathered), "parameters differ after the wrapper's broadcast"
        x = torch.randn(2, 3, 8, 8)
        wrapped(x).square().mean().backward()
        local = [p.grad.clone() for p in net.parameters()]
        udist.allreduce_grads(wrapped.module)
        for p, l in zip(net.parameters(), local):
            parts = [torch.zeros_like(l) for _ in range(world)]
            dist.all_gather(parts, l)
            assert torch.allclose(p.grad, sum(parts) / world, atol=1e-7)
        rm_local = net[1].running_mean.clone()
        udist.allreduce_bn(wrapped.module)
        parts = [torch.zeros_like(rm_local) for _ in range(world)]
        dist.all_gather(parts, rm_local)
        assert torch.allclose(net[1].running_mean, sum(parts) / world, atol=1e-7)
        t = torch.tensor([float(rank + 1)])
        udist.dist_all_reduce_tensor(t)
        assert abs(float(t) - (1 + world) / 2) < 1e-6
        assert udist.is_master() == (rank == 0) and udist.get_world_size_fallback() == world
        # masks computed from rank-identical gammas are rank-identical: shrink needs no collective (train.py:46-63)
        gamma = net[1].weight.detach()
        masks = [torch.zeros(4, dtype=torch.bool) for _ in range(world)]
        dist.all_gather(masks, gamma.abs() > 1e-3)
        assert all(torch.equal(m, masks[0]) for m in masks)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_gloo_two_ranks():
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert dict(out) == {0: 1, 1: 1}
