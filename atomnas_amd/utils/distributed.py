"""Data-parallel plumbing with the reference's interface (utils/distributed.py) over RCCL.

The reference flattens every gradient into a fresh buffer, all-reduces it with NCCL, divides and copies 497 tensors back
(:131-139).  Here the gradients already live in one arena, so `allreduce_grads(model)` is ONE in-place RCCL all-reduce on
that arena plus a scale; `allreduce_bn` likewise on the BN-statistics arena, and the wrapper's initial parameter / buffer
broadcast (:183-190) is three broadcasts.  Modules that are not arena-backed (plain CPU modules in the gloo tests) fall back
to a coalesced all-reduce of their tensors -- communication only, no arithmetic of the hot path runs on the host.
"""
import functools
import os

import torch
import torch.distributed as dist
import torch.nn as nn
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


def _get_env(name):
    if name not in os.environ:
        raise RuntimeError('${} should be set'.format(name))
    return os.environ[name]


def init_dist(backend='nccl', **kwargs):
    """One process per GPU; rank / local rank / world size / master address from the launcher's environment (:25-32).
    backend 'nccl' is RCCL on ROCm."""
    if dist.is_initialized():
        raise RuntimeError('Should not init distributed twice')
    rank, local_rank = int(_get_env('RANK')), int(_get_env('LOCAL_RANK'))
    if backend == 'nccl':
        assert rank % torch.cuda.device_count() == local_rank
        torch.cuda.set_device(local_rank)
    dist.init_process_group(backend=backend, **kwargs)


def assert_initialized():
    if not dist.is_initialized():
        raise RuntimeError('Default process group is not initialized')


def get_local_rank():
    assert_initialized()
    return int(_get_env('LOCAL_RANK'))


def get_local_size():
    assert_initialized()
    return torch.cuda.device_count()


def get_rank_fallback():
    return dist.get_rank() if dist.is_initialized() else 0


def get_world_size_fallback():
    return dist.get_world_size() if dist.is_initialized() else 1


get_rank = dist.get_rank
get_world_size = dist.get_world_size


def is_master():
    return get_rank_fallback() == 0


def master_only(func):
    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        return func(*args, **kwargs) if is_master() else None
    return wrapper


def dist_all_reduce_tensor(tensor):
    world = dist.get_world_size()
    if world < 2:
        return tensor
    with torch.no_grad():
        dist.all_reduce(tensor)
        tensor.div_(world)
    return tensor


def _manager(model):
    m = model.module if isinstance(model, AllReduceDistributedDataParallel) else model
    mgr = getattr(m, '_arena', None)
    if mgr is not None and not mgr.dirty and mgr.G is not None:
        return mgr
    return None


def _allreduce_coalesced(tensors, world):
    buckets = {}
    for t in tensors:
        buckets.setdefault(t.type(), []).append(t)
    for group in buckets.values():
        flat = _flatten_dense_tensors(group)
        dist.all_reduce(flat)
        flat.div_(world)
        for t, s in zip(group, _unflatten_dense_tensors(flat, group)):
            t.copy_(s)


def allreduce_grads(model, *args, **kwargs):
    """Average gradients over the ranks (:155-161): one all-reduce of the gradient arena."""
    world = dist.get_world_size()
    mgr = _manager(model)
    if mgr is not None:
        dist.all_reduce(mgr.G)
        mgr.G.mul_(1.0 / world)
        return
    grads = [p.grad.data for p in model.parameters() if p.requires_grad and p.grad is not None]
    _allreduce_coalesced(grads, world)


def allreduce_bn(model, *args, **kwargs):
    """Average BN running statistics over the ranks (:164-169)."""
    world = dist.get_world_size()
    mgr = _manager(model)
    if mgr is not None:
        dist.all_reduce(mgr.S)
        mgr.S.mul_(1.0 / world)
        return
    tensors = [b for n, b in model.named_buffers() if 'running_var' in n or 'running_mean' in n]
    _allreduce_coalesced(tensors, world)


class AllReduceDistributedDataParallel(nn.Module):
    """Replicates rank 0's parameters and buffers at construction; forward is a pass-through; gradients are averaged by an
    explicit `allreduce_grads` after backward (no autograd hooks), exactly like the reference (:172-199)."""

    def __init__(self, module, dim=0, broadcast_buffers=True, bucket_cap_mb=25):
        super().__init__()
        self.module = module
        self.dim = dim
        self.broadcast_buffers = broadcast_buffers
        self._sync_params()

    def _sync_params(self):
        dev = next(self.module.parameters()).device
        if dev.type == 'cuda' and hasattr(self.module, 'features'):
            from .. import runtime
            mgr = runtime.manager_of(self.module)
            mgr.ensure()
            dist.broadcast(mgr.P, 0)
            if self.broadcast_buffers:
                dist.broadcast(mgr.S, 0)
                dist.broadcast(mgr.CNT, 0)
            return
        for t in list(self.module.state_dict().values()):
            dist.broadcast(t, 0)

    def forward(self, *inputs, **kwargs):
        return self.module(*inputs, **kwargs)
