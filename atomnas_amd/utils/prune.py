"""Resource-aware atomic block selection with the reference's interface (utils/prune.py)."""
import collections
import itertools
import logging

import torch
from torch import nn

from .. import ops
from ..models import mobilenet_base as mb


class PruneInfo(object):
    """Ordered name -> {penalty, per_channel_flops, mask, compress_masked} for the prunable BN gammas (utils/prune.py:11-86)."""

    def __init__(self, names, penalties):
        assert len(names) == len(penalties)
        self._info = collections.OrderedDict((n, {'compress_masked': False, 'penalty': p}) for n, p in zip(names, penalties))

    def add_info_list(self, name, values):
        assert len(values) == len(self.weight)
        for key, v in zip(self.weight, values):
            self._info[key][name] = v

    def get_info_list(self, name):
        return [v[name] for v in self._info.values()]

    @property
    def weight(self):
        return list(self._info.keys())

    @property
    def penalty(self):
        return self.get_info_list('penalty')

    # -- dynamic shrinkage protocol
    def compress_start(self):
        for v in self._info.values():
            v['compress_masked'] = False

    def compress_check_exist(self, info):
        return info['var_old_name'] in self._info

    def compress_mask(self, info, verbose=False):
        old, new = info['var_old_name'], info['var_new_name']
        if verbose:
            logging.info('PruneInfo compress: {} -> {}'.format(old, new))
        if self._info[old]['compress_masked']:
            raise RuntimeError('May have dependencies in compress')
        if new in self._info and self._info[new]['compress_masked']:
            raise RuntimeError('Compress {} twice'.format(new))
        self._info[new] = self._info.pop(old)   # rename onto an existing key keeps that key's position (as the reference)
        self._info[new]['compress_masked'] = True

    def compress_drop(self, info, verbose=False):
        name = info['var_old_name']
        if verbose:
            logging.info('PruneInfo drop: {}'.format(name))
        if self._info[name]['compress_masked']:
            return None
        return self._info.pop(name)


def get_bn_to_prune(model, flags, verbose=True):
    """PruneInfo over the depthwise-BN gammas of every (expanding) block; penalty = per-channel MACs normalised so that the
    atom-weighted mean is 1 (utils/prune.py:89-158).  Needs the `n_macs` stamps of utils.model_profiling."""
    bn_prune_filter = flags.get('bn_prune_filter', None)
    weights, penalties, pcf = [], [], []
    if bn_prune_filter in ('expansion_only', 'expansion_only_skip_expand1', 'equal_penalty_skip_expand1'):
        pairs = []
        for name, m in model.get_named_block_list().items():
            if not isinstance(m, mb.InvertedResidualChannels):
                continue
            if bn_prune_filter.endswith('skip_expand1') and not m.expand:
                continue
            for op, (bn_name, bn) in zip(m.ops, m.get_named_depthwise_bn(prefix=name).items()):
                hidden = bn.weight.numel()
                pairs.append((hidden, op.n_macs / hidden))
                weights.append('{}.weight'.format(bn_name))
        pcf = [v for _, v in pairs]
        if bn_prune_filter.startswith('equal_penalty'):
            penalties = [1 for _ in pairs]
        else:
            numel_total = sum(n for n, _ in pairs)
            normalizer = sum(n * v for n, v in pairs) / (numel_total + 1e-5)
            penalties = [v / normalizer for _, v in pairs]
    elif bn_prune_filter is not None:
        raise NotImplementedError()
    prune_info = PruneInfo(weights, penalties)
    prune_info.add_info_list('per_channel_flops', pcf)
    if verbose:
        for n, p in zip(prune_info.weight, prune_info.penalty):
            logging.info('{} penalty: {}'.format(n, p))
    known = set(k for k, _ in model.named_parameters())
    for n in prune_info.weight:
        assert n in known
    return prune_info


class _L1Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, mgr, table, njobs):
        out = torch.zeros(1, dtype=torch.float32, device=mgr.P.device)
        rho_ptr = mgr.hyper[ops.HYP_RHO:ops.HYP_RHO + 1]
        ops.reg_value(mgr.P, table, njobs, 1, rho_ptr, 1.0, out)
        ctx.args = (mgr, table, njobs, rho_ptr)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        mgr, table, njobs, rho_ptr = ctx.args
        ops.reg_grad(mgr.P, mgr.G, table, njobs, 1, rho_ptr, gout.reshape(1).float().contiguous())
        return None, None, None, None


def cal_bn_l1_loss(bn_weights, penalties, rho):
    """sum_i rho * penalty_i * ||gamma_i||_1 (utils/prune.py:161-167); gradient rho*penalty*sign(gamma) (sign(0) = 0) is added
    to the gradient arena in backward.  rho is staged through the device hyper-parameter vector so that the launches can be
    replayed from a hipGraph with a new value every iteration."""
    assert len(bn_weights) == len(penalties)
    if len(bn_weights) == 0:
        return 0.0
    mgr = getattr(bn_weights[0], '_atomnas_mgr', None)
    if mgr is None:
        raise ops._lib.AtomnasHipError('cal_bn_l1_loss needs arena-backed BN weights (run the model on the GPU first)')
    mgr.ensure()
    key = ('l1', tuple(id(w) for w in bn_weights), tuple(float(p) for p in penalties), mgr.version)
    cache = mgr.__dict__.setdefault('_reg_cache', {})
    if key not in cache:
        cache[key] = mgr.reg_table([(w._atomnas_off, w.numel(), p) for w, p in zip(bn_weights, penalties)])
    table, njobs = cache[key]
    if float(mgr.hyper_host[ops.HYP_RHO]) != float(rho):
        mgr.hyper_host[ops.HYP_RHO] = float(rho)
        mgr.push_hyper()
    return _L1Function.apply(mgr.anchor, mgr, table, njobs)


def alive_masks(weights, threshold, mode=0, with_index=False):
    """Alive masks of a list of arena-backed BN gammas in ONE launch (atomnas_gamma_mask; train.py:46-63, utils/prune.py:190-195):
    mode 0: |gamma| > thr, 1: |gamma| > thr OR |gamma_ema| > thr, 2: the EMA shadow only (the shadows live at the same offsets
    of the EMA arena).  The compare is the plain fp32 one torch does for `tensor > python_float`, so masks, kept counts and
    the ascending kept-channel indices are bit-exact with the reference.  Returns bool tensors (views of one byte buffer);
    with_index: also (index int32 per tensor, kept counts int32[n])."""
    import ctypes
    if len(weights) == 0:
        return ([], [], None) if with_index else []
    mgr = getattr(weights[0], '_atomnas_mgr', None)
    if mgr is None or any(getattr(w, '_atomnas_mgr', None) is not mgr for w in weights):
        raise ops._lib.AtomnasHipError('alive_masks needs arena-backed BN weights of one model (run the model on the GPU first)')
    mgr.ensure()
    if mode != 0 and mgr.EMA is None:
        raise RuntimeError('alive_masks(mode={}) needs the EMA attached to the model arenas'.format(mode))
    if mode != 0:
        # the shadows are read at the parameters' offsets of the EMA arena: a gamma without a registered shadow would read zeros
        # and be pruned as dead, where the reference raises (utils/optim.py:71-76 `average`)
        registered = set()
        for e in getattr(mgr, 'emas', []):
            registered.update(e.average_names())
        if getattr(mgr, 'emas', None):
            off2name = {slot[0]: n for n, slot in mgr.param_slots.items()}
            for w in weights:
                n = off2name.get(int(w._atomnas_off))
                if n is None or not any(r == n or r.endswith('.' + n) or n.endswith('.' + r) for r in registered):
                    raise RuntimeError('{} has not been registered'.format(n))

    class J(ctypes.Structure):
        _fields_ = [("off", ctypes.c_long), ("count", ctypes.c_int), ("out_off", ctypes.c_int)]

    jobs, pos = [], 0
    for w in weights:
        jobs.append(J(int(w._atomnas_off), int(w.numel()), pos))
        pos += int(w.numel())
    dev = mgr.P.device
    table = torch.frombuffer(bytearray(bytes((J * len(jobs))(*jobs))), dtype=torch.uint8).clone().to(dev)
    mask = torch.empty(pos, dtype=torch.uint8, device=dev)
    index = torch.empty(pos, dtype=torch.int32, device=dev)
    kept = torch.empty(len(jobs), dtype=torch.int32, device=dev)
    ops.gamma_mask(mgr.P, mgr.EMA if mode != 0 else None, table, len(jobs), threshold, mode, mask, index, kept)
    mb_ = mask.view(torch.bool)
    masks = [mb_[j.out_off:j.out_off + j.count] for j in jobs]
    if with_index:
        return masks, [index[j.out_off:j.out_off + j.count] for j in jobs], kept
    return masks


def cal_mask_network_slimming_by_threshold(weights, threshold):
    """Alive masks |gamma| > threshold (utils/prune.py:190-195); bit-exact fp32 compare.  Arena-backed gammas (the training
    path: train.py:383-386 passes the EMA model's BN weights) take one launch for all tensors; free-standing tensors, as in the
    reference's unit tests (tests/utils/prune_test.py:54-64), are compared where they live."""
    weights = list(weights)
    mgr = getattr(weights[0], '_atomnas_mgr', None) if weights else None
    if mgr is not None and all(getattr(w, '_atomnas_mgr', None) is mgr for w in weights):
        return alive_masks(weights, threshold, mode=0)
    return [w.detach().abs() > threshold for w in weights]


def cal_mask_network_slimming_by_flops(weights, prune_info, flops_to_prune, incremental=False):
    """Masks for a MACs budget (utils/prune.py:170-187; used by the reference's tests only)."""
    absw = [w.detach().abs() for w in weights]
    flat = torch.cat(absw)
    sorted_w, order = torch.sort(flat)
    flops = torch.cat([torch.full_like(w, f) for w, f in zip(absw, prune_info.get_info_list('per_channel_flops'))])[order]
    idx = torch.nonzero(torch.cumsum(flops, 0) > flops_to_prune)[0].item()
    threshold = sorted_w[idx].item()
    return [w > threshold for w in absw], threshold


def cal_pruned_flops(prune_info):
    """Total MACs of dead atoms and a per-tensor report (utils/prune.py:198-212)."""
    report, total = [], 0
    for name, pcf, mask in zip(prune_info.weight, prune_info.get_info_list('per_channel_flops'), prune_info.get_info_list('mask')):
        n_pruned = (~mask.detach()).sum().item()
        n_total = mask.numel()
        report.append([name, n_total, n_pruned, n_total * pcf, n_pruned * pcf, n_pruned / n_total])
        total += n_pruned * pcf
    return total, report


def get_rho_scheduler(prune_params, steps_per_epoch):
    """Linear warm-up of the L1 weight: 0 until epoch_free, linear to rho at epoch_warmup (utils/prune.py:215-245)."""
    free = prune_params['epoch_free'] * steps_per_epoch
    warm = prune_params['epoch_warmup'] * steps_per_epoch
    rho, stepwise = prune_params['rho'], prune_params['stepwise']
    if prune_params['scheduler'] != 'linear':
        raise ValueError('Unknown sparsity scheduler {}'.format(prune_params['scheduler']))

    def linear_fun(i):
        if not stepwise:
            i = (i // steps_per_epoch) * steps_per_epoch
        if i < free:
            return 0.0
        if i >= warm:
            return rho
        return (i - free) / (warm - free) * rho

    return linear_fun


def output_searched_network(model, infos, flags):
    """Searched-network kwargs of the supernet with dead atoms removed (utils/prune.py:248-289)."""
    setting = model.inverted_residual_setting
    blocks = list(model.get_named_block_list().values())
    kwargs = {k: getattr(model, k) for k in ['input_channel', 'last_channel', 'width_mult', 'round_nearest', 'active_fn', 'num_classes']}
    res = []
    if 'skip_expand1' in flags.get('bn_prune_filter', None):
        t, c, n, s, ks = setting[0]
        assert t == 1 and n == 1 and len(ks) == 1 and ks[0] == 3
        res.append([c, n, s, ks, [kwargs['input_channel']], False])
        blocks = blocks[n:]
    pos = 0
    for block in blocks:
        remain = []
        for k, c in zip(block.kernel_sizes, block.channels):
            info = infos[pos]
            assert c == info[1], '{}, {}, {}, {}'.format(block, k, c, str(info))
            remain.append(info[1] - info[2])
            pos += 1
        alive = [c != 0 for c in remain]
        ks_alive, ch_alive = [list(itertools.compress(x, alive)) for x in (block.kernel_sizes, remain)]
        res.append([block.output_dim, 1, block.stride, ks_alive, ch_alive, block.expand])
    assert pos == len(infos)
    kwargs['inverted_residual_setting'] = res
    return kwargs
