"""Dynamic network shrinkage with the reference's protocol (models/compress_utils.py).

`copmress_inverted_residual_channels(m, masks, ema=, optimizer=, prune_info=, prefix=)` (the misspelling is the reference's
public name) rebuilds a block from per-branch alive masks: new (smaller) branch modules are built, and for every tensor an
`info` dict {var_old_name, var_old, type, mask, module_class, var_new_name, var_new, mask_hook} is handed to the optimizer,
the EMA and the PruneInfo in the reference's order (shared pw_bn -> kept branches -> dropped branches) before the module is
swapped.  The data movement behind `mask_hook` is the index-packed channel gather of libatomnas_hip.so
(atomnas_mask_index + atomnas_gather_dim) -- there is no host copy of the weights.  After a shrink the arenas of the owning
model are rebuilt lazily (runtime.ArenaManager.materialize) at the next forward / optimizer step.
"""
import functools
import itertools
import warnings

import torch
from torch import nn

from .. import ops
from ..utils.common import add_prefix


class _build_on(object):
    """Builds replacement modules directly on the device, without running their initialisers: every parameter and buffer of a rebuilt
    branch is overwritten by the shrink's gathers / copies (the reference builds on the host, initialises, then moves: ~630 small
    host-to-device copies and as many random initialisations per shrink of the supernet, for values that are thrown away)."""

    def __init__(self, device):
        self.dev = torch.device(device)

    def __enter__(self):
        self.ctx = self.dev
        self.ctx.__enter__()   # first: if the device context cannot be entered nothing has been patched yet
        self.saved = (nn.Conv2d.reset_parameters, nn.modules.batchnorm._NormBase.reset_parameters)
        nn.Conv2d.reset_parameters = lambda self_: None
        nn.modules.batchnorm._NormBase.reset_parameters = lambda self_: None
        return self

    def __exit__(self, *exc):
        nn.Conv2d.reset_parameters, nn.modules.batchnorm._NormBase.reset_parameters = self.saved
        self.ctx.__exit__(*exc)
        return False


def _mask_along_dim(lhs, rhs, mask, dim=0):
    """lhs <- rhs[mask] (dim 0) / rhs[:, mask] (dim 1), on device."""
    if dim not in (0, 1):
        raise NotImplementedError()
    if lhs.numel() == 0:
        return
    ops.gather_by_mask(lhs.data, rhs.data, mask, dim)


def _copy(lhs, rhs, *args):
    lhs.data.copy_(rhs.data)


def _scatter_by_bool(items, flags, pad=None):
    it = iter(items)
    out = [next(it) if f else pad for f in flags]
    assert next(it, None) is None
    return out


def build_default_info(m_new, m_old, mask, attr, mask_hook, var_type='variable', prefix_new=None, prefix_old=None):
    assert var_type in ('variable', 'buffer')
    info = {'var_old_name': add_prefix(attr, prefix_old), 'var_old': getattr(m_old, attr), 'type': var_type, 'mask': mask,
            'module_class': type(m_old)}
    if m_new is not None:
        info.update({'var_new_name': add_prefix(attr, prefix_new), 'var_new': getattr(m_new, attr), 'mask_hook': mask_hook})
    return info


def compress_conv(m_new, m_old, mask, dim, prefix_new=None, prefix_old=None):
    assert m_new is None or isinstance(m_new, nn.Conv2d)
    assert isinstance(m_old, nn.Conv2d) and dim in (0, 1)
    mk = functools.partial(build_default_info, m_new, m_old, mask, prefix_new=prefix_new, prefix_old=prefix_old)
    infos = [mk('weight', functools.partial(_mask_along_dim, dim=dim))]
    if m_old.bias is not None:
        infos.append(mk('bias', _mask_along_dim if dim == 0 else _copy))
    return infos


def compress_bn(m_new, m_old, mask, prefix_new=None, prefix_old=None):
    assert m_new is None or isinstance(m_new, nn.BatchNorm2d)
    assert isinstance(m_old, nn.BatchNorm2d)
    assert m_new is None or m_new.affine == m_old.affine
    infos = []
    if m_old.affine:
        mk = functools.partial(build_default_info, m_new, m_old, mask, prefix_new=prefix_new, prefix_old=prefix_old)
        infos += [mk('weight', _mask_along_dim), mk('bias', _mask_along_dim)]
    if m_old.track_running_stats:
        mkb = functools.partial(build_default_info, m_new, m_old, mask, var_type='buffer', prefix_new=prefix_new, prefix_old=prefix_old)
        infos += [mkb('running_var', _mask_along_dim), mkb('running_mean', _mask_along_dim), mkb('num_batches_tracked', _copy)]
    return infos


def adjust_bn(m_new, m_old, post_hook_params, **kwargs):
    """The shared pw_bn keeps every channel; the reference attaches (and then disables) a running-mean correction."""
    mask = torch.ones_like(m_new.weight, dtype=torch.bool)
    if ops.gather_deferring():
        ops.register_mask(mask, None, mask.numel())   # every channel kept: a strided copy in the shrink's job table, no index
    infos = compress_bn(m_new, m_old, mask, **kwargs)
    for info in infos:
        if 'running_mean' in info['var_old_name']:
            info['post_hook_params'] = post_hook_params
            info['post_hook'] = None  # models/compress_utils.py:201-204: warns and skips the adjustment
    return infos


def compress_conv_bn_relu(m_new, m_old, mask, prefix_new=None, prefix_old=None, dim=0):
    from . import mobilenet_base as mb
    assert m_new is None or isinstance(m_new, mb.ConvBNReLU)
    assert isinstance(m_old, mb.ConvBNReLU)
    old = list(m_old.children())
    new = [None] * len(old) if m_new is None else list(m_new.children())
    return (compress_conv(new[0], old[0], mask, dim=dim, prefix_new='{}.0'.format(prefix_new), prefix_old='{}.0'.format(prefix_old)) +
            compress_bn(new[1], old[1], mask, prefix_new='{}.1'.format(prefix_new), prefix_old='{}.1'.format(prefix_old)))


def copmress_inverted_residual_channels(m, masks, ema=None, optimizer=None, prune_info=None, prefix=None, verbose=False):
    def is_prunable_bn_var(info):
        return prune_info is not None and issubclass(info['module_class'], nn.BatchNorm2d) and info['type'] == 'variable'

    def update(infos):
        for info in infos:
            if optimizer is not None and info['type'] != 'buffer':
                optimizer.compress_mask(info, verbose=verbose)
            if ema is not None and 'num_batches_tracked' not in info['var_old_name']:
                ema.compress_mask(info, verbose=verbose)
            if is_prunable_bn_var(info) and prune_info.compress_check_exist(info):
                prune_info.compress_mask(info, verbose=verbose)
            info['mask_hook'](info['var_new'], info['var_old'], info['mask'])
            if 'post_hook' in info:
                warnings.warn('Do not adjust bn mean!!!')

    def clean(infos):
        for info in infos:
            if optimizer is not None and info['type'] != 'buffer':
                optimizer.compress_drop(info, verbose=verbose)
            if ema is not None and 'num_batches_tracked' not in info['var_old_name']:
                ema.compress_drop(info, verbose=verbose)
            if is_prunable_bn_var(info) and prune_info.compress_check_exist(info):
                prune_info.compress_drop(info, verbose=verbose)

    assert len(m.kernel_sizes) == len(masks)
    device = m.pw_bn.weight.device
    hidden = [ops.mask_count(mask) for mask in masks]   # from the shrink's one mask launch when it registered them, else mask.sum().item()
    keeps = [h > 0 for h in hidden]
    m.channels, m.kernel_sizes = [list(itertools.compress(x, keeps)) for x in (hidden, m.kernel_sizes)]
    with _build_on(device):
        new_ops, new_pw_bn = m._build(m.channels, m.kernel_sizes, m.expand)
    for mod in list(new_ops.modules()) + [new_pw_bn]:   # train / eval state follows the block
        mod.training = m.training
    idx_depth, idx_proj = (1, 2) if m.expand else (0, 1)
    new_padded = _scatter_by_bool(list(new_ops), keeps)
    new_idx_padded = _scatter_by_bool(list(range(len(new_ops))), keeps)

    keep_infos, drop_infos = [], []
    for new_op, new_i, old_op, old_i, mask in zip(new_padded, new_idx_padded, m.ops, range(len(m.ops)), masks):
        old_ch = list(old_op.children())
        new_ch = [None] * len(old_ch) if new_op is None else list(new_op.children())
        bucket = drop_infos if new_op is None else keep_infos
        if m.expand:
            bucket.append(compress_conv_bn_relu(new_ch[0], old_ch[0], mask, add_prefix('ops.{}.0'.format(new_i), prefix),
                                                add_prefix('ops.{}.0'.format(old_i), prefix)))
        bucket.append(compress_conv_bn_relu(new_ch[idx_depth], old_ch[idx_depth], mask,
                                            add_prefix('ops.{}.{}'.format(new_i, idx_depth), prefix),
                                            add_prefix('ops.{}.{}'.format(old_i, idx_depth), prefix)))
        bucket.append(compress_conv(new_ch[idx_proj], old_ch[idx_proj], mask, dim=1,
                                    prefix_new=add_prefix('ops.{}.{}'.format(new_i, idx_proj), prefix),
                                    prefix_old=add_prefix('ops.{}.{}'.format(old_i, idx_proj), prefix)))
    name_pw = add_prefix('pw_bn', prefix)
    pw_infos = adjust_bn(new_pw_bn, m.pw_bn, None, prefix_new=name_pw, prefix_old=name_pw)

    if ema is not None:
        ema.compress_start()
    if prune_info is not None:
        prune_info.compress_start()
    update(pw_infos)           # must come first for the EMA bookkeeping (reference note)
    for infos in keep_infos:
        update(infos)
    for infos in drop_infos:   # dropped branches last
        clean(infos)

    del m.ops
    del m.pw_bn
    m.ops, m.pw_bn = new_ops, new_pw_bn
    # the structure changed: the arenas of the owning model are rebuilt at the next use
    pl = getattr(m, '_plan', None)
    if pl is not None and getattr(pl, 'mgr', None) is not None:
        pl.mgr.mark_dirty()
    object.__setattr__(m, '_plan', None)
