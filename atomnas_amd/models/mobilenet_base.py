"""Atomic-block building blocks with the reference's public interface (models/mobilenet_base.py) on the HIP path.

The classes below are containers for parameters and structure: constructor signatures, attribute names, child-module
layout and therefore `state_dict()` keys are those of the reference, but `forward` dispatches to
the gfx950 kernels through `atomnas_amd.functional`.  Parameters become views into flat arenas the first time a module
(or the model that contains it) runs on the GPU -- see `atomnas_amd.runtime`.
"""
import collections
import functools
import logging
import math

import torch
from torch import nn

from .. import functional as AF
from .. import runtime
from ..utils.common import add_prefix


def _make_divisible(v, divisor, min_value=None):
    """Rounds a channel count to a multiple of `divisor` without losing more than 10% (models/mobilenet_base.py:17-31)."""
    floor = divisor if min_value is None else min_value
    rounded = (int(v + divisor / 2) // divisor) * divisor
    rounded = max(floor, rounded)
    return rounded + divisor if rounded < 0.9 * v else rounded


class Identity(nn.Module):
    def forward(self, x):
        return x


class Narrow(nn.Module):
    """`x.narrow(dimension, start, length)` as a module (used by the fused block)."""

    def __init__(self, dimension, start, length):
        super().__init__()
        self.dimension, self.start, self.length = dimension, start, length

    def forward(self, x):
        return x.narrow(self.dimension, self.start, self.length)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class HSwish(nn.Module):
    def forward(self, x):
        return x * nn.functional.relu6(x + 3.0) / 6.0


class SqueezeAndExcitation(nn.Module):
    """Squeeze-and-excitation gate (models/mobilenet_base.py:93-117).  Inside InvertedResidualChannelsFused the gate runs in the
    block's executor (functional.block_forward: atomnas_se_squeeze / se_mlp_fwd / se_scale), where the activated depthwise output it
    acts on is re-derived from the raw depthwise output instead of being materialised; called on its own it runs the same entry
    points on its input (functional.SEFunction)."""

    def __init__(self, n_feature, n_hidden, spatial_dims=[2, 3], active_fn=None):
        super().__init__()
        self.n_feature, self.n_hidden, self.spatial_dims = n_feature, n_hidden, spatial_dims
        self.se_reduce = nn.Conv2d(n_feature, n_hidden, 1, bias=True)
        self.se_expand = nn.Conv2d(n_hidden, n_feature, 1, bias=True)
        self.active_fn = active_fn()

    def forward(self, x):
        # stand-alone call (the reference's module API, :109-112); inside InvertedResidualChannelsFused the executor runs the gate
        if tuple(self.spatial_dims) != (2, 3):
            raise NotImplementedError('SqueezeAndExcitation on the HIP path squeezes over the spatial dimensions [2, 3]')
        return AF.run_se(self, x)

    def __repr__(self):
        return '{}({}, {}, spatial_dims={}, active_fn={})'.format(self._get_name(), self.n_feature, self.n_hidden,
                                                                    self.spatial_dims, self.active_fn)


class ConvBNReLU(nn.Sequential):
    """conv (no bias) -> BatchNorm2d -> activation, children `0`, `1`, `2` as in the reference (:120-142)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, groups=1, active_fn=None, batch_norm_kwargs=None):
        bn_kw = {} if batch_norm_kwargs is None else batch_norm_kwargs
        conv = nn.Conv2d(in_planes, out_planes, kernel_size, stride, (kernel_size - 1) // 2, groups=groups, bias=False)
        super().__init__(conv, nn.BatchNorm2d(out_planes, **bn_kw), active_fn())

    def forward(self, x):
        mgr = runtime.manager_of(self)
        mgr.enter()
        try:
            return AF.run_convbn(runtime.plan_of(self), x, mgr.anchor)
        finally:
            mgr.leave()


def _branch(inp, oup, hidden, k, stride, expand, active_fn, bn_kw):
    """One atomic-block branch: [1x1 expand ConvBNReLU] -> k x k depthwise ConvBNReLU -> linear 1x1 projection."""
    mods = []
    if expand:
        mods.append(ConvBNReLU(inp, hidden, kernel_size=1, active_fn=active_fn, batch_norm_kwargs=bn_kw))
    mods.append(ConvBNReLU(hidden, hidden, kernel_size=k, stride=stride, groups=hidden, active_fn=active_fn,
                           batch_norm_kwargs=bn_kw))
    mods.append(nn.Conv2d(hidden, oup, 1, 1, 0, bias=False))
    return nn.Sequential(*mods)


class InvertedResidualChannels(nn.Module):
    """MobileNetV2 block whose hidden layer is a sum of atomic branches (models/mobilenet_base.py:277-404).

    out = pw_bn(sum_i project_i(dw_i(expand_i(x)))) (+ x when stride == 1 and inp == oup); an empty `ops` is the identity.
    """

    def __init__(self, inp, oup, stride, channels, kernel_sizes, expand, active_fn=None, batch_norm_kwargs=None):
        super().__init__()
        assert stride in [1, 2]
        assert len(channels) == len(kernel_sizes)
        self.input_dim, self.output_dim = inp, oup
        self.expand, self.stride = expand, stride
        self.kernel_sizes, self.channels = kernel_sizes, channels
        self.use_res_connect = stride == 1 and inp == oup
        self.batch_norm_kwargs, self.active_fn = batch_norm_kwargs, active_fn
        self.ops, self.pw_bn = self._build(channels, kernel_sizes, expand)

    def _build(self, hidden_dims, kernel_sizes, expand):
        bn_kw = self.batch_norm_kwargs if self.batch_norm_kwargs is not None else {}
        used = 0
        ops = nn.ModuleList()
        for k, hidden in zip(kernel_sizes, hidden_dims):
            if not expand:
                if hidden != self.input_dim:
                    raise RuntimeError('a non-expanding branch must keep the input width ({} != {})'.format(hidden, self.input_dim))
                used += hidden
            ops.append(_branch(self.input_dim, self.output_dim, hidden, k, self.stride, expand, self.active_fn, bn_kw))
        if not expand and used != self.input_dim:
            raise ValueError('Part of input are not used')
        return ops, nn.BatchNorm2d(self.output_dim, **bn_kw)

    # -- names of the prunable BatchNorms (the gammas the L1 penalty acts on)
    def get_named_depthwise_bn(self, prefix=None):
        """OrderedDict name -> BatchNorm2d after each depthwise conv; names are `ops.{i}.{1 if expand else 0}.1`."""
        pos = 1 if self.expand else 0
        out = collections.OrderedDict()
        for i, op in enumerate(self.ops):
            cbr = list(op.children())[pos]
            assert isinstance(cbr, ConvBNReLU)
            bn = list(cbr.children())[1]
            assert isinstance(bn, nn.BatchNorm2d)
            out[add_prefix('ops.{}.{}.1'.format(i, pos), prefix)] = bn
        return out

    def get_depthwise_bn(self):
        return list(self.get_named_depthwise_bn().values())

    def forward(self, x):
        if len(self.ops) == 0:
            if not self.use_res_connect:
                logging.warning('The whole block is pruned without skip connection')
            return x
        mgr = runtime.manager_of(self)
        mgr.enter()
        try:
            return AF.run_block(runtime.plan_of(self), x, mgr.anchor)
        finally:
            mgr.leave()

    def __repr__(self):
        return '{}({}, {}, channels={}, kernel_sizes={}, expand={}, stride={})'.format(
            self._get_name(), self.input_dim, self.output_dim, self.channels, self.kernel_sizes, self.expand, self.stride)

    # -- dynamic shrinkage
    def compress_by_mask(self, masks, **kwargs):
        """Rebuilds the block keeping only the atoms whose mask is True (per-branch bool tensors)."""
        from . import compress_utils as cu
        cu.copmress_inverted_residual_channels(self, masks, **kwargs)

    def compress_by_threshold(self, threshold, **kwargs):
        masks = [bn.weight.detach().abs() > threshold for bn in self.get_depthwise_bn()]
        self.compress_by_mask(masks, **kwargs)


class InvertedResidualChannelsFused(nn.Module):
    """Single expand conv + per-kernel depthwise slices + optional SE + single projection (:145-274).

    Same parameter layout as the reference (`expand_conv`, `depth_ops`, `project_conv`, `se_op`: state_dict keys and shapes).
    Numerically it IS the branch block with concatenated weights, which is how the HIP executor always runs a block, so the
    forward dispatches to the same executor (functional.block_forward) with the SE stage between depthwise and projection.
    """

    def __init__(self, inp, oup, stride, channels, kernel_sizes, expand, active_fn=None, batch_norm_kwargs=None,
                 se_ratio=None):
        super().__init__()
        assert stride in [1, 2]
        assert len(channels) == len(kernel_sizes)
        self.input_dim, self.output_dim = inp, oup
        self.expand, self.stride = expand, stride
        self.kernel_sizes, self.channels = kernel_sizes, channels
        self.use_res_connect = stride == 1 and inp == oup
        self.batch_norm_kwargs, self.active_fn, self.se_ratio = batch_norm_kwargs, active_fn, se_ratio
        self.expand_conv, self.depth_ops, self.project_conv, self.se_op = self._build(channels, kernel_sizes, expand, se_ratio)

    def _build(self, hidden_dims, kernel_sizes, expand, se_ratio):
        bn_kw = self.batch_norm_kwargs if self.batch_norm_kwargs is not None else {}
        total = sum(hidden_dims)
        expand_conv = (ConvBNReLU(self.input_dim, total, kernel_size=1, batch_norm_kwargs=bn_kw, active_fn=self.active_fn)
                       if self.expand else Identity())
        depth_ops = nn.ModuleList()
        start = 0
        for k, hidden in zip(kernel_sizes, hidden_dims):
            layers = []
            if expand:
                layers.append(Narrow(1, start, hidden))
                start += hidden
            elif hidden != self.input_dim:
                raise RuntimeError('a non-expanding branch must keep the input width')
            layers.append(ConvBNReLU(hidden, hidden, kernel_size=k, stride=self.stride, groups=hidden, batch_norm_kwargs=bn_kw,
                                     active_fn=self.active_fn))
            depth_ops.append(nn.Sequential(*layers))
        if expand and start != total:
            raise ValueError('Part of expanded are not used')
        project_conv = nn.Sequential(nn.Conv2d(total, self.output_dim, 1, 1, 0, bias=False),
                                     nn.BatchNorm2d(self.output_dim, **bn_kw))
        se_op = (SqueezeAndExcitation(total, int(round(self.input_dim * se_ratio)), active_fn=self.active_fn)
                 if se_ratio is not None else Identity())
        return expand_conv, depth_ops, project_conv, se_op

    def get_named_depthwise_bn(self, prefix=None):
        if not self.expand:
            raise RuntimeError('Not search_first')
        out = collections.OrderedDict()
        for i, op in enumerate(self.depth_ops):
            bn = list(list(op.children())[1].children())[1]
            out[add_prefix('depth_ops.{}.1.1'.format(i), prefix)] = bn
        return out

    def get_depthwise_bn(self):
        return list(self.get_named_depthwise_bn().values())

    def forward(self, x):
        mgr = runtime.manager_of(self)
        mgr.enter()
        try:
            return AF.run_block(runtime.plan_of(self), x, mgr.anchor)
        finally:
            mgr.leave()

    def __repr__(self):
        return '{}({}, {}, channels={}, kernel_sizes={}, expand={}, stride={}, se_ratio={})'.format(
            self._get_name(), self.input_dim, self.output_dim, self.channels, self.kernel_sizes, self.expand, self.stride,
            self.se_ratio)


_ACTIVATIONS = {
    'nn.ReLU6': functools.partial(nn.ReLU6, inplace=True),
    'nn.ReLU': functools.partial(nn.ReLU, inplace=True),
    'nn.Swish': Swish,
    'nn.HSwish': HSwish,
}
_BLOCKS = {
    'InvertedResidualChannels': InvertedResidualChannels,
    'InvertedResidualChannelsFused': InvertedResidualChannelsFused,
}


def get_active_fn(name):
    return _ACTIVATIONS[name]


def get_block(name):
    return _BLOCKS[name]


def init_weights_slimmable(m):
    """Slimmable-networks initialisation (models/mobilenet_base.py:426-437)."""
    if isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode='fan_out')
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Linear):
        nn.init.normal_(m.weight, 0, 0.01)
        nn.init.zeros_(m.bias)


def init_weights_mnas(m):
    """MnasNet initialisation (:440-459): conv ~ N(0, sqrt(2/fan_out)) with fan_out = k*k for depthwise, BN (1, 0),
    linear ~ U(+-1/sqrt(out_features)) with zero bias."""
    if isinstance(m, nn.Conv2d):
        if m.groups == m.in_channels:
            fan_out = m.kernel_size[0] * m.kernel_size[1]
        else:
            fan_out = m.out_channels * m.kernel_size[0] * m.kernel_size[1]
        nn.init.normal_(m.weight, 0.0, math.sqrt(2.0) / math.sqrt(fan_out))
        if m.bias is not None:
            nn.init.zeros_(m.bias)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)
    elif isinstance(m, nn.Linear):
        bound = 1.0 / math.sqrt(m.out_features)
        nn.init.uniform_(m.weight, -bound, bound)
        nn.init.zeros_(m.bias)


def output_network(model):
    """Model kwargs in `searched_network` row format [c, n, s, ks, hiddens, expand] (:462-479); this is what a shrunk
    supernet is exported as (checkpoint `.yml`)."""
    kwargs = {key: getattr(model, key) for key in
              ['input_channel', 'last_channel', 'width_mult', 'round_nearest', 'active_fn', 'num_classes']}
    kwargs['inverted_residual_setting'] = [[b.output_dim, 1, b.stride, b.kernel_sizes, b.channels, b.expand]
                                           for b in model.get_named_block_list().values()]
    return kwargs


def _get_named_block_list(m):
    """`features.N` -> block for the inverted-residual blocks (everything between the stem and the last two entries)."""
    children = list(m.features.named_children())[1:-2]
    return collections.OrderedDict(('features.{}'.format(n), b) for n, b in children)
