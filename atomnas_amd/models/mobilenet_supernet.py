"""Supernet builder with the reference's interface (models/mobilenet_supernet.py): `Model(**model_kwparams, input_size=...)`.

Rows of `inverted_residual_setting` are [expand ratio t, output channels c, repeats n, first stride s, kernel sizes ks];
every repeat becomes one InvertedResidualChannels block whose branches (one per kernel size) have round(inp * t) atoms.
"""
import numbers

import torch
from torch import nn

from .. import functional as AF
from .. import runtime
from .mobilenet_base import ConvBNReLU, _get_named_block_list, _make_divisible, get_active_fn, get_block

__all__ = ['MobileNetV2']


def get_block_wrapper(block_str):
    """Block class that takes an expand ratio (number or per-branch list) instead of explicit hidden widths (:14-55)."""
    base = get_block(block_str)

    class InvertedResidual(base):

        def __init__(self, inp, oup, stride, expand_ratio, kernel_sizes, active_fn=None, batch_norm_kwargs=None):
            if isinstance(expand_ratio, list):
                assert len(expand_ratio) == len(kernel_sizes)
                ratios, expand = expand_ratio, True
            elif isinstance(expand_ratio, numbers.Number):
                ratios, expand = [expand_ratio] * len(kernel_sizes), expand_ratio != 1
            else:
                raise ValueError('Unknown expand_ratio type: {}'.format(expand_ratio))
            hidden = [int(round(inp * r)) for r in ratios]
            super().__init__(inp, oup, stride, hidden, kernel_sizes, expand, active_fn=active_fn,
                             batch_norm_kwargs=batch_norm_kwargs)
            self.expand_ratio = ratios if isinstance(expand_ratio, list) else ratios

    return InvertedResidual


class _HipModel(nn.Module):
    """Shared forward of the supernet and searched-network containers: stem -> blocks -> fused tail."""

    compute_dtype = torch.bfloat16
    dropout_seed = 1995

    def set_compute_dtype(self, dtype):
        """bf16 (default) or fp32 activation storage; fp32 is the tight-tolerance parity mode."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError('compute dtype must be float32 or bfloat16')
        self.compute_dtype = dtype
        mgr = getattr(self, '_arena', None)
        if mgr is not None:
            mgr.mark_dirty()
        return self

    def get_named_block_list(self):
        return _get_named_block_list(self)

    def forward(self, x, loss_args=None):
        """loss_args (engine.TrainStep only): (target, label_smoothing, loss_vec, topk, loss_out) -- the tail and the
        label-smoothed cross entropy then run as one autograd node and the scalar mean loss is returned instead of the logits."""
        if not x.is_cuda:
            raise AF.ops._lib.AtomnasHipError('this model runs on the GPU through libatomnas_hip.so only (input is on %s)' % x.device)
        mgr = runtime.manager_of(self)
        mgr.enter()
        try:
            feats = list(self.features.children())
            y = feats[0](x)
            for blk in feats[1:-2]:
                y = blk(y)
            last, pool = feats[-2], feats[-1]
            drop, fc = list(self.classifier.children())
            k = pool.kernel_size if isinstance(pool.kernel_size, int) else pool.kernel_size[0]
            if not (isinstance(last, ConvBNReLU) and isinstance(pool, nn.AvgPool2d) and y.shape[2] == k and y.shape[3] == k):
                raise NotImplementedError('the tail must be 1x1 ConvBNReLU -> global AvgPool2d -> Dropout -> Linear')
            if loss_args is not None:
                target, eps, loss_vec, topk, loss_out = loss_args
                return AF.TailLossFunction.apply(y, mgr.anchor, runtime.plan_of(last), runtime.plan_of(fc), drop.p,
                                                 self.training and drop.training, self.dropout_seed, mgr.step_counter, target,
                                                 float(eps), loss_vec, topk, loss_out)
            return AF.run_tail(runtime.plan_of(last), runtime.plan_of(fc), y, mgr.anchor, drop.p, self.training and drop.training,
                               self.dropout_seed, mgr.step_counter)
        finally:
            mgr.leave()


class MobileNetV2(_HipModel):
    """MobileNetV2-like supernet (models/mobilenet_supernet.py:58-173)."""

    def __init__(self, num_classes=1000, input_size=224, input_channel=32, last_channel=1280, width_mult=1.0,
                 inverted_residual_setting=None, dropout_ratio=0.2, batch_norm_momentum=0.1, batch_norm_epsilon=1e-5,
                 active_fn='nn.ReLU6', block='InvertedResidualChannels', round_nearest=8):
        super().__init__()
        bn_kw = {'momentum': batch_norm_momentum, 'eps': batch_norm_epsilon}
        self.input_size, self.input_channel, self.last_channel = input_size, input_channel, last_channel
        self.num_classes, self.width_mult, self.round_nearest = num_classes, width_mult, round_nearest
        self.inverted_residual_setting = inverted_residual_setting
        self.active_fn, self.block, self.batch_norm_kwargs = active_fn, block, bn_kw

        if not inverted_residual_setting or len(inverted_residual_setting[0]) != 5:
            raise ValueError('inverted_residual_setting should be non-empty or a 5-element list, got {}'.format(
                inverted_residual_setting))
        if input_size % 32 != 0:
            raise ValueError('Input size must divide 32')
        act = get_active_fn(active_fn)
        block_cls = get_block_wrapper(block)

        width = _make_divisible(input_channel * width_mult, round_nearest)
        final = _make_divisible(last_channel * max(1.0, width_mult), round_nearest)
        layers = [ConvBNReLU(3, width, stride=2, batch_norm_kwargs=bn_kw, active_fn=act)]
        for t, c, n, s, ks in inverted_residual_setting:
            out_c = _make_divisible(c * width_mult, round_nearest)
            for rep in range(n):
                layers.append(block_cls(width, out_c, s if rep == 0 else 1, t, ks, active_fn=act, batch_norm_kwargs=bn_kw))
                width = out_c
        layers.append(ConvBNReLU(width, final, kernel_size=1, batch_norm_kwargs=bn_kw, active_fn=act))
        layers.append(nn.AvgPool2d(input_size // 32))
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(nn.Dropout(dropout_ratio), nn.Linear(final, num_classes))


Model = MobileNetV2
