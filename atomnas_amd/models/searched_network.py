"""Searched-network builder (models/searched_network.py): rows are [c, n, s, ks, hiddens, expand] with explicit per-kernel
atom counts (e.g. [15, 23, 13]) -- the form a shrunk supernet is exported in and the final AtomNAS nets are defined in."""
import warnings

from torch import nn

from .mobilenet_base import ConvBNReLU, InvertedResidualChannelsFused, get_active_fn, get_block
from .mobilenet_supernet import _HipModel

__all__ = ['MobileNetSearched']


class MobileNetSearched(_HipModel):

    def __init__(self, num_classes=1000, input_size=224, input_channel=32, last_channel=1280, width_mult=1.0,
                 inverted_residual_setting=None, dropout_ratio=0.2, se_ratio=None, batch_norm_momentum=0.1,
                 batch_norm_epsilon=1e-5, active_fn='nn.ReLU6', block='InvertedResidualChannels', round_nearest=8):
        super().__init__()
        bn_kw = {'momentum': batch_norm_momentum, 'eps': batch_norm_epsilon}
        if width_mult != 1.0:
            raise ValueError('Searched model should have width 1')
        self.input_size, self.input_channel, self.last_channel = input_size, input_channel, last_channel
        self.num_classes, self.width_mult, self.round_nearest = num_classes, width_mult, round_nearest
        self.inverted_residual_setting = inverted_residual_setting
        self.active_fn, self.block, self.batch_norm_kwargs = active_fn, block, bn_kw

        if not inverted_residual_setting or len(inverted_residual_setting[0]) != 6:
            raise ValueError('inverted_residual_setting should be non-empty or a 6-element list, got {}'.format(
                inverted_residual_setting))
        if input_size % 32 != 0:
            raise ValueError('Input size must divide 32')
        for label, ch in (('Input', input_channel), ('Last', last_channel)):
            if (ch * width_mult) % round_nearest:
                warnings.warn('{} channel could not divide {}'.format(label, round_nearest))
        act = get_active_fn(active_fn)
        block_cls = get_block(block)
        extra = {}
        if se_ratio is not None:
            if not issubclass(block_cls, InvertedResidualChannelsFused):
                raise NotImplementedError('SE module not supported for block: {}'.format(block_cls))
            extra['se_ratio'] = se_ratio

        layers = [ConvBNReLU(3, input_channel, stride=2, batch_norm_kwargs=bn_kw, active_fn=act)]
        width = input_channel
        for c, n, s, ks, hiddens, expand in inverted_residual_setting:
            for rep in range(n):
                layers.append(block_cls(width, c, s if rep == 0 else 1, hiddens, ks, expand, active_fn=act,
                                        batch_norm_kwargs=bn_kw, **extra))
                width = c
        layers.append(ConvBNReLU(width, last_channel, kernel_size=1, batch_norm_kwargs=bn_kw, active_fn=act))
        layers.append(nn.AvgPool2d(input_size // 32))
        self.features = nn.Sequential(*layers)
        self.classifier = nn.Sequential(nn.Dropout(dropout_ratio), nn.Linear(last_channel, num_classes))


Model = MobileNetSearched
