"""Analytic MAC / parameter stamps with the reference's attribute names (utils/model_profiling.py).

The reference obtains `n_macs` by running a hooked forward pass; the numbers only depend on shapes, so they are computed
here in closed form (conv: cin*cout*kh*kw*ho*wo/groups*batch, :82-88; Linear; AvgPool2d = its input numel; blocks = sum of
their branch Sequentials, :121-127).  `op.n_macs` of every branch feeds the resource-aware L1 penalty (utils/prune.py:117)
and `model.n_macs` the shrink log (train.py:75-80).
"""
import logging

from torch import nn

from ..models import mobilenet_base as mb


def _conv_out(h, k, s, p):
    return (h + 2 * p - k) // s + 1


def _stamp_conv(m, h, w, batch):
    ho = _conv_out(h, m.kernel_size[0], m.stride[0], m.padding[0])
    wo = _conv_out(w, m.kernel_size[1], m.stride[1], m.padding[1])
    m.n_macs = (m.in_channels * m.out_channels * m.kernel_size[0] * m.kernel_size[1] * ho * wo // m.groups) * batch
    m.n_params = sum(p.numel() for p in m.parameters())
    m.n_seconds = 0
    return ho, wo


def _stamp_seq(m, h, w, batch):
    """Sequential-like container: children applied in order; n_macs = sum of children (the reference's generic branch)."""
    total_m, total_p = 0, 0
    for c in m.children():
        h, w = _stamp(c, h, w, batch)
        total_m += getattr(c, 'n_macs', 0)
        total_p += getattr(c, 'n_params', 0)
    m.n_macs, m.n_params, m.n_seconds = total_m, total_p, 0
    return h, w


def _stamp(m, h, w, batch):
    if isinstance(m, nn.Conv2d):
        return _stamp_conv(m, h, w, batch)
    if isinstance(m, nn.Linear):
        m.n_macs = m.in_features * m.out_features * batch
        m.n_params = sum(p.numel() for p in m.parameters())
        m.n_seconds = 0
        return h, w
    if isinstance(m, (nn.AvgPool2d, nn.AdaptiveAvgPool2d)):
        m.n_macs = m._profiling_channels * h * w * batch
        m.n_params, m.n_seconds = 0, 0
        k = m.kernel_size if isinstance(m.kernel_size, int) else m.kernel_size[0]
        return h // k, w // k
    if isinstance(m, mb.InvertedResidualChannels):
        m.n_macs = m.n_params = m.n_seconds = 0
        ho, wo = h, w
        for op in m.ops:
            ho, wo = _stamp_seq(op, h, w, batch)
            m.n_macs += op.n_macs
            m.n_params += op.n_params
        _stamp(m.pw_bn, ho, wo, batch)
        if len(m.ops) == 0:
            ho, wo = h, w
        return ho, wo
    if isinstance(m, mb.SqueezeAndExcitation):   # :121-129: the gate's input numel + the two 1x1 convs on the pooled vector
        _stamp_conv(m.se_reduce, 1, 1, batch)
        _stamp_conv(m.se_expand, 1, 1, batch)
        m.n_macs = m.n_feature * h * w * batch + m.se_reduce.n_macs + m.se_expand.n_macs
        m.n_params = m.se_reduce.n_params + m.se_expand.n_params
        m.n_seconds = 0
        return h, w
    if isinstance(m, mb.InvertedResidualChannelsFused):   # :138-147: depth_ops + expand_conv + project_conv + se_op
        m.n_macs = m.n_params = m.n_seconds = 0
        _stamp(m.expand_conv, h, w, batch)
        ho, wo = h, w
        for op in m.depth_ops:
            ho, wo = _stamp_seq(op, h, w, batch)
        _stamp(m.se_op, ho, wo, batch)
        _stamp_seq(m.project_conv, ho, wo, batch)
        for sub in list(m.depth_ops) + [m.expand_conv, m.project_conv, m.se_op]:
            m.n_macs += getattr(sub, 'n_macs', 0)
            m.n_params += getattr(sub, 'n_params', 0)
        return ho, wo
    if len(list(m.children())) > 0:
        return _stamp_seq(m, h, w, batch)
    m.n_macs = m.n_params = m.n_seconds = 0   # BN, activations, dropout: zero-cost leaves in the reference's table
    if isinstance(m, nn.BatchNorm2d):
        m.n_params = 0
    return h, w


def model_profiling(model, height, width, batch=1, channel=3, use_cuda=True, num_forwards=0, verbose=True):
    """Stamps n_macs / n_params on every module and returns (model.n_macs, model.n_params).  num_forwards is accepted for
    signature compatibility; nothing is executed."""
    feats = list(model.features.children())
    last_conv = list(feats[-2].children())[0]
    for f in feats:
        if isinstance(f, (nn.AvgPool2d, nn.AdaptiveAvgPool2d)):
            f._profiling_channels = last_conv.out_channels
    h, w = _stamp_seq(model.features, height, width, batch)
    _stamp_seq(model.classifier, 1, 1, batch)
    model.n_macs = model.features.n_macs + model.classifier.n_macs
    model.n_params = model.features.n_params + model.classifier.n_params   # conv / linear parameters, as the reference's table
    model.n_seconds = 0
    if verbose:
        logging.info('Total params {:,} macs {:,}'.format(model.n_params, model.n_macs))
    return model.n_macs, model.n_params
