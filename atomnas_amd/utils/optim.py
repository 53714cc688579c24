"""Optimisation utilities with the reference's interface (utils/optim.py): label-smoothed CE, L2 regulariser, EMA,
learning-rate schedule, optimizer factory -- as O(1) launches over the arenas of `atomnas_amd.runtime`."""
import torch
from torch import nn

from .. import functional as AF


class CrossEntropyLabelSmooth(nn.Module):
    """Label-smoothed cross entropy (utils/optim.py:180-207) on device; also counts top-1/top-5 hits of the batch into
    `self.topk_correct` (int32[2]) so that forward_loss needs no host synchronisation (common.py:70-79 does two)."""

    def __init__(self, num_classes, label_smoothing, reduction='none'):
        super().__init__()
        self.num_classes, self.label_smoothing = num_classes, label_smoothing
        if reduction not in ('none', 'mean', 'sum'):
            raise ValueError('Unknown reduction: {}'.format(reduction))
        self.reduction = reduction
        self.topk_correct = None

    def forward(self, inputs, targets):
        assert inputs.size(1) == self.num_classes
        if self.topk_correct is None or self.topk_correct.device != inputs.device:
            self.topk_correct = torch.zeros(2, dtype=torch.int32, device=inputs.device)
        loss = AF.CESmoothFunction.apply(inputs, targets, float(self.label_smoothing), self.topk_correct)
        if self.reduction == 'mean':
            return loss.mean()
        if self.reduction == 'sum':
            return loss.sum()
        return loss


# ---------------------------------------------------------------------------------------------- L2 regulariser
class _RegFunction(torch.autograd.Function):
    """Scalar regulariser over an arena job table: value in forward, gradient contribution added to the gradient arena in
    backward (so p.grad after backward() equals what the reference's autograd leaves there)."""

    @staticmethod
    def forward(ctx, anchor, mgr, table, njobs, use_abs, mult_ptr, post_scale):
        from .. import ops
        out = torch.zeros(1, dtype=torch.float32, device=mgr.P.device)
        ops.reg_value(mgr.P, table, njobs, use_abs, mult_ptr, post_scale, out)
        ctx.args = (mgr, table, njobs, use_abs, mult_ptr)
        return out[0]

    @staticmethod
    def backward(ctx, gout):
        from .. import ops
        mgr, table, njobs, use_abs, mult_ptr = ctx.args
        go = gout.reshape(1).float().contiguous()
        ops.reg_grad(mgr.P, mgr.G, table, njobs, use_abs, mult_ptr, go)
        return None, None, None, None, None, None, None


def cal_l2_loss(model, weight_decay, method):
    """0.5 * sum_w wd_w * ||w||^2 (utils/optim.py:210-249).  'mnas': every conv / fc weight and the classifier bias;
    'slimmable': dense conv and fc weights only.  One launch forward, one backward (the reference spends ~2k + ~3k ATen
    dispatches on this and the L1 term every step)."""
    from .. import runtime
    mgr = runtime.manager_of(model)
    mgr.ensure()
    if method == 'mnas':
        kinds = ('dense', 'dw', 'fc', 'fcbias')
    elif method == 'slimmable':
        kinds = ('dense', 'fc')
    elif method == 'mnas_no_bias':
        raise NotImplementedError()
    else:
        raise ValueError('Unknown weight_decay method: {}'.format(method))
    key = ('l2', method, float(weight_decay), mgr.version)
    cache = mgr.__dict__.setdefault('_reg_cache', {})
    if key not in cache:
        entries = [(off, n, weight_decay) for kind, off, n in mgr.reg_slots if kind in kinds]
        cache[key] = mgr.reg_table(entries)
    table, njobs = cache[key]
    return _RegFunction.apply(mgr.anchor, mgr, table, njobs, 0, None, 0.5)


# ---------------------------------------------------------------------------------------------- EMA
class ExponentialMovingAverage(nn.Module):
    """tf.train.ExponentialMovingAverage as used by the reference (utils/optim.py:15-177), with the shadows living in the
    EMA arenas once the owning model is materialised.  `update_all(num_updates)` is the fused form of the reference's
    per-name loop (train.py:231-236): two launches for 799 tensors."""

    def __init__(self, momentum, zero_debias=False):
        if zero_debias:
            raise NotImplementedError('zero_debias')
        if momentum < 0.0 or momentum > 1.0:
            raise ValueError('Invalid momentum value: {}'.format(momentum))
        super().__init__()
        self._momentum, self._zero_debias = momentum, zero_debias
        self._mgr = None
        self.clear()

    def clear(self):
        from collections import OrderedDict
        self._shadow, self._info = OrderedDict(), OrderedDict()

    def _check_exist(self, name):
        if name not in self._shadow:
            raise RuntimeError('{} has not been registered'.format(name))

    def register(self, name, val, zero_init=False):
        if name in self._shadow:
            raise ValueError('Should not register twice for {}'.format(name))
        if val.dtype not in (torch.float16, torch.float32, torch.float64):
            raise TypeError('The variables must be half, float, or double: {}'.format(name))
        self._shadow[name] = torch.zeros_like(val) if zero_init else val.detach().clone()
        self._info[name] = {'num_updates': 0, 'last_momemtum': None, 'zero_init': zero_init, 'compress_masked': False}
        mgr = getattr(val, '_atomnas_mgr', None)
        if mgr is not None:
            self.attach(mgr)

    def attach(self, mgr):
        """Moves the shadows into the model's EMA arenas (done at the next materialisation)."""
        if self._mgr is not mgr:
            self._mgr = mgr
            mgr.attach_ema(self)

    def _on_materialize(self, mgr):
        self._mgr = mgr

    def momentum_at(self, num_updates):
        if num_updates is None:
            return self._momentum
        return min(self._momentum, (1.0 + num_updates) / (10.0 + num_updates))

    def forward(self, name, x, num_updates=None):
        """Per-variable update (reference API).  Prefer update_all() on the hot path."""
        self._check_exist(name)
        m = self.momentum_at(num_updates)
        self._info[name]['num_updates'] += 1
        self._info[name]['last_momemtum'] = m
        return self._shadow[name].mul_(m).add_(x.detach(), alpha=1.0 - m)

    def update_all(self, num_updates=None, push=True):
        """shadow = d*shadow + (1-d)*value for every registered parameter and BN running statistic, in two launches."""
        from .. import ops
        mgr = self._mgr
        if mgr is None:
            raise ops._lib.AtomnasHipError('EMA.update_all needs arena-backed variables (attach(manager) / register model tensors)')
        mgr.ensure()
        m = self.momentum_at(num_updates)
        if push:
            mgr.hyper_host[ops.HYP_EMA_DECAY] = m
            mgr.push_hyper()
        self.launch(mgr)
        for info in self._info.values():
            info['num_updates'] += 1
            info['last_momemtum'] = m

    def note_updates(self, n, momentum):
        """host-side counters of the reference's per-variable `_info` for updates that ran inside a replayed graph"""
        for info in self._info.values():
            info['num_updates'] += n
            info['last_momemtum'] = momentum

    def launch(self, mgr):
        from .. import ops
        ops.ema_update(mgr.EMA, mgr.P, mgr.nP, mgr.hyper)
        ops.ema_update(mgr.SEMA, mgr.S, mgr.nS, mgr.hyper)

    def pop(self, name):
        self._check_exist(name)
        return self._shadow.pop(name), self._info.pop(name)

    def average_names(self):
        return list(self._shadow.keys())

    def average(self, name):
        self._check_exist(name)
        return self._shadow[name]

    def state_dict(self):
        return {'info': self._info, 'shadow': {k: v.detach().clone() for k, v in self._shadow.items()},
                'param': {'momentum': self._momentum, 'zero_debias': self._zero_debias}}

    def load_state_dict(self, state_dict):
        import copy
        import logging
        import warnings
        for key, val in state_dict['param'].items():
            cur = getattr(self, '_{}'.format(key))
            if val != cur:
                msg = 'EMA {} mismatch: current {} vs previous {}'.format(key, cur, val)
                warnings.warn(msg, RuntimeWarning)
                logging.warning(msg)
        self._info = copy.deepcopy(state_dict['info'])
        # the checkpoint REPLACES the shadow set (utils/optim.py:98-117 deep-copies state['shadow']): entries the checkpoint
        # does not hold (e.g. branches a resumed, already shrunk run no longer has) must not survive next to the new `_info`
        stale = [k for k in self._shadow if k not in state_dict['shadow']]
        for k in stale:
            del self._shadow[k]
        if stale and self._mgr is not None:
            self._mgr.mark_dirty()
        for k, v in state_dict['shadow'].items():
            if k in self._shadow and self._shadow[k].shape == v.shape:
                self._shadow[k].copy_(v)
            else:
                self._shadow[k] = v.detach().clone()
                if self._mgr is not None:
                    self._mgr.mark_dirty()

    def to(self, *args, **kwargs):
        device, dtype, non_blocking = torch._C._nn._parse_to(*args, **kwargs)[:3]
        for k in list(self._shadow.keys()):
            v = self._shadow[k]
            self._shadow[k] = v.to(device, dtype if v.is_floating_point() else None, non_blocking)
        return self

    # ---- dynamic shrinkage protocol (utils/optim.py:119-165)
    def compress_start(self):
        for val in self._info.values():
            val['compress_masked'] = False

    def compress_mask(self, info, verbose=False):
        import logging
        old, new = info['var_old_name'], info['var_new_name']
        if verbose:
            logging.info('EMA compress: {} -> {}'.format(old, new))
        if self._info[old]['compress_masked']:
            raise RuntimeError('May have dependencies in compress')
        if new in self._info and self._info[new]['compress_masked']:
            raise RuntimeError('Compress {} twice'.format(new))
        ema_old = self._shadow.pop(old)
        ema_new = torch.zeros_like(info['var_new'], device=ema_old.device)
        info['mask_hook'](ema_new, ema_old, info['mask'])
        self._info[new] = self._info.pop(old)
        self._info[new]['compress_masked'] = True
        self._shadow[new] = ema_new
        if self._mgr is not None:
            self._mgr.mark_dirty()

    def compress_drop(self, info, verbose=False):
        import logging
        name = info['var_old_name']
        if verbose:
            logging.info('EMA drop: {}'.format(name))
        self._check_exist(name)
        if self._info[name]['compress_masked']:
            return None
        if self._mgr is not None:
            self._mgr.mark_dirty()
        return self.pop(name)

    @staticmethod
    def adjust_momentum(momentum, steps_multi):
        return momentum ** (1.0 / steps_multi)


# ---------------------------------------------------------------------------------------------- schedules / factories
def get_lr_scheduler(optimizer, FLAGS):
    """Per-iteration LambdaLR of the reference (utils/optim.py:252-306): linear warm-up from base_lr when lr > base_lr,
    then exponential decay (staircase unless lr_stepwise), optionally truncated; 'linear_decaying'; 'multistep'."""
    import functools
    stepwise = FLAGS.get('lr_stepwise', True)
    steps_per_epoch = FLAGS._steps_per_epoch
    warmup_iterations = FLAGS.get('epoch_warmup', 5) * steps_per_epoch
    use_warmup = FLAGS.lr > FLAGS.base_lr

    def with_warmup(fn, i):
        if use_warmup and i <= warmup_iterations:
            r = FLAGS.base_lr / FLAGS.lr
            return r + i / warmup_iterations * (1 - r)
        return fn(i)

    name = FLAGS.lr_scheduler
    if name == 'multistep':
        if use_warmup:
            raise NotImplementedError('Warmup not implemented for multistep')
        return torch.optim.lr_scheduler.MultiStepLR(optimizer, milestones=[steps_per_epoch * v for v in FLAGS.multistep_lr_milestones],
                                                    gamma=FLAGS.multistep_lr_gamma)
    if name in ('exp_decaying', 'exp_decaying_trunc'):
        floor = 0.05 if 'trunc' in name else 0.0
        interval = steps_per_epoch * FLAGS.exp_decay_epoch_interval

        def decay(i):
            if not stepwise:
                i = (i // interval) * interval
            return max(FLAGS.exp_decaying_lr_gamma ** (i / interval), floor)

        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=functools.partial(with_warmup, decay))
    if name == 'linear_decaying':
        assert stepwise
        total = FLAGS.num_epochs * steps_per_epoch
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=functools.partial(with_warmup, lambda i: 1 - i / total))
    raise NotImplementedError('Learning rate scheduler {} is not yet implemented.'.format(name))


def get_optimizer(model, FLAGS):
    """Optimizer factory (utils/optim.py:309-332); 'rmsprop' is the fused arena optimizer."""
    import importlib

    from .rmsprop import RMSprop
    if FLAGS.optimizer == 'rmsprop':
        return RMSprop(model.parameters(), lr=FLAGS.lr, alpha=FLAGS.alpha, momentum=FLAGS.momentum, eps=FLAGS.epsilon,
                       eps_inside_sqrt=FLAGS.eps_inside_sqrt, weight_decay=0)
    if FLAGS.optimizer == 'sgd':
        raise NotImplementedError('SGD is outside the AtomNAS search hot path (apps/slimming/shrink/*.yml use rmsprop)')
    try:
        return importlib.import_module(FLAGS.optimizer).get_optimizer(model)
    except ImportError:
        raise NotImplementedError('Optimizer {} is not yet implemented.'.format(FLAGS.optimizer))
