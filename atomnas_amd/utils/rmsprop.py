"""TF-flavoured RMSprop with the reference's interface (utils/rmsprop.py) as one fused launch over the parameter arena.

`RMSprop(model.parameters(), lr, alpha, momentum, eps, eps_inside_sqrt, weight_decay=0)`; `step()` reads `p.grad`
(views into the gradient arena) and updates every parameter, `square_avg` and `momentum_buffer` in a single kernel
(atomnas_fused_rmsprop_ema).
`state[p]['square_avg']` / `['momentum_buffer']` remain per-parameter tensors (arena views) so that `state_dict()` keeps the
reference's format, and `compress_mask` / `compress_drop` keep the reference's re-keying protocol for dynamic shrinkage.
"""
import logging

import torch
from torch.optim.optimizer import Optimizer

from .. import ops
from .common import check_tensor_in, index_tensor_in


class RMSprop(Optimizer):

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, eps_inside_sqrt=False, weight_decay=0, momentum=0, centered=False):
        for label, v in (('learning rate', lr), ('epsilon value', eps), ('momentum value', momentum),
                         ('weight_decay value', weight_decay), ('alpha value', alpha)):
            if not 0.0 <= v:
                raise ValueError('Invalid {}: {}'.format(label, v))
        if centered:
            raise NotImplementedError('centered RMSprop is not on the AtomNAS hot path')
        if weight_decay != 0:
            raise NotImplementedError('weight decay enters through cal_l2_loss (utils/optim.py), as in the reference configs')
        defaults = dict(lr=lr, momentum=momentum, alpha=alpha, eps=eps, eps_inside_sqrt=eps_inside_sqrt, centered=centered,
                        weight_decay=weight_decay)
        super().__init__(params, defaults)
        if len(self.param_groups) != 1:
            raise NotImplementedError('one parameter group (the reference never uses more)')
        self._mgr = None
        self._steps = 0

    # ---- arena plumbing
    def _manager(self):
        params = self.param_groups[0]['params']
        mgr = None
        for p in params:
            m = getattr(p, '_atomnas_mgr', None)
            if m is not None:
                mgr = m
                break
        if mgr is None:
            raise ops._lib.AtomnasHipError(
                'RMSprop.step needs arena-backed parameters: run the model on the GPU once (or call '
                'atomnas_amd.runtime.manager_of(model).ensure()) before the first step; there is no CPU fallback')
        if self._mgr is not mgr:
            self._mgr = mgr
            mgr.attach_optimizer(self)
        mgr.ensure()
        return mgr

    def _on_materialize(self, mgr):
        """Called by the arena manager after (re)building arenas: make sure every parameter has its state views."""
        group = self.param_groups[0]
        for p in group['params']:
            off = getattr(p, '_atomnas_off', None)
            if off is None or getattr(p, '_atomnas_mgr', None) is not mgr:
                raise RuntimeError('optimizer holds a parameter that is not part of the model arena')
            st = self.state[p]
            shape, strides = tuple(p.shape), (tuple(p.stride()) if not p.is_contiguous() else None)
            if 'square_avg' not in st:
                st['step'] = self._steps
                st['square_avg'] = _view(mgr.SQ, off, shape, strides)
                if group['momentum'] > 0:
                    st['momentum_buffer'] = _view(mgr.BUF, off, shape, strides)

    def load_state_dict(self, state_dict):
        """torch's index-ordered format (what utils/common.py:123-137 saves).  The loaded state tensors are ordinary tensors; they
        move into the optimizer arenas at the next materialisation."""
        super().load_state_dict(state_dict)
        if self._mgr is not None:
            self._mgr.mark_dirty()

    def zero_grad(self, set_to_none=False):
        """Gradients live in one arena: a single memset (the kernels accumulate into it during backward)."""
        mgr = self._mgr or self._manager()
        mgr.ensure()
        mgr.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        mgr = self._manager()
        group = self.param_groups[0]
        mgr.hyper_host[ops.HYP_LR] = float(group['lr'])
        mgr.push_hyper()
        self.launch(mgr)
        self._steps += 1
        return loss

    def launch(self, mgr):
        """The device part of step(): capturable into a hipGraph (reads lr / EMA decay from mgr.hyper)."""
        group = self.param_groups[0]
        ops.fused_rmsprop_ema(mgr.P, mgr.G, mgr.SQ, mgr.BUF if group['momentum'] > 0 else None, None, None, mgr.nP, mgr.hyper,
                              group['alpha'], group['eps'], group['eps_inside_sqrt'], group['momentum'])

    # ---- dynamic shrinkage protocol (utils/rmsprop.py:134-182)
    def compress_mask(self, info, verbose=False):
        var_old, var_new, mask_hook, mask = info['var_old'], info['var_new'], info['mask_hook'], info['mask']
        if verbose:
            logging.info('RMSProp compress: {} -> {}'.format(info['var_old_name'], info['var_new_name']))
        for group in self.param_groups:
            index = index_tensor_in(var_old, group['params'], raise_error=False)
            if index is None:
                continue
            if check_tensor_in(var_old, self.state):
                state = self.state.pop(var_old)
                if len(state) != 0:
                    new_state = {'step': state['step']}
                    for key in ('square_avg', 'momentum_buffer', 'grad_avg'):
                        if key in state:
                            new_state[key] = torch.zeros_like(var_new.data, device=var_old.device)
                            mask_hook(new_state[key], state[key], mask)
                    self.state[var_new] = new_state
            del group['params'][index]
            group['params'].append(var_new)  # appended, as in the reference: optimizer order != model order after a shrink
            if self._mgr is not None:
                self._mgr.mark_dirty()
            return
        raise AssertionError('Var: {} not in RMSProp'.format(info['var_old_name']))

    def compress_drop(self, info, verbose=False):
        var_old = info['var_old']
        if verbose:
            logging.info('RMSProp drop: {}'.format(info['var_old_name']))
        assert info['type'] == 'variable'
        for group in self.param_groups:
            index = index_tensor_in(var_old, group['params'], raise_error=False)
            if index is None:
                continue
            if check_tensor_in(var_old, self.state):
                self.state.pop(var_old)
            del group['params'][index]
            if self._mgr is not None:
                self._mgr.mark_dirty()
            return
        raise AssertionError('Var: {} not in RMSProp'.format(info['var_old_name']))


def _view(arena, off, shape, strides):
    if strides is None:
        n = 1
        for s in shape:
            n *= s
        return arena[off:off + n].view(shape)
    return torch.as_strided(arena, shape, strides, off)
