"""Small helpers shared by the host-side mirrors (counterparts of the hot-path-adjacent parts of utils/common.py)."""
import logging
import numbers

import torch


def add_prefix(name, prefix=None, split='.'):
    """`prefix.name`, or `name` when there is no prefix (utils/common.py:163-168)."""
    return name if prefix is None else '{}{}{}'.format(prefix, split, name)


def get_params_by_name(model, names):
    """Parameters / buffers of `model` in the order of `names` (utils/common.py:13-19)."""
    table = dict(model.named_parameters())
    table.update(dict(model.named_buffers()))
    return [table[n] for n in names]


def index_tensor_in(tensor, seq, raise_error=True):
    """Position of `tensor` in `seq` by identity (utils/common.py:171-186)."""
    for pos, item in enumerate(seq):
        if item is tensor:
            return pos
    if raise_error:
        raise ValueError('Tensor not in list')
    return None


def check_tensor_in(tensor, container):
    """Identity membership test for lists and dict keys (utils/common.py:189-203)."""
    if isinstance(container, dict):
        container = container.keys()
    elif not isinstance(container, list):
        raise ValueError('Unknown iterable: {}'.format(type(container)))
    return any(item is tensor for item in container)


def get_device(x):
    """Device of a tensor or of a module's first parameter (utils/common.py:140-150)."""
    if isinstance(x, torch.Tensor):
        return x.device
    if isinstance(x, torch.nn.Module):
        return next(x.parameters()).device
    raise RuntimeError('{} do not have `device`'.format(type(x)))


def extract_item(x):
    if isinstance(x, numbers.Number):
        return x
    if isinstance(x, torch.Tensor):
        return x.item()
    raise ValueError('Unknown type: {}'.format(type(x)))


def unwrap_state_dict(checkpoint, verbose=True):
    """Strips a leading `module.` from every key (checkpoints saved through the DDP wrapper)."""
    out = {}
    stripped = False
    for key, val in checkpoint.items():
        if key.startswith('module.'):
            key = key[len('module.'):]
            stripped = True
        out[key] = val
    if stripped and verbose:
        logging.info('Unwrap state_dict')
    return out


def set_random_seed(seed):
    import random

    import numpy as np
    logging.info('Set seed: {}'.format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def bn_calibration(m, cumulative_bn_stats=True):
    """`model.apply(bn_calibration)`: reset running statistics, batch-statistics mode, cumulative average
    (utils/common.py:214-226)."""
    if isinstance(m, torch.nn.BatchNorm2d):
        m.reset_running_stats()
        m.train()
        if cumulative_bn_stats:
            m.momentum = None
