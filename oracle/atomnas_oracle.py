"""ORACLE -- test infrastructure only.  CPU restatement of the AtomNAS supernet-training hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product package
(atomnas_amd/) never does, and has no CPU fallback.

The reference (meijieru/AtomNAS) is pure Python over PyTorch; its arithmetic lives in ATen (requirements.txt:1,
`torch>=1.0`, unpinned).  This file restates the *algorithm* of the path functionally -- plain functions over a
state_dict, evaluated with torch CPU ops in fp32 (or fp64 when asked) -- citing the reference line each function follows.
It is pinned against fixtures generated from the reference itself in this container (tools/make_golden.py ->
tests/golden/*.pt, checked by tests/test_oracle_golden.py) and against the reference's own known-answer tests
(tests/utils/prune_test.py, optim_test.py, rmsprop_test.py, models/compress_utils_test.py: their vectors are restated in
tests/test_oracle_golden.py and tests/test_host_logic.py).
"""
import collections
import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ storage emulation
class _RoundFwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class NoQuant:
    """fp32 storage: nothing is rounded."""
    @staticmethod
    def f(x):
        return x

    @staticmethod
    def b(x):
        return x


class Bf16Storage:
    """Emulates where the HIP path rounds to bf16 when activations are stored in bf16 ('numerics'): f() rounds a
    tensor that is written to HBM in the forward pass, b() rounds the gradient that flows back through this point (the
    gradient tensors the backward pass writes to HBM / feeds to an MFMA).  With these hooks the oracle and the bf16 HIP
    path agree up to fp32 accumulation order, so bf16 parity is tested tightly instead of against bf16-vs-fp64 noise."""
    @staticmethod
    def f(x):
        return _RoundFwd.apply(x)

    @staticmethod
    def b(x):
        return _RoundBwd.apply(x)


class _DwMatrixCore(torch.autograd.Function):
    """The depthwise convolution as atomnas_amd/csrc/dwconv_mm.hip (stride 2 forward: dwconv_mm2.hip) computes it in bf16 storage mode (tap arithmetic on the matrix
    cores), restated: forward operands -- the activated input, clamped to the fp16 range, and the taps -- rounded to fp16, fp32
    accumulation; backward (where the backward kernel of that file runs): the gradient of the raw output and the taps rounded to bf16
    for the input gradient, the gradient and the activated input rounded to bf16 for the weight gradient.  The operation itself is
    the reference's nn.Conv2d(groups = channels) (models/mobilenet_base.py:330-336)."""

    @staticmethod
    def forward(ctx, x, w, stride, pad, groups, fwd_mm, bwd_mm):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, groups, bwd_mm)
        if fwd_mm:
            x = x.clamp(-65504.0, 65504.0).to(torch.float16).to(x.dtype)
            w = w.to(torch.float16).to(w.dtype)
        return F.conv2d(x, w, None, stride, pad, 1, groups)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad, groups, bwd_mm = ctx.cfg
        r = (lambda t: t.to(torch.bfloat16).to(t.dtype)) if bwd_mm else (lambda t: t)
        gy = r(gy)
        gx = torch.nn.grad.conv2d_input(x.shape, r(w), gy, stride, pad, 1, groups)
        gw = torch.nn.grad.conv2d_weight(r(x), w.shape, gy, stride, pad, 1, groups)
        return gx, gw, None, None, None, None, None


def bf16_storage_mm(pred):
    """Bf16Storage for a HIP path that runs depthwise convolutions on the matrix cores: pred(k, stride, N, C, H, W, slab) -> (forward
    kernel is csrc/dwconv_mm.hip's, backward kernel is) -- tests build it from the library's own query atomnas_dwconv_mm_supported."""
    class Bf16StorageMM(Bf16Storage):
        @staticmethod
        def dwconv(x, w, stride, k, slab):
            fwd_mm, bwd_mm = pred(k, stride, x.shape[0], x.shape[1], x.shape[2], x.shape[3], slab)
            if not (fwd_mm or bwd_mm):
                return F.conv2d(x, w, None, stride, (k - 1) // 2, 1, x.shape[1])
            return _DwMatrixCore.apply(x, w, stride, (k - 1) // 2, x.shape[1], bool(fwd_mm), bool(bwd_mm))
    return Bf16StorageMM


# ------------------------------------------------------------------------------------------------ structure
def spec_from_model(model):
    """Structure description read from public attributes only (works on the reference's modules and on atomnas_amd's):
    models/mobilenet_supernet.py:124-163 (stem, blocks, last conv, pool, classifier)."""
    feats = list(model.features.named_children())
    stem_name, stem = feats[0]
    last_name, last = feats[-2]
    blocks = []
    for name, b in feats[1:-2]:
        fused = hasattr(b, 'depth_ops')   # InvertedResidualChannelsFused (models/mobilenet_base.py:145-274)
        blocks.append(dict(name='features.' + name, inp=b.input_dim, oup=b.output_dim, stride=b.stride, expand=b.expand,
                           channels=list(b.channels), ks=list(b.kernel_sizes), res=b.use_res_connect, fused=fused,
                           se=bool(fused and getattr(b, 'se_ratio', None) is not None)))
    bn = list(stem.children())[1]
    drop = list(model.classifier.children())[0]
    return dict(stem='features.' + stem_name, last='features.' + last_name, blocks=blocks, eps=bn.eps, momentum=bn.momentum,
                dropout=drop.p, act=getattr(model, 'active_fn', 'nn.ReLU'), pool=model.input_size // 32,
                num_classes=model.num_classes)


def _act(x, name):
    if name == 'nn.ReLU':
        return F.relu(x)
    if name == 'nn.ReLU6':
        return F.relu6(x)
    if name == 'nn.Swish':
        return x * torch.sigmoid(x)
    raise ValueError(name)


def bn(x, sd, prefix, training, eps, momentum, stats_out=None):
    """nn.BatchNorm2d forward (models/mobilenet_base.py:142,342).  In training mode the new running statistics are written
    to stats_out[prefix] = (running_mean, running_var) instead of mutating sd."""
    w, b = sd[prefix + '.weight'], sd[prefix + '.bias']
    rm, rv = sd[prefix + '.running_mean'], sd[prefix + '.running_var']
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, 0.0, eps)
    rm2, rv2 = rm.clone(), rv.clone()
    if momentum is None:  # cumulative moving average, utils/common.py:225-226
        nbt = int(sd[prefix + '.num_batches_tracked']) + 1
        momentum = 1.0 / nbt
    y = F.batch_norm(x, rm2, rv2, w, b, True, momentum, eps)
    if stats_out is not None:
        stats_out[prefix] = (rm2, rv2)
    return y


def conv_bn_act(x, sd, prefix, stride, groups, k, training, spec, stats_out, q=NoQuant, dense=True, store_out=True, slab=False):
    """ConvBNReLU (models/mobilenet_base.py:120-142): conv(no bias, pad (k-1)/2) -> BN -> activation.
    q: storage emulation; dense convs run on MFMA with weights rounded to the storage type, depthwise taps stay fp32 unless the
    storage model says the depthwise kernel is a matrix-core one (q.dwconv, bf16_storage_mm).  slab: the input is a hidden tensor of
    an expanding block (slab-major in the HIP path: the layout the matrix-core depthwise kernels exist for)."""
    w = sd[prefix + '.0.weight']
    if not dense and hasattr(q, 'dwconv'):
        y = q.dwconv(x, w, stride, k, slab)
    else:
        y = F.conv2d(x, q.f(w) if dense else w, None, stride, (k - 1) // 2, 1, groups)
    y = q.f(y)                       # raw conv output is stored
    if dense:
        y = q.b(y)                   # its gradient is rounded before the weight / input gradient GEMMs
    y = bn(y, sd, prefix + '.1', training, spec['eps'], spec['momentum'], stats_out)
    y = _act(q.b(y), spec['act'])    # the masked gradient wrt the BN output is stored
    return q.f(y) if store_out else y


def se_forward(x, sd, prefix, act):
    """SqueezeAndExcitation.forward (models/mobilenet_base.py:109-112): mean over H, W -> 1x1 conv + bias -> activation -> 1x1 conv +
    bias -> sigmoid -> scale."""
    s = x.mean([2, 3], keepdim=True)
    s = F.conv2d(s, sd[prefix + '.se_reduce.weight'], sd[prefix + '.se_reduce.bias'])
    s = _act(s, act)
    s = F.conv2d(s, sd[prefix + '.se_expand.weight'], sd[prefix + '.se_expand.bias'])
    return torch.sigmoid(s) * x


def fused_block_forward(x, sd, blk, training, spec, stats_out=None, q=NoQuant):
    """InvertedResidualChannelsFused.forward (models/mobilenet_base.py:256-267) with _build's layout (:181-231): one expand
    ConvBNReLU over all hidden channels, Narrow + depthwise ConvBNReLU per kernel size, concatenation, optional SE, projection
    conv + BN, residual."""
    name = blk['name']
    t = x
    if blk['expand']:
        t = conv_bn_act(x, sd, name + '.expand_conv', 1, 1, 1, training, spec, stats_out, q, dense=True, store_out=False)
    outs, start = [], 0
    j = 1 if blk['expand'] else 0
    for i, (h, k) in enumerate(zip(blk['channels'], blk['ks'])):
        ti = t.narrow(1, start, h) if blk['expand'] else t
        start += h
        # without SE the activated depthwise output is the (rounded) MFMA operand of the projection; with SE the gated tensor is
        outs.append(conv_bn_act(ti, sd, '{}.depth_ops.{}.{}'.format(name, i, j), blk['stride'], h, k, training, spec, stats_out, q,
                                dense=False, store_out=not blk.get('se'), slab=bool(blk['expand'])))
    res = torch.cat(outs, 1) if len(outs) != 1 else outs[0]
    if blk.get('se'):
        res = q.b(q.f(se_forward(res, sd, name + '.se_op', spec['act'])))
    res = F.conv2d(res, q.f(sd[name + '.project_conv.0.weight']))
    res = q.b(q.f(res))
    res = bn(res, sd, name + '.project_conv.1', training, spec['eps'], spec['momentum'], stats_out)
    out = x + res if blk['res'] else res
    return q.b(q.f(out))


def block_forward(x, sd, blk, training, spec, stats_out=None, q=NoQuant):
    """InvertedResidualChannels.forward (models/mobilenet_base.py:371-382) with _build's layout (:305-346)."""
    if blk.get('fused'):
        return fused_block_forward(x, sd, blk, training, spec, stats_out, q)
    if len(blk['channels']) == 0:
        return x
    name = blk['name']
    outs = []
    for i, (h, k) in enumerate(zip(blk['channels'], blk['ks'])):
        p = '{}.ops.{}'.format(name, i)
        t = x
        j = 0
        if blk['expand']:
            # the activated expand output feeds the depthwise kernel in fp32 registers: not stored
            t = conv_bn_act(t, sd, p + '.0', 1, 1, 1, training, spec, stats_out, q, dense=True, store_out=False)
            j = 1
        # the activated depthwise output is the (rounded) MFMA operand of the projection
        t = conv_bn_act(t, sd, '{}.{}'.format(p, j), blk['stride'], h, k, training, spec, stats_out, q, dense=False, store_out=True,
                        slab=bool(blk['expand']))
        t = F.conv2d(t, q.f(sd['{}.{}.weight'.format(p, j + 1)]))
        outs.append(t)
    tmp = q.b(q.f(sum(outs)))
    tmp = bn(tmp, sd, name + '.pw_bn', training, spec['eps'], spec['momentum'], stats_out)
    out = x + tmp if blk['res'] else tmp
    return q.b(q.f(out))


def model_forward(x, sd, spec, training, stats_out=None, dropout_mask=None, return_features=False, q=NoQuant):
    """MobileNetV2.forward (models/mobilenet_supernet.py:169-173).  dropout_mask: optional [N, last_channel] keep mask
    (already scaled by 1/(1-p)) so that a run can be compared with the HIP path's own mask; None = no dropout."""
    y = conv_bn_act(q.f(x), sd, spec['stem'], 2, 1, 3, training, spec, stats_out, q)
    y = q.b(y)
    feats = [y]
    for blk in spec['blocks']:
        y = block_forward(y, sd, blk, training, spec, stats_out, q)
        feats.append(y)
    y = conv_bn_act(y, sd, spec['last'], 1, 1, 1, training, spec, stats_out, q, store_out=False)
    y = F.avg_pool2d(y, spec['pool']).squeeze(3).squeeze(2)
    if dropout_mask is not None:
        y = y * dropout_mask
    y = q.b(q.f(y))
    logits = F.linear(y, q.f(sd['classifier.1.weight']), sd['classifier.1.bias'])
    logits = q.b(logits)
    return (logits, feats) if return_features else logits


# ------------------------------------------------------------------------------------------------ losses / regularisers
def ce_label_smooth(logits, target, eps):
    """CrossEntropyLabelSmooth.forward, reduction 'none' (utils/optim.py:199-207)."""
    logp = F.log_softmax(logits, dim=1)
    k = logits.shape[1]
    t = torch.zeros_like(logp).scatter_(1, target.unsqueeze(1), 1)
    t = (1 - eps) * t + eps / k
    return torch.sum(-t * logp, 1)


def topk_errors(logits, target, ks=(1, 5)):
    """top-k error lists of forward_loss (common.py:73-79)."""
    _, pred = logits.topk(max(ks))
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return {k: (1.0 - correct[:k].float().sum(0)) for k in ks}


def l2_loss(named_params, weight_decay, method='mnas'):
    """cal_l2_loss (utils/optim.py:210-249).  named_params: iterable of (name, tensor)."""
    loss = 0.0
    for name, p in named_params:
        nd = p.dim()
        if method == 'mnas':
            wd = weight_decay if (nd in (2, 4) or (nd == 1 and 'classifier' in name)) else 0.0
        elif method == 'slimmable':
            wd = weight_decay if ((nd == 4 and p.shape[1] != 1) or nd == 2) else 0.0
        else:
            raise ValueError(method)
        loss = loss + wd * (p ** 2).sum()
    return loss * 0.5


def bn_l1_loss(bn_weights, penalties, rho):
    """cal_bn_l1_loss (utils/prune.py:161-167)."""
    loss = 0.0
    for w, pen in zip(bn_weights, penalties):
        loss = loss + rho * pen * w.abs().sum()
    return loss


# ------------------------------------------------------------------------------------------------ MACs, penalties, schedules
def conv_macs(cin, cout, k, groups, ho, wo, batch=1):
    """n_macs of a conv (utils/model_profiling.py:82-88)."""
    return (cin * cout * k * k * ho * wo // groups) * batch


def block_branch_macs(blk, h_in):
    """Per-branch n_macs (expand + depthwise + project), input resolution h_in (square); utils/model_profiling.py:121-127."""
    s = blk['stride']
    h_out = (h_in - 1) // s + 1
    res = []
    for h, k in zip(blk['channels'], blk['ks']):
        m = 0
        if blk['expand']:
            m += conv_macs(blk['inp'], h, 1, 1, h_in, h_in)
        m += conv_macs(h, h, k, h, h_out, h_out)
        m += conv_macs(h, blk['oup'], 1, 1, h_out, h_out)
        res.append(m)
    return res, h_out


def model_macs(spec, image_size, stem_out, last_in, last_out):
    """model.n_macs as stamped by model_profiling (utils/model_profiling.py:191-241) for batch 1."""
    h = (image_size - 1) // 2 + 1
    total = conv_macs(3, stem_out, 3, 1, h, h)
    per_block = []
    for blk in spec['blocks']:
        br, h2 = block_branch_macs(blk, h)
        per_block.append(br)
        total += sum(br)
        h = h2
    total += conv_macs(last_in, last_out, 1, 1, h, h)
    total += last_out * h * h            # AvgPool2d: ins[1]*ins[2]*ins[3]*ins[0]  (:106-111)
    total += last_out * spec['num_classes']  # Linear (:101-105)
    return total, per_block


def prune_penalties(spec, image_size, bn_prune_filter='expansion_only_skip_expand1'):
    """get_bn_to_prune (utils/prune.py:89-158): names, penalties, per-channel MACs of the prunable depthwise-BN gammas."""
    h = (image_size - 1) // 2 + 1
    names, pairs = [], []
    for blk in spec['blocks']:
        br, h2 = block_branch_macs(blk, h)
        h = h2
        if bn_prune_filter.endswith('skip_expand1') and not blk['expand']:
            continue
        pos = 1 if blk['expand'] else 0
        for i, (hid, macs) in enumerate(zip(blk['channels'], br)):
            names.append('{}.ops.{}.{}.1.weight'.format(blk['name'], i, pos))
            pairs.append((hid, macs / hid))
    pcf = [v for _, v in pairs]
    if bn_prune_filter.startswith('expansion_only'):
        numel_total = sum(n for n, _ in pairs)
        normalizer = sum(n * v for n, v in pairs) / (numel_total + 1e-5)
        pen = [v / normalizer for _, v in pairs]
    elif bn_prune_filter.startswith('equal_penalty'):
        pen = [1 for _ in pairs]
    else:
        raise NotImplementedError(bn_prune_filter)
    return names, pen, pcf


def rho_schedule(i, rho, epoch_free, epoch_warmup, steps_per_epoch, stepwise=True):
    """get_rho_scheduler.linear_fun (utils/prune.py:231-240)."""
    free, warm = epoch_free * steps_per_epoch, epoch_warmup * steps_per_epoch
    if not stepwise:
        i = (i // steps_per_epoch) * steps_per_epoch
    if i < free:
        return 0.0
    if i >= warm:
        return rho
    return (i - free) / (warm - free) * rho


def lr_lambda(i, lr, base_lr, steps_per_epoch, scheduler='exp_decaying', gamma=0.97, epoch_interval=2.4, stepwise=False,
              epoch_warmup=5, num_epochs=350):
    """Multiplier of get_lr_scheduler's LambdaLR (utils/optim.py:252-306): linear warm-up from base_lr/lr to 1 over
    epoch_warmup epochs when lr > base_lr, then gamma**(i/interval) (staircase when lr_stepwise is False)."""
    warm = epoch_warmup * steps_per_epoch
    if lr > base_lr and i <= warm:
        r = base_lr / lr
        return r + i / warm * (1 - r)
    if scheduler.startswith('exp_decaying'):
        interval = steps_per_epoch * epoch_interval
        j = i
        if not stepwise:
            j = (j // interval) * interval
        res = gamma ** (j / interval)
        floor = 0.05 if 'trunc' in scheduler else 0.0
        return res if res > floor else floor
    if scheduler == 'linear_decaying':
        return 1 - i / (num_epochs * steps_per_epoch)
    raise NotImplementedError(scheduler)


def ema_decay(momentum, num_updates):
    """ExponentialMovingAverage.forward's momentum (utils/optim.py:57-61)."""
    if num_updates is None:
        return momentum
    return min(momentum, (1.0 + num_updates) / (10.0 + num_updates))


def ema_adjust_momentum(momentum, steps_multi):
    """ExponentialMovingAverage.adjust_momentum (utils/optim.py:167-177)."""
    return momentum ** (1.0 / steps_multi)


# ------------------------------------------------------------------------------------------------ optimizer / EMA arithmetic
def rmsprop_update(p, g, sq, buf, lr, alpha, eps, momentum, eps_inside_sqrt=True):
    """One RMSprop.step on a tensor, in the reference's operation order (utils/rmsprop.py:106-130, weight_decay 0,
    not centered).  Mutates p, sq, buf in place; returns nothing."""
    sq.mul_(alpha).addcmul_(g, g, value=1 - alpha)
    avg = sq.add(eps).sqrt_() if eps_inside_sqrt else sq.sqrt().add_(eps)
    if momentum > 0:
        buf.mul_(momentum).addcdiv_(g, avg)
        p.add_(buf, alpha=-lr)
    else:
        p.addcdiv_(g, avg, value=-lr)


def ema_update(shadow, x, decay):
    """shadow.mul_(d).add_(1-d, x) (utils/optim.py:64-65)."""
    shadow.mul_(decay).add_(x, alpha=1.0 - decay)


# ------------------------------------------------------------------------------------------------ masks and shrink
def alive_mask(gamma, threshold):
    """weight.abs() > threshold (utils/prune.py:190-195, train.py:46-63)."""
    return gamma.detach().abs() > threshold


def mask_along_dim(src, mask, dim):
    """_mask_along_dim (models/compress_utils.py:31-37): kept slices of src along dim 0 or 1."""
    if dim == 0:
        return src[mask]
    if dim == 1:
        return src[:, mask]
    raise NotImplementedError()


def pruned_flops(masks, per_channel_flops):
    """cal_pruned_flops (utils/prune.py:198-212): sum over tensors of n_pruned * per-channel MACs (python floats)."""
    total = 0
    for m, pcf in zip(masks, per_channel_flops):
        total += int((~m).sum().item()) * pcf
    return total


def shrink_state_dict(sd, spec, masks_by_block):
    """Whole-model restatement of shrink_model (train.py:27-81) + copmress_inverted_residual_channels
    (models/compress_utils.py:180-301) on a state_dict: returns (new_sd, new_spec).  masks_by_block: block name -> list of
    bool masks (one per branch).  Branches with no survivor are dropped and the remaining ones renumbered."""
    new_sd = collections.OrderedDict()
    new_blocks = []
    done = set()
    for blk in spec['blocks']:
        name = blk['name']
        masks = masks_by_block.get(name)
        if masks is None:
            new_blocks.append(dict(blk))
            continue
        nb = dict(blk)
        nb['channels'], nb['ks'] = [], []
        new_i = 0
        for i, mask in enumerate(masks):
            old = '{}.ops.{}'.format(name, i)
            for key in [k for k in sd if k.startswith(old + '.')]:
                done.add(key)
            n_keep = int(mask.sum().item())
            if n_keep == 0:
                continue
            new = '{}.ops.{}'.format(name, new_i)
            j = 0
            if blk['expand']:
                new_sd[new + '.0.0.weight'] = mask_along_dim(sd[old + '.0.0.weight'], mask, 0).clone()
                for a in ('weight', 'bias', 'running_mean', 'running_var'):
                    new_sd[new + '.0.1.' + a] = mask_along_dim(sd[old + '.0.1.' + a], mask, 0).clone()
                new_sd[new + '.0.1.num_batches_tracked'] = sd[old + '.0.1.num_batches_tracked'].clone()
                j = 1
            new_sd['{}.{}.0.weight'.format(new, j)] = mask_along_dim(sd['{}.{}.0.weight'.format(old, j)], mask, 0).clone()
            for a in ('weight', 'bias', 'running_mean', 'running_var'):
                new_sd['{}.{}.1.{}'.format(new, j, a)] = mask_along_dim(sd['{}.{}.1.{}'.format(old, j, a)], mask, 0).clone()
            new_sd['{}.{}.1.num_batches_tracked'.format(new, j)] = sd['{}.{}.1.num_batches_tracked'.format(old, j)].clone()
            new_sd['{}.{}.weight'.format(new, j + 1)] = mask_along_dim(sd['{}.{}.weight'.format(old, j + 1)], mask, 1).clone()
            nb['channels'].append(n_keep)
            nb['ks'].append(blk['ks'][i])
            new_i += 1
        new_blocks.append(nb)
    out = collections.OrderedDict()
    for k, v in sd.items():  # keep the reference's key order for everything that is not a rebuilt branch
        if k not in done:
            out[k] = v.clone()
    out.update(new_sd)
    new_spec = dict(spec)
    new_spec['blocks'] = new_blocks
    return out, new_spec


# ------------------------------------------------------------------------------------------------ one training iteration
def train_step(sd, spec, opt_state, ema_shadow, x, target, hp, prune_names=None, penalties=None, dropout_mask=None):
    """One iteration of run_one_epoch (train.py:165-236) on a state_dict, single process:
    forward -> CE-smooth mean + L2 + L1 -> backward -> RMSprop -> EMA.  Mutates sd / opt_state / ema_shadow in place.
    hp: dict(lr, rho, weight_decay, wd_method, label_smoothing, alpha, eps, momentum, ema_decay).  Returns a dict of
    diagnostics (loss terms, logits, grads)."""
    params = collections.OrderedDict()
    work = {}
    for k, v in sd.items():
        if v.is_floating_point() and not ('running_' in k):
            params[k] = v.detach().clone().requires_grad_(True)
            work[k] = params[k]
        else:
            work[k] = v
    stats = {}
    logits = model_forward(x, work, spec, True, stats, dropout_mask)
    loss_vec = ce_label_smooth(logits, target, hp['label_smoothing'])
    loss = loss_vec.mean()
    loss_l2 = l2_loss(params.items(), hp['weight_decay'], hp.get('wd_method', 'mnas'))
    gam = [params[n] for n in (prune_names or [])]
    loss_l1 = bn_l1_loss(gam, penalties or [], hp['rho'])
    total = loss + loss_l2 + loss_l1
    total.backward()
    grads = {k: p.grad.detach().clone() for k, p in params.items()}
    params_before = {k: p.detach().clone() for k, p in params.items()}
    with torch.no_grad():
        for k, p in params.items():
            st = opt_state.setdefault(k, {})
            if 'square_avg' not in st:
                st['square_avg'] = torch.zeros_like(p)
                st['momentum_buffer'] = torch.zeros_like(p)
            newp = sd[k]
            rmsprop_update(newp, p.grad, st['square_avg'], st['momentum_buffer'], hp['lr'], hp['alpha'], hp['eps'], hp['momentum'])
        for prefix, (rm, rv) in stats.items():
            sd[prefix + '.running_mean'].copy_(rm)
            sd[prefix + '.running_var'].copy_(rv)
            sd[prefix + '.num_batches_tracked'] += 1
        if ema_shadow is not None:
            d = hp['ema_decay']
            for k in ema_shadow:
                ema_update(ema_shadow[k], sd[k], d)
    return dict(loss=float(loss.detach()), loss_l2=float(torch.as_tensor(loss_l2).detach()), loss_l1=float(torch.as_tensor(loss_l1).detach()), logits=logits.detach(), grads=grads,
                loss_vec=loss_vec.detach(), params_before=params_before)
