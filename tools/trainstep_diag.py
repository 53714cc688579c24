"""Diagnostic (GPU): where does the HIP training step leave the CPU oracle, and is it bit-reproducible?

For each (input size, batch) and eager / graph mode: two iterations of engine.TrainStep against oracle.train_step on the same
inputs; prints per-tensor gradient errors after iteration 1 and 2 and optimizer-state errors, then repeats the whole run and
compares every arena bit for bit.  Used to decide 'bug or chaos' for tests/test_train_step_gpu.py."""
import collections
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atomnas_oracle as orc  # noqa: E402
from test_block_gpu import TINY, _randomize  # noqa: E402


def setup(dtype, size):
    from atomnas_amd import engine
    from atomnas_amd.models import mobilenet_supernet as ms
    from atomnas_amd.utils import model_profiling as mp
    from atomnas_amd.utils import optim as aopt
    from atomnas_amd.utils import prune as aprune
    from atomnas_amd.utils import rmsprop
    kw = dict(TINY, input_size=size)
    model = ms.Model(**kw)
    model.set_compute_dtype(dtype)
    _randomize(model, 21)
    mp.model_profiling(model, size, size, verbose=False)
    sd = collections.OrderedDict((k, v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items())
    spec = orc.spec_from_model(model)
    model.cuda().train()
    pinfo = aprune.get_bn_to_prune(model, {'bn_prune_filter': 'expansion_only_skip_expand1'}, verbose=False)
    opt = rmsprop.RMSprop(model.parameters(), lr=0.01, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
    ema = aopt.ExponentialMovingAverage(0.99)
    for n, p in model.named_parameters():
        ema.register(n, p)
    for n, b in model.named_buffers():
        if 'running' in n:
            ema.register(n, b)
    return model, sd, spec, pinfo, opt, ema, engine


def digest(mgr, ts):
    h = hashlib.sha1()
    for a in (mgr.P, mgr.G, mgr.SQ, mgr.BUF, mgr.EMA, mgr.S, mgr.SEMA, ts.loss):
        if a is not None:
            h.update(a.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


def run(size, N, use_graph, dtype, compare):
    model, sd, spec, pinfo, opt, ema, engine = setup(dtype, size)
    ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-3, label_smoothing=0.1, batch_size=N, image_size=size, use_graph=use_graph)
    names, pen, _ = orc.prune_penalties(spec, size)
    opt_state = {}
    ema_o = collections.OrderedDict((k, v.clone()) for k, v in sd.items() if v.is_floating_point())
    g = torch.Generator().manual_seed(5)
    digs = []
    for step in range(2):
        x = torch.randn(N, 3, size, size, generator=g)
        y = torch.randint(0, 10, (N,), generator=g)
        lr, rho = 0.002 * (1 + step), 1e-3 * (1 + step)
        d = ema.momentum_at(step + 1)
        ts.set_batch(x.cuda(), y.cuda())
        ts.step(lr=lr, rho=rho)
        torch.cuda.synchronize()
        digs.append(digest(ts.mgr, ts))
        if not compare:
            continue
        ref = orc.train_step(sd, spec, opt_state, ema_o, x.double(), y, dict(lr=lr, rho=rho, weight_decay=1e-3, wd_method='mnas',
                             label_smoothing=0.1, alpha=0.9, eps=1e-3, momentum=0.9, ema_decay=d), names, pen)
        print("  step %d loss got %s ref %.6f %.6f %.6f" % (step, ["%.6f" % v for v in ts.loss.tolist()], ref['loss'], ref['loss_l2'], ref['loss_l1']))
        rows = []
        num = den = 0.0
        for n, p in model.named_parameters():
            gg, rr = p.grad.double().cpu(), ref['grads'][n]
            if rr.dim() in (2, 4) or (rr.dim() == 1 and 'classifier' in n):   # the L2 gradient is applied inside the optimizer kernel
                rr = rr - 1e-3 * ref['params_before'][n]
            e = (gg - rr)
            s = max(1e-12, float(rr.abs().max()))
            bad = float(((e.abs() > 1e-3 * s + 2e-3 * rr.abs()).double().mean()))
            rl = float(e.norm() / max(float(rr.norm()), 1e-30))
            num += float((e * e).sum()); den += float((rr * rr).sum())
            rows.append((rl, bad, n, rr.numel(), float(e.abs().max()), s))
        rows.sort(reverse=True)
        print("    grads: total rel-L2 %.3e; worst tensors (rel-L2, frac>tol, name, numel, maxerr, refmax):" % ((num / den) ** 0.5))
        for r in rows[:6]:
            print("      %.3e %.3f %-44s %6d %.3e %.3e" % r)
        nbad = sum(1 for r in rows if r[0] > 1e-2)
        print("    tensors with rel-L2 > 1e-2: %d / %d" % (nbad, len(rows)))
    if compare:
        for key in ("square_avg", "momentum_buffer"):
            rows = []
            for n, p in model.named_parameters():
                got, rr = opt.state[p][key].double().cpu(), opt_state[n][key]
                e = got - rr
                s = max(1e-12, float(rr.abs().max()))
                rows.append((float(e.norm() / max(float(rr.norm()), 1e-30)), float((e.abs() > 1e-2 * s + 2e-2 * rr.abs()).double().mean()), n, float(e.abs().max()), s))
            rows.sort(reverse=True)
            print("    %s worst (rel-L2, frac>tol, name, maxerr, refmax):" % key)
            for r in rows[:5]:
                print("      %.3e %.3f %-44s %.3e %.3e" % r)
    return digs


if __name__ == "__main__":
    for dtype in (torch.float32, torch.bfloat16):
        for size, N in ((64, 6), (128, 8)):
            for use_graph in (False, True):
                print("== dtype %s size %d N %d graph %s" % (dtype, size, N, use_graph))
                d1 = run(size, N, use_graph, dtype, compare=(dtype == torch.float32))
                d2 = run(size, N, use_graph, dtype, compare=False)
                print("  digests run1 %s run2 %s -> %s" % (d1, d2, "BIT-IDENTICAL" if d1 == d2 else "DIFFERENT"))
