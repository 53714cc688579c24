"""BASELINE.json configurations and the remaining SURVEY section 8 rows on the GPU ("configs not exercised"):

  cfg 2  AtomNAS-A supernet (apps/slimming/shrink/atomnas_a.yml), full size, forward/backward vs the oracle + bf16 training steps
  cfg 3  AtomNAS-A after a forced 30 % of dead atoms: masks bit-exact, shrink, the ragged network against the oracle
  cfg 4  full-size AtomNAS-C supernet eval logits against the fixture generated from the reference (tests/golden)
  a5     BatchNorm cumulative-average calibration mode (utils/common.py:214-226) against the oracle
  a12    top-1 / top-5 counters of the loss kernel against common.py:73-79
"""
import collections
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import atomnas_oracle as orc  # noqa: E402

from kutil import assert_close, counter_fill, randomize_counter  # noqa: E402
from test_block_gpu import TINY, _randomize, _sd64  # noqa: E402

pytestmark = pytest.mark.gpu


def _supernet(name, dtype, size=224):
    from atomnas_amd import configs
    from atomnas_amd.models import mobilenet_supernet as ms
    model = ms.Model(**dict(configs.model_kwparams(name), input_size=size))
    model.set_compute_dtype(dtype)
    return model


def test_cfg4_full_size_c_supernet_eval_logits_match_reference_fixture(gpu_lib):
    g = torch.load(os.path.join(ROOT, "tests", "golden", "full_supernet_eval.pt"), weights_only=False)
    model = _supernet("atomnas_c_supernet", torch.float32)
    randomize_counter(model, 1)
    x = (counter_fill(torch.empty(2, 3, 224, 224), 1234) * 4).float()
    model.cuda().eval()
    with torch.no_grad():
        logits = model(x.cuda())
    assert_close("logits", logits, g["logits"], rtol=2e-3, atol=2e-4 * max(1.0, float(g["logits"].abs().max())))


def test_cfg2_atomnas_a_supernet_forward_backward_and_bf16_steps(gpu_lib):
    """Full-size AtomNAS-A supernet (input_channel 16): fp32 storage forward/backward against the float64 oracle, then three
    bf16 training iterations through the captured graphs (finite, decreasing loss on a memorisable batch)."""
    from atomnas_amd import configs, engine
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.utils import model_profiling as mp, optim as aopt, prune as aprune, rmsprop
    torch.manual_seed(3)
    model = _supernet("atomnas_a_supernet", torch.float32)
    model.apply(mb.init_weights_mnas)
    sd0 = _sd64(model)
    spec = orc.spec_from_model(model)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (2,), generator=g)
    model.cuda().train()
    drop = list(model.classifier.children())[0]
    drop.p = 0.0
    logits = model(x.cuda())
    loss = aopt.CrossEntropyLabelSmooth(1000, 0.1, reduction="none")(logits, y.cuda()).mean()
    loss.backward()
    torch.cuda.synchronize()
    work = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v) for k, v in sd0.items()}
    ref = orc.model_forward(x.double(), work, spec, True, {})
    rl = orc.ce_label_smooth(ref, y, 0.1).mean()
    rl.backward()
    assert_close("logits", logits, ref, rtol=5e-3, atol=5e-3 * max(1.0, float(ref.abs().max())))
    assert abs(float(loss.detach()) - float(rl.detach())) < 1e-3
    num = den = 0.0
    for name, p in model.named_parameters():
        d = p.grad.double().cpu() - work[name].grad
        num += float((d * d).sum()); den += float((work[name].grad ** 2).sum())
    assert (num / den) ** 0.5 < 3e-2, (num / den) ** 0.5

    # bf16 storage, the dtype of BASELINE config 2, a few captured iterations at a small batch
    model2 = _supernet("atomnas_a_supernet", torch.bfloat16)
    model2.apply(mb.init_weights_mnas)
    mp.model_profiling(model2, 224, 224, verbose=False)
    assert 1.505e9 < model2.n_macs < 1.515e9   # 1.511 GMAC
    model2.cuda().train()
    pinfo = aprune.get_bn_to_prune(model2, {"bn_prune_filter": "expansion_only_skip_expand1"}, verbose=False)
    assert len(pinfo.weight) == 63
    hp = configs.SEARCH_HPARAMS
    opt = rmsprop.RMSprop(model2.parameters(), lr=0.016, alpha=hp["alpha"], momentum=hp["momentum"], eps=hp["epsilon"], eps_inside_sqrt=True)
    ema = aopt.ExponentialMovingAverage(0.999)
    for n, p in model2.named_parameters():
        ema.register(n, p)
    for n, b in model2.named_buffers():
        if "running" in n:
            ema.register(n, b)
    ts = engine.TrainStep(model2, opt, ema, pinfo, weight_decay=1e-5, batch_size=8, image_size=224, use_graph=True)
    xb = torch.randn(8, 3, 224, 224, generator=g)
    yb = torch.randint(0, 1000, (8,), generator=g)
    ts.set_batch(xb.cuda(), yb.cuda())
    losses = []
    for _ in range(6):
        ts.step(lr=0.004, rho=1e-5)
        losses.append(ts.loss[0].item())
    assert all(v == v for v in losses) and losses[-1] < losses[0], losses


def test_cfg3_atomnas_a_forced_dead_atoms_shrink(gpu_lib):
    """SURVEY section 8d recipe for config 3: a seeded 30 % of the atoms get gamma = gamma_EMA = 0, then shrink_model.  Masks,
    kept counts and indices are bit-exact with the oracle; every surviving tensor equals the oracle's gather; the ragged network
    computes the oracle's function and keeps training."""
    sys.path.insert(0, ROOT)
    import train as T
    from atomnas_amd import engine, runtime
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.utils import config, model_profiling as mp, optim as aopt, prune as aprune, rmsprop
    torch.manual_seed(7)
    model = _supernet("atomnas_a_supernet", torch.float32)
    model.apply(mb.init_weights_mnas)
    _randomize(model, 9)
    mp.model_profiling(model, 224, 224, verbose=False)
    model.cuda().train()
    pinfo = aprune.get_bn_to_prune(model, {"bn_prune_filter": "expansion_only_skip_expand1"}, verbose=False)
    opt = rmsprop.RMSprop(model.parameters(), lr=0.002, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
    ema = aopt.ExponentialMovingAverage(0.99)
    for n, p in model.named_parameters():
        ema.register(n, p)
    for n, b in model.named_buffers():
        if "running" in n:
            ema.register(n, b)
    mgr = runtime.manager_of(model)
    mgr.attach_optimizer(opt)
    opt._mgr = mgr
    ema.attach(mgr)
    mgr.ensure()
    # kill a seeded 30 % of the atoms (a whole middle branch of one block and a whole block among them)
    g = torch.Generator().manual_seed(11)
    table = dict(model.named_parameters())
    with torch.no_grad():
        for i, name in enumerate(pinfo.weight):
            w = table[name]
            dead = torch.rand(w.numel(), generator=g) < 0.3
            if name.startswith("features.3.ops.1."):
                dead[:] = True
            if name.startswith("features.5."):
                dead[:] = True
            w[dead.cuda()] = 0.0
            ema.average(name)[dead.cuda()] = 0.0
    sd_pre = collections.OrderedDict((k, v.detach().cpu().clone()) for k, v in model.state_dict().items())
    spec = orc.spec_from_model(model)
    masks_ref = collections.OrderedDict()
    for blk in spec["blocks"]:
        if blk["expand"]:
            masks_ref[blk["name"]] = [orc.alive_mask(sd_pre["{}.ops.{}.1.1.weight".format(blk["name"], i)], 1e-3) for i in range(len(blk["ks"]))]
    # device masks: one launch, bit-exact incl. indices and counts
    masks, index, kept = aprune.alive_masks([table[n] for n in pinfo.weight], 1e-3, mode=1, with_index=True)
    flat_ref = [m for ms_ in masks_ref.values() for m in ms_]
    assert len(masks) == len(flat_ref) == 63
    for m, idx, k, r in zip(masks, index, kept.tolist(), flat_ref):
        assert torch.equal(m.cpu(), r)
        assert k == int(r.sum())
        assert torch.equal(idx[:k].cpu().long(), torch.nonzero(r).flatten())

    class F(dict):
        __getattr__ = dict.__getitem__
    config.FLAGS.bind(F(image_size=224, use_distributed=False))
    wrapper = torch.nn.Module()
    wrapper.module = model
    T.shrink_model(wrapper, ema, opt, pinfo, 1e-3, ema_only=False)
    sd_ref, spec_ref = orc.shrink_state_dict(sd_pre, spec, masks_ref)
    sd_post = model.state_dict()
    assert set(sd_post.keys()) == set(sd_ref.keys())
    for k, v in sd_ref.items():
        assert torch.equal(sd_post[k].cpu(), v), k     # a gather moves values, it does not compute: exact
    new_spec = orc.spec_from_model(model)
    assert [(b["channels"], b["ks"]) for b in new_spec["blocks"]] == [(b["channels"], b["ks"]) for b in spec_ref["blocks"]]
    assert any(len(b["channels"]) == 0 for b in new_spec["blocks"]) and any(len(b["channels"]) == 2 for b in new_spec["blocks"])
    # function of the ragged network (eval mode) against the oracle on the shrunk state_dict
    x = torch.randn(2, 3, 224, 224, generator=g)
    model.eval()
    with torch.no_grad():
        logits = model(x.cuda())
        ref = orc.model_forward(x.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd_ref.items()}, spec_ref, False)
    assert_close("logits after shrink", logits, ref, rtol=5e-3, atol=5e-3 * max(1.0, float(ref.abs().max())))
    # and it trains (bf16 storage, graph) on the ragged widths
    model.set_compute_dtype(torch.bfloat16)
    model.train()
    ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-5, batch_size=4, image_size=224, use_graph=True)
    ts.set_batch(torch.randn(4, 3, 224, 224, generator=g).cuda(), torch.randint(0, 1000, (4,), generator=g).cuda())
    l = []
    for _ in range(5):
        ts.step(lr=0.002, rho=1e-4)
        l.append(ts.loss[0].item())
    assert all(v == v for v in l) and l[-1] < l[0], l


def test_bn_calibration_cumulative_mode_matches_oracle(gpu_lib):
    """model.apply(bn_calibration) (utils/common.py:214-226: reset statistics, momentum None) followed by three forward passes:
    running statistics are the cumulative average of the batch statistics (unbiased variance), counters advance, for every
    BatchNorm of the network -- against the oracle's bn(momentum=None)."""
    from atomnas_amd.models import mobilenet_supernet as ms
    from atomnas_amd.utils.common import bn_calibration
    model = ms.Model(**TINY)
    model.set_compute_dtype(torch.float32)
    _randomize(model, 31)
    model.eval()
    model.apply(bn_calibration)
    sd = _sd64(model)
    spec = dict(orc.spec_from_model(model), momentum=None)
    assert spec["momentum"] is None
    model.cuda()
    g = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for it in range(3):
            x = torch.randn(8, 3, 64, 64, generator=g)
            model(x.cuda())
            stats = {}
            orc.model_forward(x.double(), sd, spec, True, stats)
            for prefix, (rm, rv) in stats.items():
                sd[prefix + ".running_mean"] = rm
                sd[prefix + ".running_var"] = rv
                sd[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    torch.cuda.synchronize()
    msd = model.state_dict()
    n = 0
    for k, v in sd.items():
        if "running" in k:
            s = max(1e-3, float(v.abs().max()))
            assert_close(k, msd[k], v, rtol=2e-3, atol=2e-4 * s)
            n += 1
        elif "num_batches" in k:
            assert int(msd[k]) == int(v) == 3, k
    assert n == 2 * sum(1 for m in model.modules() if isinstance(m, torch.nn.BatchNorm2d))
    # eval with the calibrated statistics
    model.eval()
    x = torch.randn(4, 3, 64, 64, generator=g)
    with torch.no_grad():
        assert_close("eval logits", model(x.cuda()), orc.model_forward(x.double(), sd, spec, False), rtol=2e-3, atol=2e-3)


def test_topk_counters_match_forward_loss(gpu_lib):
    """top-1 / top-5 hit counts of the loss kernel (no host sync) against common.py:73-79 (topk + eq), plus the loss values."""
    from atomnas_amd.utils import optim as aopt
    g = torch.Generator().manual_seed(12)
    for B, K in ((256, 1000), (37, 10), (5, 6)):
        logits = torch.randn(B, K, generator=g) * 3
        y = torch.randint(0, K, (B,), generator=g)
        logits[torch.arange(0, B, 3), y[::3]] += 6.0     # make a third of the samples (mostly) right
        crit = aopt.CrossEntropyLabelSmooth(K, 0.1, reduction="none")
        loss = crit(logits.cuda(), y.cuda())
        errs = orc.topk_errors(logits, y, (1, 5))
        want = [int(B - errs[1].sum()), int(B - errs[5].sum())]
        assert crit.topk_correct.tolist() == want, (crit.topk_correct.tolist(), want)
        assert_close("loss", loss, orc.ce_label_smooth(logits.double(), y, 0.1), rtol=1e-5, atol=1e-5)
        crit(logits.cuda(), y.cuda())                    # the counters accumulate until the meter zeroes them
        assert crit.topk_correct.tolist() == [2 * want[0], 2 * want[1]]


def test_reference_written_checkpoint_resumes_identically(gpu_lib):
    """Checkpoint interchange: a checkpoint WRITTEN BY THE REFERENCE (utils/common.py:123-137 -- model
    state_dict, torch's index-ordered RMSprop state, EMA {info, shadow, param}) is loaded through train.load_checkpoint into freshly
    built objects, and one more training iteration lands where the reference's own continuation lands (tests/golden/checkpoint_ref.pt,
    generated by tools/make_golden.py checkpoint)."""
    sys.path.insert(0, ROOT)
    import train as T
    from atomnas_amd import engine
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import mobilenet_supernet as ms
    from atomnas_amd.utils import model_profiling as mp, optim as aopt, prune as aprune, rmsprop
    from kutil import check_digest
    g = torch.load(os.path.join(ROOT, "tests", "golden", "checkpoint_ref.pt"), weights_only=False)
    torch.manual_seed(99)
    model = ms.Model(**g["kw"])
    model.apply(mb.init_weights_mnas)
    model.set_compute_dtype(torch.float32)
    mp.model_profiling(model, 64, 64, verbose=False)
    model.cuda().train()
    wrapper = torch.nn.Module()
    wrapper.module = model
    pinfo = aprune.get_bn_to_prune(model, {"bn_prune_filter": "expansion_only_skip_expand1"}, verbose=False)
    opt = rmsprop.RMSprop(wrapper.parameters(), lr=0.002, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
    ema = aopt.ExponentialMovingAverage(0.99)
    for n, p in model.named_parameters():
        ema.register(n, p)
    for n, b in model.named_buffers():
        if "running" in n:
            ema.register(n, b)
    last_epoch, best_val = T.load_checkpoint(g["checkpoint"], wrapper, opt, ema)
    assert last_epoch == 0 and best_val == 0.75
    assert mb.output_network(model) == g["kwparams"]
    ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-3, label_smoothing=0.1, batch_size=6, image_size=64, use_graph=False)
    ts.global_step = 2
    step = 2
    # the batch of the resumed iteration is the one make_golden.py chose for its ReLU margin (smallest |pre-activation| / rms 1.75e-6
    # in float64; round 4's batch had 7.8e-9, i.e. a pre-activation ON zero: whichever way fp32 rounding put it decided 1e-3 of every
    # upstream gradient, and the fp32 kernels were frozen on one slab rule to reproduce the reference's side of that coin)
    x = (counter_fill(torch.empty(6, 3, 64, 64), g["resume_seed"]) * 4).float()
    y = (torch.arange(6) * 3 + step) % 10
    ts.set_batch(x.cuda(), y.cuda())
    ts.step(lr=0.002 * (1 + step), rho=1e-3 * (1 + step))
    torch.cuda.synchronize()
    assert abs(ts.loss[0].item() - g["losses"][2]) < 2e-5 * max(1.0, g["losses"][2])
    # Tolerances (tools/ckpt_diag.py, GPU): against the fp64 oracle the HIP fp32 step and the reference's own fp32 arithmetic are
    # equally far off (median 2e-6 of the update, up to 3e-3 where the update is a few ulps of the parameter: gamma ~ 1 moving by
    # 2e-5, BN biases in front of another BN moving by rounding noise).  So: element-wise on the first 512 elements of every tensor,
    # relative to the size of the reference's UPDATE plus a few ulps of the value; digests of the whole tensors with a loose bound.
    REL = 2e-3
    sd = model.state_dict()
    head = g["after_head"]
    ck = g["checkpoint"]

    def close(name, got, want, old, rel, ulps, floor=0.0):
        got, want = got.detach().double().cpu().flatten()[:512], want.double()
        upd = float((want - old.double().flatten()[:512]).abs().max()) if old is not None else float(want.abs().max())
        tol = rel * upd + ulps * 6e-8 * max(1.0, float(want.abs().max())) + floor
        err = float((got - want).abs().max())
        assert err <= tol, (name, err, tol, upd)
    pnames = [n for n, _ in model.named_parameters()]
    for k, want in head["sd"].items():
        if want.is_floating_point():
            close("sd " + k, sd[k], want, ck["model"][k], REL, 8)
        else:
            assert torch.equal(sd[k].cpu().flatten()[:512], want), k
        check_digest("sd " + k, sd[k], g["after"]["sd"][k], rtol=2e-4, atol=1e-5)
    for i, (n, p) in enumerate(model.named_parameters()):
        st0 = ck["optimizer"]["state"][i]
        close("sq " + n, opt.state[p]["square_avg"], head["sq"][n], None, REL, 1e-5)   # floor 6e-13: squares of gradients that are rounding noise
        # floor: a gradient that is exactly zero in real arithmetic (BN bias in front of another BN) is ~5e-7 of rounding noise in
        # either implementation, and enters the buffer divided by sqrt(eps) = 0.03
        close("buf " + n, opt.state[p]["momentum_buffer"], head["buf"][n], st0["momentum_buffer"] * 0.9, REL, 16, floor=5e-5)
    for k, want in head["ema"].items():
        close("ema " + k, ema.average(k), want, ck["ema"]["shadow"][k], REL, 8)
    info = ema.state_dict()["info"]
    for k, v in g["after"]["ema_info"].items():   # the per-variable counters the reference checkpoints
        assert info[k]["num_updates"] == v["num_updates"] and abs(info[k]["last_momemtum"] - v["last_momemtum"]) < 1e-6, (k, info[k], v)
