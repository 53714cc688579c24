"""Dynamic shrinkage on the GPU against the golden fixture generated from the reference (tests/golden/shrink.pt):
masks (bit-exact), rebuilt weights / BN statistics, RMSprop state re-keying (incl. the append-on-shrink parameter order),
EMA shadows, PruneInfo order and penalties, exported architecture, MAC count, and the network function after the shrink."""
import collections
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from kutil import assert_close, check_digest, counter_fill  # noqa: E402

pytestmark = pytest.mark.gpu


def test_shrink_matches_reference(gpu_lib):
    sys.path.insert(0, ROOT)
    import train as T   # shrink_model lives in train.py, as in the reference
    from atomnas_amd import runtime
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import mobilenet_supernet as ms
    from atomnas_amd.utils import config, model_profiling as mp, optim as aopt, prune as aprune, rmsprop
    g = torch.load(os.path.join(ROOT, "tests", "golden", "shrink.pt"), weights_only=False)
    model = ms.Model(**g["kw"])
    model.set_compute_dtype(torch.float32)
    model.load_state_dict(g["sd_pre"])
    mp.model_profiling(model, 64, 64, verbose=False)
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
    mgr.ensure()
    with torch.no_grad():   # bring optimizer state and EMA shadows to the fixture's pre-shrink values
        for n, p in model.named_parameters():
            opt.state[p]["square_avg"].copy_(g["sq_pre"][n])
            opt.state[p]["momentum_buffer"].copy_(g["buf_pre"][n])
        for n in ema.average_names():
            ema.average(n).copy_(g["ema_pre"][n])

    # masks on device: |gamma| > thr OR |gamma_ema| > thr, bit-exact with the reference
    thr = 1e-3
    for bname, blk in model.get_named_block_list().items():
        m = [bn.weight.detach().abs() > thr for bn in blk.get_depthwise_bn()]
        me = [ema.average("{}.{}.weight".format(bname, nm)).detach().abs() > thr for nm in blk.get_named_depthwise_bn().keys()]
        for a, b, ref in zip(m, me, g["masks"][bname]):
            assert torch.equal((a | b).cpu(), ref), bname

    class F(dict):
        __getattr__ = dict.__getitem__
    config.FLAGS.bind(F(image_size=64, use_distributed=False))
    wrapper = torch.nn.Module()
    wrapper.module = model
    T.shrink_model(wrapper, ema, opt, pinfo, thr, ema_only=False)

    assert [n for n, _ in model.named_parameters()] == g["param_names_post"]
    id2name = {id(p): n for n, p in model.named_parameters()}
    assert [id2name[id(p)] for p in opt.param_groups[0]["params"]] == g["opt_order_post"]
    assert ema.average_names() == g["ema_names_post"]
    assert pinfo.weight == g["prune_weight_post"] and pinfo.penalty == g["prune_penalty_post"]
    assert mb.output_network(model) == g["output_network"]
    assert model.n_macs == g["n_macs_post"]
    sd = model.state_dict()
    assert list(sd.keys()) == list(g["sd_post"].keys())
    for k, dg in g["sd_post"].items():
        check_digest("post " + k, sd[k], dg, rtol=1e-6)
    for n, p in model.named_parameters():
        check_digest("sq " + n, opt.state[p]["square_avg"], g["sq_post"][n], rtol=1e-6)
        check_digest("buf " + n, opt.state[p]["momentum_buffer"], g["buf_post"][n], rtol=1e-6)
    for k, dg in g["ema_post"].items():
        check_digest("ema " + k, ema.average(k), dg, rtol=1e-6)

    # the shrunk network computes the reference's function (eval mode, running statistics) ...
    x = (counter_fill(torch.empty(4, 3, 64, 64), 400) * 4).float()
    model.eval()
    with torch.no_grad():
        logits = model(x.cuda())
    assert_close("logits after shrink", logits, g["logits_post"], rtol=2e-3, atol=2e-4)
    # ... and keeps training through the rebuilt arenas (ragged widths 32, 1, 95, ... and an empty block)
    from atomnas_amd import engine
    model.train()
    ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-5, batch_size=4, image_size=64, use_graph=True)
    ts.set_batch(x.cuda(), (torch.arange(4) % 10).cuda())
    l = []
    for _ in range(6):
        ts.step(lr=0.002, rho=1e-4)
        l.append(ts.loss[0].item())
    assert all(v == v for v in l) and l[-1] < l[0], l
    assert set(id(p) for p in opt.param_groups[0]["params"]) == set(id(p) for p in model.parameters())


def test_arena_rebuild_inside_a_gather_deferral_window(gpu_lib):
    """ArenaManager.materialize() while a shrink's gathers are still only RECORDED (ops.gather_defer): the pending gathers write the
    rebuilt modules' tensors, which the rebuild's value migration reads -- they must run first (one job-table launch has no order between
    its workgroups).  A forward, a profiling pass or an optimizer call inside the window triggers exactly that; the result must equal
    the ordinary order (flush, then rebuild) bit for bit."""
    sys.path.insert(0, ROOT)
    from atomnas_amd import ops, runtime
    from atomnas_amd.models import mobilenet_supernet as ms
    g = torch.load(os.path.join(ROOT, "tests", "golden", "shrink.pt"), weights_only=False)

    def shrunk(ensure_inside):
        model = ms.Model(**g["kw"])
        model.set_compute_dtype(torch.float32)
        model.load_state_dict(g["sd_pre"])
        model.cuda().train()
        mgr = runtime.manager_of(model)
        mgr.ensure()
        ops.gather_defer(True)
        try:
            for bname, blk in model.get_named_block_list().items():
                blk.compress_by_mask([m.cuda() for m in g["masks"][bname]], prefix=bname)
            assert mgr.dirty
            if ensure_inside:
                mgr.ensure()                      # rebuild with recorded, not yet executed gathers
                assert ops.gather_deferring()     # the window is still open afterwards
        finally:
            ops.gather_defer(False)
        mgr.ensure()
        torch.cuda.synchronize()
        return collections.OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())

    a, b = shrunk(False), shrunk(True)
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k
    for k, dg in g["sd_post"].items():
        check_digest("post " + k, b[k], dg, rtol=1e-6)


def test_arena_rebuild_restores_the_deferral_state_when_it_raises(gpu_lib, monkeypatch):
    sys.path.insert(0, ROOT)
    from atomnas_amd import ops, runtime
    from atomnas_amd.models import mobilenet_supernet as ms
    g = torch.load(os.path.join(ROOT, "tests", "golden", "shrink.pt"), weights_only=False)
    model = ms.Model(**g["kw"])
    model.cuda()
    mgr = runtime.manager_of(model)
    calls = []

    def boom(dst, src):
        calls.append(1)
        raise RuntimeError("injected")
    monkeypatch.setattr(ops, "copy_job", boom)
    with pytest.raises(RuntimeError, match="injected"):
        mgr.ensure()
    assert calls and not ops.gather_deferring()   # a failed rebuild must not leave gathers deferred for ever
