"""Measured agreement of the HIP path with the oracle (GPU): the numbers the whole-model parity tests take their bounds from.

    python tools/parity_diag.py            -> profiles/r03_parity_diag.txt
  (a) tiny supernet 64x64, N = 6, fp32 and bf16 storage: logits, loss, per-tensor and aggregate gradient agreement;
  (b) tiny supernet with dropout 0.2 (N = 32): keep rate, logits / gradients against the oracle fed with the kernel's own mask;
  (c) full-size AtomNAS-C supernet, bf16 storage, batch 16, train mode, dropout 0.2: per-block output relative L2 against
      Bf16Storage, logits, loss, aggregate gradient relative L2 / cosine.
"""
import collections
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atomnas_oracle as orc  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from kutil import bf16_storage  # noqa: E402  (Bf16Storage + the matrix-core depthwise restatement where the library runs those kernels)


def sd64(model):
    return collections.OrderedDict((k, v.detach().clone().double() if v.is_floating_point() else v.clone()) for k, v in model.state_dict().items())


def run_pair(model, x, y, dtype, num_classes, p_drop):
    """-> dict(logits, ref_logits, loss, ref_loss, grads {name: (got, ref)}, feats [(got, ref)], keep)"""
    from atomnas_amd import functional as AF
    from atomnas_amd.utils import optim as aopt
    sd0 = sd64(model)
    spec = orc.spec_from_model(model)
    model.cuda().train()
    drop = list(model.classifier.children())[0]
    drop.p = p_drop
    outs = []
    hooks = [m.register_forward_hook(lambda mod, i, o: outs.append(o.detach())) for m in list(model.features.children())[:-2]]
    AF.TAIL_TAP = []
    logits = model(x.cuda())
    keep = AF.TAIL_TAP[0] if AF.TAIL_TAP else None
    AF.TAIL_TAP = None
    for h in hooks:
        h.remove()
    loss = aopt.CrossEntropyLabelSmooth(num_classes, 0.1, reduction="none")(logits, y.cuda()).mean()
    loss.backward()
    torch.cuda.synchronize()
    work = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v) for k, v in sd0.items()}
    q = orc.NoQuant if dtype == torch.float32 else bf16_storage()
    xin = x.bfloat16().double() if dtype == torch.bfloat16 else x.double()
    mask = None if keep is None else keep.double().cpu() / (1.0 - p_drop)
    t0 = time.perf_counter()
    ref_logits, feats = orc.model_forward(xin, work, spec, True, {}, dropout_mask=mask, return_features=True, q=q)
    ref_loss = orc.ce_label_smooth(ref_logits, y, 0.1).mean()
    ref_loss.backward()
    dt = time.perf_counter() - t0
    grads = collections.OrderedDict((n, (p.grad.double().cpu(), work[n].grad)) for n, p in model.named_parameters())
    return dict(logits=logits.double().cpu(), ref_logits=ref_logits.detach(), loss=float(loss.detach()), ref_loss=float(ref_loss.detach()),
                grads=grads, feats=[(a.double().cpu(), b.detach()) for a, b in zip(outs, feats)], keep=keep, oracle_s=dt)


def teacher_forced(model, x, y, num_classes, p_drop, oracle_dtype=torch.float32, storage=None, oracle_device="cpu"):
    """Per-layer parity of a deep bf16 network without the chaos of the whole chain: the oracle (Bf16Storage) runs the full
    training forward / backward once; then every block of the HIP model is run ALONE on the oracle's input of that block and on the
    oracle's gradient of its output.  Returns rows (name, out rel-L2, input-grad rel-L2, param-grad rel-L2, param-grad cosine).
    storage: the oracle's storage model (default: Bf16Storage with the matrix-core operand roundings where the LIBRARY says it runs those
    kernels; pass orc.Bf16Storage() for the predicate-independent model: bf16 tensors, fp32 arithmetic, no operand roundings).
    oracle_device: "cuda" runs the oracle's torch ops on the GPU (ATen / MIOpen: still not this library) -- what makes batch 256 feasible."""
    from atomnas_amd import runtime
    od = torch.device(oracle_device)
    sd0 = collections.OrderedDict((k, v.detach().to(od).clone().to(oracle_dtype) if v.is_floating_point() else v.detach().to(od).clone())
                                  for k, v in model.state_dict().items())
    x, y = x.to(od), y.to(od)
    spec = orc.spec_from_model(model)
    work = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v) for k, v in sd0.items()}
    t0 = time.perf_counter()
    ref_logits, feats = orc.model_forward(x.bfloat16().to(oracle_dtype), work, spec, True, {}, dropout_mask=None, return_features=True, q=storage if storage is not None else bf16_storage())
    for f in feats:
        f.retain_grad()
    orc.ce_label_smooth(ref_logits, y, 0.1).mean().backward()
    oracle_s = time.perf_counter() - t0
    model.cuda().train()
    mgr = runtime.manager_of(model)
    mgr.ensure()
    blocks = list(model.features.children())[1:-2]
    names = [n for n, _ in list(model.features.named_children())[1:-2]]
    rows = []
    for i, (name, blk) in enumerate(zip(names, blocks)):
        if not len(blk.channels):
            continue
        xin = feats[i].detach().to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
        gout = feats[i + 1].grad.to(torch.bfloat16).cuda()
        mgr.zero_grad()
        mgr.enter()
        try:
            out = blk(xin)
            out.backward(gout)
        finally:
            mgr.leave()
        torch.cuda.synchronize()
        pg = torch.cat([p.grad.double().cpu().flatten() for _, p in blk.named_parameters()])
        pr = torch.cat([work["features.%s.%s" % (name, n)].grad.double().cpu().flatten() for n, _ in blk.named_parameters()])
        rows.append(("features." + name, rel_l2(out.double().cpu(), feats[i + 1].detach().double().cpu()),
                     rel_l2(xin.grad.double().cpu(), feats[i].grad.double().cpu()), rel_l2(pg, pr),
                     float(torch.dot(pg, pr) / (pg.norm() * pr.norm()))))
    return rows, oracle_s


NARROW = {"features.7.ops.0.": 1, "features.9.ops.2.": 3, "features.12.ops.1.": 13, "features.16.ops.0.": 2}


def shrunk_atomnas_a(seed=7):
    """BASELINE config 3 as tests/test_configs_gpu.py builds it: the full-size AtomNAS-A supernet with a seeded 30 % of its atoms (a whole
    middle branch of one block and a whole block among them) forced dead, four branches cut down to 1 / 2 / 3 / 13 atoms, then
    train.shrink_model -- ragged hidden widths, very narrow segments, a dropped branch, an empty block.  Returns the shrunk model (fp32 storage, on the GPU)."""
    import train as T
    from atomnas_amd import configs, runtime
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import mobilenet_supernet as ms
    from atomnas_amd.utils import config, model_profiling as mp, optim as aopt, prune as aprune, rmsprop
    from test_block_gpu import _randomize
    torch.manual_seed(seed)
    model = ms.Model(**dict(configs.model_kwparams("atomnas_a_supernet"), input_size=224))
    model.set_compute_dtype(torch.float32)
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
    g = torch.Generator().manual_seed(11)
    table = dict(model.named_parameters())
    with torch.no_grad():
        for name in pinfo.weight:
            w = table[name]
            dead = torch.rand(w.numel(), generator=g) < 0.3
            if name.startswith("features.3.ops.1.") or name.startswith("features.5."):
                dead[:] = True
            for prefix, keep in NARROW.items():   # very narrow segments: 1, 2, 3, 13 atoms (slab padding, narrow-GEMM dispatch)
                if name.startswith(prefix):
                    dead[:] = True
                    dead[:keep] = False
            w[dead.cuda()] = 0.0
            ema.average(name)[dead.cuda()] = 0.0

    class F(dict):
        __getattr__ = dict.__getitem__
    config.FLAGS.bind(F(image_size=224, use_distributed=False))
    wrapper = torch.nn.Module()
    wrapper.module = model
    T.shrink_model(wrapper, ema, opt, pinfo, 1e-3, ema_only=False)
    return model


def atomnas_c_plus():
    """BASELINE config 5: the searched AtomNAS-C architecture with SE (ratio 0.5), Swish and fused blocks, full size, random init"""
    from atomnas_amd import configs
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import searched_network as sn
    from test_block_gpu import _randomize
    torch.manual_seed(5)
    model = sn.Model(**dict(configs.searched_kwparams("atomnas_c_plus"), input_size=224, dropout_ratio=0.0))
    model.apply(mb.init_weights_mnas)
    _randomize(model, 15)
    return model


def rel_l2(a, b):
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def summarize(tag, r, out):
    ga = torch.cat([g.flatten() for g, _ in r["grads"].values()])
    ra = torch.cat([q.flatten() for _, q in r["grads"].values()])
    cos = float(torch.dot(ga, ra) / (ga.norm() * ra.norm()))
    out("%s: logits rel-L2 %.3e max|d| %.3e; loss %.6f vs %.6f; gradients: aggregate rel-L2 %.3e cosine %.6f norm ratio %.4f (oracle %.1f s)"
        % (tag, rel_l2(r["logits"], r["ref_logits"]), float((r["logits"] - r["ref_logits"]).abs().max()), r["loss"], r["ref_loss"],
           rel_l2(ga, ra), cos, float(ga.norm() / ra.norm()), r["oracle_s"]))
    gmax = max(float(q.norm()) for _, q in r["grads"].values())
    worst = []
    for n, (g, q) in r["grads"].items():
        if float(q.norm()) > 1e-2 * gmax:
            c = float(torch.dot(g.flatten(), q.flatten()) / (g.norm() * q.norm()))
            worst.append((rel_l2(g, q), c, float(g.norm() / q.norm()), n))
    worst.sort(reverse=True)
    out("    significant tensors %d: worst rel-L2 %.3e (%s); min cosine %.5f; norm ratio range %.3f..%.3f" %
        (len(worst), worst[0][0], worst[0][3], min(w[1] for w in worst), min(w[2] for w in worst), max(w[2] for w in worst)))
    if r["feats"]:
        fl = [rel_l2(a, b) for a, b in r["feats"]]
        out("    block outputs rel-L2: first %.3e  max %.3e  last %.3e   [%s]" % (fl[0], max(fl), fl[-1], " ".join("%.1e" % v for v in fl)))


def main():
    from test_block_gpu import TINY, _randomize
    from atomnas_amd import configs
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import mobilenet_supernet as ms
    lines = []

    def out(s):
        print(s, flush=True)
        lines.append(s)
    only_d = bool(os.environ.get("PARITY_ONLY_D"))
    N = int(os.environ.get("PARITY_N", "16"))
    g = torch.Generator().manual_seed(8)
    x, y = torch.randn(N, 3, 224, 224, generator=g), torch.randint(0, 1000, (N,), generator=g)
    for dtype in (() if only_d else (torch.float32, torch.bfloat16)):
        model = ms.Model(**TINY)
        model.set_compute_dtype(dtype)
        _randomize(model, 5)
        g = torch.Generator().manual_seed(3)
        x, y = torch.randn(6, 3, 64, 64, generator=g), torch.randint(0, 10, (6,), generator=g)
        summarize("(a) tiny 64x64 N=6 %s" % str(dtype).split(".")[1], run_pair(model, x, y, dtype, 10, 0.0), out)
    for dtype in (() if only_d else (torch.float32, torch.bfloat16)):
        model = ms.Model(**dict(TINY, dropout_ratio=0.2))
        model.set_compute_dtype(dtype)
        _randomize(model, 5)
        g = torch.Generator().manual_seed(4)
        x, y = torch.randn(32, 3, 64, 64, generator=g), torch.randint(0, 10, (32,), generator=g)
        r = run_pair(model, x, y, dtype, 10, 0.2)
        k = r["keep"].float()
        out("(b) dropout 0.2: keep rate %.4f over %d units (3 sigma = %.4f); per-sample kept min %d max %d of %d" %
            (float(k.mean()), k.numel(), 3 * (0.2 * 0.8 / k.numel()) ** 0.5, int(k.sum(1).min()), int(k.sum(1).max()), k.shape[1]))
        summarize("(b) tiny 64x64 N=32 dropout %s" % str(dtype).split(".")[1], r, out)
    torch.manual_seed(3)
    model = ms.Model(**dict(configs.model_kwparams("atomnas_c_supernet"), input_size=224))
    model.set_compute_dtype(torch.bfloat16)
    model.apply(mb.init_weights_mnas)
    if not only_d:
        summarize("(c) full-size AtomNAS-C supernet bf16 N=%d dropout 0.2" % N, run_pair(model, x, y, torch.bfloat16, 1000, 0.2), out)
    torch.manual_seed(3)
    model = ms.Model(**dict(configs.model_kwparams("atomnas_c_supernet"), input_size=224))
    model.set_compute_dtype(torch.bfloat16)
    model.apply(mb.init_weights_mnas)
    rows, osec = teacher_forced(model, x, y, 1000, 0.0)
    out("(d) full-size AtomNAS-C supernet bf16 N=%d, every block alone on the oracle's input / output gradient (oracle fp32, %.1f s):" % (N, osec))
    for r in rows:
        out("    %-12s out rel-L2 %.3e   input-grad rel-L2 %.3e   param-grad rel-L2 %.3e cosine %.6f" % r)
    out("    worst: out %.3e  input-grad %.3e  param-grad %.3e  min cosine %.6f" % (max(r[1] for r in rows), max(r[2] for r in rows),
        max(r[3] for r in rows), min(r[4] for r in rows)))
    # (e) / (f): the same per-block statement for the other bf16 configurations (their bf16 legs were property-only)
    g = torch.Generator().manual_seed(12)
    N2 = int(os.environ.get("PARITY_N2", "8"))
    x2, y2 = torch.randn(N2, 3, 224, 224, generator=g), torch.randint(0, 1000, (N2,), generator=g)
    for tag, make in (("(e) AtomNAS-A supernet after the forced 30 %% shrink (ragged widths, dropped branch, empty block)", shrunk_atomnas_a),
                      ("(f) full-size AtomNAS-C+ (fused blocks, SE, Swish)", atomnas_c_plus)):
        model = make()
        model.set_compute_dtype(torch.bfloat16)
        rows, osec = teacher_forced(model, x2, y2, 1000, 0.0)
        out("%s bf16 N=%d, every block alone on the oracle's input / output gradient (oracle fp32, %.1f s):" % (tag, N2, osec))
        for r in rows:
            out("    %-12s out rel-L2 %.3e   input-grad rel-L2 %.3e   param-grad rel-L2 %.3e cosine %.6f" % r)
        out("    worst: out %.3e  input-grad %.3e  param-grad %.3e  min cosine %.6f" % (max(r[1] for r in rows), max(r[2] for r in rows),
            max(r[3] for r in rows), min(r[4] for r in rows)))
    with open(os.path.join(ROOT, "gpurun_out", "parity_diag.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
