"""Debug helper: one eager training step of the full-size supernet, then lists non-finite gradients / parameters / buffers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomnas_amd.models import mobilenet_supernet as ms, mobilenet_base as mb
from atomnas_amd.utils import rmsprop, optim as aopt, prune as aprune, model_profiling as mp
from atomnas_amd import engine

SETTING = [[1, 16, 1, 1, [3]], [6, 24, 4, 2, [3, 5, 7]], [6, 40, 4, 2, [3, 5, 7]], [6, 80, 4, 2, [3, 5, 7]], [6, 96, 4, 1, [3, 5, 7]],
           [6, 192, 4, 2, [3, 5, 7]], [6, 320, 1, 1, [3, 5, 7]]]
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(1995)
model = ms.Model(inverted_residual_setting=SETTING, active_fn='nn.ReLU', batch_norm_momentum=0.01, batch_norm_epsilon=1e-3,
                 input_channel=32, input_size=224)
model.apply(mb.init_weights_mnas)
model.set_compute_dtype(torch.bfloat16)
mp.model_profiling(model, 224, 224, verbose=False)
model.cuda().train()
x = torch.randn(bs, 3, 224, 224, device='cuda'); y = torch.randint(0, 1000, (bs,), device='cuda')
if os.environ.get("DBG_WRAP"):
    from atomnas_amd import ops as _ops
    def _fin(t):
        if t is None: return True
        tt = t.t if isinstance(t, _ops.Slab) else t
        return bool(torch.isfinite(tt.float()).all())
    def wrap(name, innames, outnames):
        orig = getattr(_ops, name)
        import inspect
        sig = inspect.signature(orig)
        def f(*a, **kw):
            ba = sig.bind(*a, **kw); ba.apply_defaults()
            torch.cuda.synchronize()
            badin = [n for n in innames if not _fin(ba.arguments.get(n))]
            r = orig(*a, **kw)
            torch.cuda.synchronize()
            badout = [n for n in outnames if n != "stats" and not _fin(ba.arguments.get(n))]
            if "stats" in outnames and ba.arguments.get("stats") is not None and "stat_ld" in ba.arguments:
                st, ld, C, rows = ba.arguments["stats"], ba.arguments["stat_ld"], ba.arguments["C"], ba.arguments["stat_rows"]
                need = (rows - 1) * 2 * ld + ld + C
                v = st.reshape(-1)
                own = torch.stack([v[r * 2 * ld + pl * ld: r * 2 * ld + pl * ld + C] for r in range(rows) for pl in range(2)]).view(rows, 2, C)
                nf = ~torch.isfinite(own)
                if nf.any():
                    badrows = torch.nonzero(nf.any(2).any(1)).flatten().tolist()
                    badch = torch.nonzero(nf.any(0).any(0)).flatten().tolist()
                    c0 = badch[0]
                    print("    channel", c0, "plane0 rows0-9", own[:10, 0, c0].tolist(), "plane1 rows0-9", own[:10, 1, c0].tolist(), flush=True)
                    print("    channel", c0 + 1, "plane0 rows0-9", own[:10, 0, c0 + 1].tolist(), "plane1 rows0-9", own[:10, 1, c0 + 1].tolist(), flush=True)
                    print("    rows 10..127 all zero:", bool((own[10:] == 0).all()), " per-plane nonfinite counts", int(nf[:, 0].sum()), int(nf[:, 1].sum()))
                    badout.append("stats rows %s..(%d) channels %s..(%d) of rows=%d ld=%d" % (badrows[:6], len(badrows), badch[:8], len(badch), rows, ld))
            if name == "dwconv_fwd" and "y" in badout:
                y = ba.arguments["y"]; C = ba.arguments["C"]; Nn, Hh = ba.arguments["N"], ba.arguments["H"]
                yp = y.to_plain()[:, :C].float().reshape(Nn, -1, C)
                nf = ~torch.isfinite(yp)
                print("    y non-finite: images", torch.nonzero(nf.any(2).any(1)).flatten().tolist()[:12], "pixels", torch.nonzero(nf.any(2).any(0)).flatten().tolist()[:12],
                      "channels", torch.nonzero(nf.any(0).any(0)).flatten().tolist()[:12], "count", int(nf.sum()))
                xin = ba.arguments["x"].to_plain()[:, :C].float().reshape(Nn, -1, C)
                chs = torch.nonzero(nf.any(0).any(0)).flatten()[:2].tolist()
                for c in chs:
                    print("    channel", c, "x absmax", float(xin[:, :, c].abs().max()), "scale", float(ba.arguments["in_scale"][c]), "shift", float(ba.arguments["in_shift"][c]),
                          "taps", ba.arguments["w_taps"][:, c].tolist()[:9])
            if badin or badout:
                desc = {k: ba.arguments.get(k) for k in ("N", "H", "W", "C", "k", "stride", "M", "K") if k in ba.arguments}
                print("  [%s] %s bad inputs %s bad outputs %s" % (name, desc, badin, badout), flush=True)
            return r
        setattr(_ops, name, f)
        import atomnas_amd.functional as fn
    _orig_nt = _ops.gemm_nt
    def gemm_nt_chk(a, wp, c, M, N, K, **kw):
        r = _orig_nt(a, wp, c, M, N, K, **kw)
        torch.cuda.synchronize()
        ct = (c.to_plain() if isinstance(c, _ops.Slab) else c)[:, :N].float()
        am = float(ct.abs().max())
        at = (a.to_plain() if isinstance(a, _ops.Slab) else a)[:, :K].float()
        if not (am < 1e6):
            big = ct.abs() > 1e6
            rows = torch.nonzero(big.any(1)).flatten(); cols = torch.nonzero(big.any(0)).flatten()
            print("  [gemm_nt] M%d N%d K%d kw %s: |c| max %.3g, input |a| max %.3g; big rows %s (n=%d) cols %s (n=%d)" % (
                M, N, K, sorted(k for k, v in kw.items() if v is not None and v is not False and not (isinstance(v, int) and v == 0)), am, float(at.abs().max()), rows[:6].tolist(), rows.numel(), cols[:8].tolist(), cols.numel()), flush=True)
        return r
    _ops.gemm_nt = gemm_nt_chk
    wrap("dwconv_fwd", ["x", "in_scale", "in_shift", "w_taps"], ["stats", "y"])
    wrap("dwconv_bwd", ["g", "yraw", "c1", "c2", "c3", "x", "in_scale", "in_shift", "w_taps"], ["h", "stats"])
    wrap("bn_finalize_fwd", ["stats", "gamma", "beta"], ["scale", "shift", "save_mean", "save_invstd"])
    wrap("bn_finalize_bwd", ["stats2", "gamma", "save_mean", "save_invstd"], ["c1", "c2", "c3"])
for it in range(2):
    model.zero_grad()
    logits = model(x)
    loss = aopt.CrossEntropyLabelSmooth(1000, 0.1)(logits, y).mean()
    loss.backward()
    torch.cuda.synchronize()
    print("iter", it, "loss", float(loss))
    bad = 0
    for n, p in model.named_parameters():
        if p.grad is not None and not torch.isfinite(p.grad).all():
            k = int((~torch.isfinite(p.grad)).sum())
            print("  non-finite grad", n, tuple(p.shape), k, "of", p.numel(), "first idx", torch.nonzero(~torch.isfinite(p.grad.flatten()))[:6].flatten().tolist())
            bad += 1
    for n, b in model.named_buffers():
        if b.is_floating_point() and not torch.isfinite(b).all():
            print("  non-finite buffer", n, tuple(b.shape), int((~torch.isfinite(b)).sum()))
            bad += 1
    print("  bad tensors:", bad)
    if bad: break
if os.environ.get("DBG_TRAINSTEP"):
    pinfo = aprune.get_bn_to_prune(model, {'bn_prune_filter': 'expansion_only_skip_expand1'}, verbose=False)
    opt = rmsprop.RMSprop(model.parameters(), lr=0.016, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
    ema = aopt.ExponentialMovingAverage(0.9999)
    for n, p in model.named_parameters(): ema.register(n, p)
    for n, b in model.named_buffers():
        if 'running' in n: ema.register(n, b)
    ts = engine.TrainStep(model, opt, ema, pinfo, batch_size=bs, use_graph=False)
    ts.set_batch(x, y)
    for it in range(2):
        ts.step(rho=1e-5); torch.cuda.synchronize()
        print("trainstep", it, "loss", ts.loss.tolist())
        for n, p in model.named_parameters():
            if not torch.isfinite(p).all():
                print("  non-finite param", n, tuple(p.shape), int((~torch.isfinite(p)).sum()), "grad finite:", bool(torch.isfinite(p.grad).all()) if p.grad is not None else None)
        for n, b in model.named_buffers():
            if b.is_floating_point() and not torch.isfinite(b).all():
                print("  non-finite buffer", n, tuple(b.shape), int((~torch.isfinite(b)).sum()))
