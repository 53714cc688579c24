"""Diagnostic (GPU): the resumed step of tests/golden/checkpoint_ref.pt -- how far does the HIP fp32 step land from the oracle in
fp64, compared with how far the oracle in fp32 (= the reference's own ATen arithmetic, bit for bit) lands from fp64?
Prints, per parameter tensor, the error of the UPDATE (new - old) relative to the update's norm; decides the tolerance of
tests/test_configs_gpu.py::test_reference_written_checkpoint_resumes_identically."""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import atomnas_oracle as orc  # noqa: E402
import train as T  # noqa: E402
from kutil import counter_fill  # noqa: E402
from atomnas_amd import engine  # noqa: E402
from atomnas_amd.models import mobilenet_base as mb, mobilenet_supernet as ms  # noqa: E402
from atomnas_amd.utils import model_profiling as mp, optim as aopt, prune as aprune, rmsprop  # noqa: E402

g = torch.load(os.path.join(ROOT, "tests", "golden", "checkpoint_ref.pt"), weights_only=False)
ck, kw = g["checkpoint"], g["kw"]
step = 2
x = (counter_fill(torch.empty(6, 3, 64, 64), 700 + step) * 4).float()
y = (torch.arange(6) * 3 + step) % 10
hp = dict(lr=0.002 * (1 + step), rho=1e-3 * (1 + step), weight_decay=1e-3, wd_method="mnas", label_smoothing=0.1, alpha=0.9, eps=1e-3,
          momentum=0.9, ema_decay=orc.ema_decay(0.99, step + 1))


def oracle(dt):
    sd = collections.OrderedDict((k, v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in ck["model"].items())
    model = ms.Model(**kw)
    spec = orc.spec_from_model(model)
    names, pen, _ = orc.prune_penalties(spec, kw["input_size"])
    pn = [k for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    st = {k: dict(square_avg=ck["optimizer"]["state"][i]["square_avg"].clone().to(dt), momentum_buffer=ck["optimizer"]["state"][i]["momentum_buffer"].clone().to(dt))
          for i, k in enumerate(pn)}
    ema = collections.OrderedDict((k, v.clone().to(dt)) for k, v in ck["ema"]["shadow"].items())
    r = orc.train_step(sd, spec, st, ema, x.to(dt), y, hp, names, pen)
    return sd, r, pn


s64, r64, pn = oracle(torch.float64)
s32, r32, _ = oracle(torch.float32)

torch.manual_seed(99)
model = ms.Model(**kw); model.apply(mb.init_weights_mnas); model.set_compute_dtype(torch.float32)
mp.model_profiling(model, 64, 64, verbose=False); model.cuda().train()
wrapper = torch.nn.Module(); wrapper.module = model
pinfo = aprune.get_bn_to_prune(model, {"bn_prune_filter": "expansion_only_skip_expand1"}, verbose=False)
opt = rmsprop.RMSprop(wrapper.parameters(), lr=0.002, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
ema = aopt.ExponentialMovingAverage(0.99)
for n, p in model.named_parameters(): ema.register(n, p)
for n, b in model.named_buffers():
    if "running" in n: ema.register(n, b)
T.load_checkpoint(ck, wrapper, opt, ema)
ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-3, label_smoothing=0.1, batch_size=6, image_size=64, use_graph=False)
ts.global_step = 2
ts.set_batch(x.cuda(), y.cuda())
ts.step(lr=hp["lr"], rho=hp["rho"])
torch.cuda.synchronize()
print("loss hip %.7f o32 %.7f o64 %.7f ref %.7f" % (ts.loss[0].item(), r32["loss"], r64["loss"], g["losses"][2]))
sd = model.state_dict()
rows = []
for k in pn:
    old = ck["model"][k].double()
    u64 = s64[k].double() - old
    eh = (sd[k].double().cpu() - old) - u64
    e32 = (s32[k].double() - old) - u64
    nrm = max(float(u64.norm()), 1e-30)
    gh = (dict(model.named_parameters())[k].grad.double().cpu())
    g64 = r64["grads"][k]
    if g64.dim() in (2, 4) or (g64.dim() == 1 and "classifier" in k):
        g64 = g64 - 1e-3 * r64["params_before"][k]
        g32 = r32["grads"][k].double() - 1e-3 * r32["params_before"][k].double()
    else:
        g32 = r32["grads"][k].double()
    gn = max(float(g64.norm()), 1e-30)
    rows.append((float(eh.norm()) / nrm, float(e32.norm()) / nrm, float((gh - g64).norm()) / gn, float((g32 - g64).norm()) / gn, k, old.numel(), float(eh.abs().max()), float(e32.abs().max())))
rows.sort(reverse=True)
print("update error rel. to |update| (hip vs o64, o32 vs o64), gradient error rel (hip, o32), name, numel, max abs update err (hip, o32)")
for r in rows[:25]:
    print("  %.2e %.2e | %.2e %.2e  %-40s %6d  %.2e %.2e" % r)
import statistics
print("median hip %.2e o32 %.2e; max hip %.2e o32 %.2e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows), max(r[0] for r in rows), max(r[1] for r in rows)))
