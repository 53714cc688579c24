"""debug: where does the resumed step leave the reference's continuation? (GPU)"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import train as T
from atomnas_amd import engine
from atomnas_amd.models import mobilenet_base as mb, mobilenet_supernet as ms
from atomnas_amd.utils import model_profiling as mp, optim as aopt, prune as aprune, rmsprop
from kutil import counter_fill
g = torch.load(os.path.join(ROOT, "tests", "golden", "checkpoint_ref.pt"), weights_only=False)
torch.manual_seed(99)
model = ms.Model(**g["kw"]); model.apply(mb.init_weights_mnas); model.set_compute_dtype(torch.float32)
mp.model_profiling(model, 64, 64, verbose=False); model.cuda().train()
wrapper = torch.nn.Module(); wrapper.module = model
pinfo = aprune.get_bn_to_prune(model, {"bn_prune_filter": "expansion_only_skip_expand1"}, verbose=False)
opt = rmsprop.RMSprop(wrapper.parameters(), lr=0.002, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
ema = aopt.ExponentialMovingAverage(0.99)
for n, p in model.named_parameters(): ema.register(n, p)
for n, b in model.named_buffers():
    if "running" in n: ema.register(n, b)
T.load_checkpoint(g["checkpoint"], wrapper, opt, ema)
ck = g["checkpoint"]
names = [n for n, _ in model.named_parameters()]
print("groups", {k: v for k, v in opt.param_groups[0].items() if k != "params"})
ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-3, label_smoothing=0.1, batch_size=6, image_size=64, use_graph=False)
for i, (n, p) in enumerate(model.named_parameters()):
    st = ck["optimizer"]["state"][i]
    d1 = float((opt.state[p]["square_avg"].cpu() - st["square_avg"]).abs().max()); d2 = float((opt.state[p]["momentum_buffer"].cpu() - st["momentum_buffer"]).abs().max())
    d3 = float((p.detach().cpu() - ck["model"][n]).abs().max())
    if i < 3 or d1 > 1e-7 or d2 > 1e-7 or d3 > 1e-7:
        print("after init", n, "sq diff", d1, "buf diff", d2, "param diff", d3, "is arena view", opt.state[p]["square_avg"].data_ptr() >= ts.mgr.SQ.data_ptr())
    if i > 6 and d1 < 1e-7: break
ts.global_step = 2
x = (counter_fill(torch.empty(6, 3, 64, 64), 702) * 4).float(); y = (torch.arange(6) * 3 + 2) % 10
ts.set_batch(x.cuda(), y.cuda())
p0 = {n: p.detach().clone() for n, p in model.named_parameters()}
ts.step(lr=0.006, rho=3e-3)
torch.cuda.synchronize()
print("loss", ts.loss.tolist(), g["losses"])
n0 = names[0]; p = dict(model.named_parameters())[n0]
print("hyper", ts.mgr.hyper.tolist())
print(n0, "delta sum", float((p.detach() - p0[n0]).sum()), "new sum", float(p.sum()), "ref sum", g["after"]["sd"][n0]["sum"], "old sum", float(p0[n0].sum()))
print("grad sum", float(p.grad.sum()), "sq sum", float(opt.state[p]["square_avg"].sum()), g["after"]["sq"][n0]["sum"], "buf sum", float(opt.state[p]["momentum_buffer"].sum()), g["after"]["buf"][n0]["sum"])
