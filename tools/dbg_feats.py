import sys, os, collections, torch
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _R); sys.path.insert(0, os.path.join(_R, "oracle")); sys.path.insert(0, os.path.join(_R, "tests"))
import atomnas_oracle as orc
from test_block_gpu import TINY, _randomize, _sd64
from atomnas_amd.models import mobilenet_supernet as ms
from atomnas_amd.utils import optim as aopt
dtype = torch.bfloat16
model = ms.Model(**TINY); model.set_compute_dtype(dtype); _randomize(model, 5)
sd0 = _sd64(model); spec = orc.spec_from_model(model)
g = torch.Generator().manual_seed(3); N = 6
x = torch.randn(N, 3, 64, 64, generator=g); y = torch.randint(0, 10, (N,), generator=g)
model.cuda().train()
outs, grads = {}, {}
def mk(name):
    def hook(m, i, o):
        outs[name] = o.detach()
        o.register_hook(lambda gr: grads.__setitem__(name, gr.detach()))
    return hook
for i, m in enumerate(list(model.features.children())[:-2]):
    m.register_forward_hook(mk(i))
crit = aopt.CrossEntropyLabelSmooth(10, 0.1, reduction="none")
logits = model(x.cuda()); loss = crit(logits, y.cuda()).mean(); loss.backward(); torch.cuda.synchronize()
work = {k: (v.clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v) for k, v in sd0.items()}
ref_logits, feats = orc.model_forward(x.bfloat16().double(), work, spec, True, {}, q=orc.Bf16Storage, return_features=True)
for f in feats: f.retain_grad()
orc.ce_label_smooth(ref_logits, y, 0.1).mean().backward()
rel = lambda a, b: float((a.double().cpu() - b).norm() / b.norm().clamp_min(1e-30))
for i, f in enumerate(feats):
    print(i, tuple(f.shape), "fwd relL2 %.5f  nexact %.4f | bwd relL2 %.5f" % (rel(outs[i], f.detach()), float((outs[i].double().cpu() != f.detach()).double().mean()), rel(grads[i], f.grad)))
