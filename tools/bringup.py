"""Bring-up / per-kernel timing of the full-size supernet step (not a test; run on the GPU box)."""
import os, sys, time, json, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atomnas_amd.models import mobilenet_supernet as ms, mobilenet_base as mb
from atomnas_amd.utils import rmsprop, optim as aopt, prune as aprune, model_profiling as mp
from atomnas_amd import engine, _lib

SETTING = [[1, 16, 1, 1, [3]], [6, 24, 4, 2, [3, 5, 7]], [6, 40, 4, 2, [3, 5, 7]], [6, 80, 4, 2, [3, 5, 7]], [6, 96, 4, 1, [3, 5, 7]],
           [6, 192, 4, 2, [3, 5, 7]], [6, 320, 1, 1, [3, 5, 7]]]
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dtype = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
if os.environ.get("PREALLOC_GB"):   # experiment: one large allocator segment up front (physical contiguity / TLB reach)
    _big = torch.empty(int(float(os.environ["PREALLOC_GB"]) * 2**30), dtype=torch.uint8, device="cuda"); del _big
torch.manual_seed(1995)
model = ms.Model(inverted_residual_setting=SETTING, active_fn='nn.ReLU', batch_norm_momentum=0.01, batch_norm_epsilon=1e-3,
                 input_channel=32, input_size=224)
model.apply(mb.init_weights_mnas)
model.set_compute_dtype(dtype)
mp.model_profiling(model, 224, 224, verbose=False)
print("macs", model.n_macs, "params", model.n_params)
model.cuda().train()
pinfo = aprune.get_bn_to_prune(model, {'bn_prune_filter': 'expansion_only_skip_expand1'}, verbose=False)
print("prunable", len(pinfo.weight), "pen0", pinfo.penalty[0], "penlast", pinfo.penalty[-1])
opt = rmsprop.RMSprop(model.parameters(), lr=0.016, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
ema = aopt.ExponentialMovingAverage(0.9999)
for n, p in model.named_parameters(): ema.register(n, p)
for n, b in model.named_buffers():
    if 'running' in n: ema.register(n, b)
ts = engine.TrainStep(model, opt, ema, pinfo, batch_size=bs, use_graph=False)
ts.set_batch(torch.randn(bs, 3, 224, 224, device='cuda'), torch.randint(0, 1000, (bs,), device='cuda'))
t0 = time.time(); ts.step(rho=1e-5); torch.cuda.synchronize(); print("first eager step %.2fs" % (time.time() - t0), "loss", ts.loss.tolist())
t0 = time.time()
for _ in range(3): ts.step(rho=1e-5)
torch.cuda.synchronize(); print("eager ms/step %.1f" % ((time.time() - t0) / 3 * 1e3), "loss", ts.loss.tolist(), "mem GB %.1f" % (torch.cuda.max_memory_allocated() / 2**30))
# per-kernel profile (eager, events around each C-ABI call)
_lib.PROFILE = []
ts.step(rho=1e-5); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, tag, e0, e1 in _lib.PROFILE:
    a = agg.setdefault(name, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
prof = _lib.PROFILE
_lib.PROFILE = None
tot = sum(v[1] for v in agg.values())
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("  %-28s n=%4d  %8.3f ms  %5.1f%%" % (k, v[0], v[1], 100 * v[1] / tot))
print("  sum of kernel times %.2f ms" % tot)
if os.environ.get("DETAIL"):
    det = collections.OrderedDict()
    for name, tag, e0, e1 in prof:
        if name in ("atomnas_dwconv_bwd", "atomnas_dwconv_fwd", "atomnas_pw_gemm_nt", "atomnas_pw_gemm_tn", "atomnas_expand_bwd", "atomnas_project_bwd"):
            a = det.setdefault((name, tag), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
    for (name, tag), v in det.items(): print("    %-22s %-40s n=%2d %8.3f ms" % (name[8:], tag, v[0], v[1]))
ts.use_graph = True
t0 = time.time(); ts.step(rho=1e-5); torch.cuda.synchronize(); print("capture %.2fs" % (time.time() - t0))
for _ in range(3): ts.step(rho=1e-5)
torch.cuda.synchronize(); t0 = time.time()
K = 10
for _ in range(K): ts.step(rho=1e-5)
torch.cuda.synchronize(); dt = (time.time() - t0) / K
print("graph ms/step %.2f  img/s %.0f" % (dt * 1e3, bs / dt), "loss", ts.loss.tolist(), "top", ts.topk.tolist())
