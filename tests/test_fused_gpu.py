"""AtomNAS+ on the GPU (BASELINE config 5): InvertedResidualChannelsFused with Swish and
SqueezeAndExcitation (models/mobilenet_base.py:72-117,145-274) and MobileNetSearched (models/searched_network.py) built from
them -- against the fixture generated from the reference itself (tests/golden/fused_se.pt) and against the oracle."""
import collections
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import atomnas_oracle as orc  # noqa: E402

from kutil import assert_close, check_digest  # noqa: E402
from test_block_gpu import _randomize, _sd64  # noqa: E402

pytestmark = pytest.mark.gpu


def _golden():
    return torch.load(os.path.join(ROOT, "tests", "golden", "fused_se.pt"), weights_only=False)


@pytest.mark.parametrize("key", ["block0", "block1", "block2"])
def test_fused_block_matches_reference_fixture(gpu_lib, key):
    """fp32 storage: output, input gradient, every parameter gradient (incl. the SE weights and biases), running statistics and
    the eval-mode output of the reference's own fused block."""
    from atomnas_amd.models import mobilenet_base as mb
    d = _golden()[key]
    cfg = d["cfg"]
    blk = mb.InvertedResidualChannelsFused(cfg["inp"], cfg["oup"], cfg["stride"], cfg["channels"], cfg["ks"], cfg["expand"],
                                           active_fn=mb.get_active_fn(cfg["act"]), batch_norm_kwargs={"momentum": 0.01, "eps": 1e-3},
                                           se_ratio=cfg["se_ratio"])
    assert list(blk.state_dict().keys()) == list(d["sd"].keys())
    blk.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in d["sd"].items()})
    blk.compute_dtype = torch.float32
    blk.cuda().train()
    x = d["x"].float().cuda().requires_grad_(True)
    out = blk(x)
    out.backward(d["gout"].float().cuda())
    torch.cuda.synchronize()
    assert_close("out", out, d["out"], rtol=1e-3, atol=1e-4 * max(1.0, float(d["out"].abs().max())))
    assert_close("dx", x.grad, d["dx"], rtol=2e-3, atol=2e-4 * max(1.0, float(d["dx"].abs().max())))
    for n, p in blk.named_parameters():
        r = d["grads"][n]
        assert_close("grad " + n, p.grad, r, rtol=2e-3, atol=3e-4 * max(1e-2, float(r.abs().max())))
    sd = blk.state_dict()
    for k, v in d["sd_after"].items():
        if "running" in k:
            assert_close(k, sd[k], v, rtol=2e-3, atol=1e-4)
        elif "num_batches" in k:
            assert int(sd[k]) == int(v) == 1
    blk.eval()
    with torch.no_grad():
        ev = blk(d["x"].float().cuda())
    assert_close("eval", ev, d["out_eval"], rtol=2e-3, atol=2e-4 * max(1.0, float(d["out_eval"].abs().max())))


def test_searched_network_with_fused_se_blocks(gpu_lib):
    """MobileNetSearched with fused SE blocks, Swish, ragged widths ([15, 23, 13], [40, 9], [100]): logits, loss and every
    gradient (digests) of the reference's training-mode step; then bf16 training steps through the captured graphs."""
    from atomnas_amd import engine
    from atomnas_amd.models import searched_network as sn
    from atomnas_amd.utils import optim as aopt, rmsprop
    n = _golden()["net"]
    model = sn.Model(**n["kw"])
    model.load_state_dict({k: v.float() if v.is_floating_point() else v for k, v in n["sd"].items()})
    model.set_compute_dtype(torch.float32)
    model.cuda().train()
    logits = model(n["x"].float().cuda())
    loss = aopt.CrossEntropyLabelSmooth(10, 0.1, reduction="none")(logits, n["target"].cuda()).mean()
    loss.backward()
    torch.cuda.synchronize()
    assert_close("logits", logits, n["logits"], rtol=2e-3, atol=2e-4 * max(1.0, float(n["logits"].abs().max())))
    assert abs(float(loss.detach()) - n["loss"]) < 1e-4
    for k, p in model.named_parameters():
        check_digest("grad " + k, p.grad, n["grad_digests"][k], rtol=2e-3)
    model.eval()
    with torch.no_grad():
        assert_close("eval logits", model(n["x"].float().cuda()), n["logits_eval"], rtol=2e-3,
                     atol=2e-4 * max(1.0, float(n["logits_eval"].abs().max())))
    # bf16 storage + graph: a memorisable batch must be learnt
    model2 = sn.Model(**n["kw"])
    _randomize(model2, 77)
    model2.cuda().train()
    opt = rmsprop.RMSprop(model2.parameters(), lr=0.003, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
    ema = aopt.ExponentialMovingAverage(0.99)
    for k, p in model2.named_parameters():
        ema.register(k, p)
    for k, b in model2.named_buffers():
        if "running" in k:
            ema.register(k, b)
    ts = engine.TrainStep(model2, opt, ema, None, weight_decay=1e-5, batch_size=16, image_size=64, use_graph=True)
    g = torch.Generator().manual_seed(3)
    ts.set_batch(torch.randn(16, 3, 64, 64, generator=g).cuda(), torch.randint(0, 10, (16,), generator=g).cuda())
    losses = []
    for _ in range(40):
        ts.step(lr=0.003)
        losses.append(ts.loss[0].item())
    assert all(v == v for v in losses) and losses[-1] < 0.6 * losses[0], losses


def test_cfg5_atomnas_c_plus_full_size(gpu_lib):
    """BASELINE config 5: AtomNAS-C+ (apps/searched/atomnas_c/atomnas_c+.yml: searched C architecture, SE ratio 0.5, Swish, fused
    blocks) at full size -- 362.9 MMACs / 5.93 M parameters as the reference's profiler reports, forward + backward against the
    float64 oracle in fp32 storage, and bf16 training steps at a small batch."""
    from atomnas_amd import configs, engine
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import searched_network as sn
    from atomnas_amd.utils import model_profiling as mp, optim as aopt, rmsprop
    kw = dict(configs.searched_kwparams("atomnas_c_plus"), input_size=224, dropout_ratio=0.0)
    torch.manual_seed(5)
    model = sn.Model(**kw)
    model.apply(mb.init_weights_mnas)
    _randomize(model, 15)
    macs, params = mp.model_profiling(model, 224, 224, verbose=False)
    assert macs == 362910842 and params == 5926600
    model.set_compute_dtype(torch.float32)
    sd0 = _sd64(model)
    spec = orc.spec_from_model(model)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 3, 224, 224, generator=g)
    y = torch.randint(0, 1000, (2,), generator=g)
    model.cuda().train()
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
        dd = p.grad.double().cpu() - work[name].grad
        num += float((dd * dd).sum()); den += float((work[name].grad ** 2).sum())
    assert (num / den) ** 0.5 < 3e-2, (num / den) ** 0.5
    # bf16, captured graph, 128-wide batch is the reference's per-GPU batch; 8 keeps the test short
    model2 = sn.Model(**kw)
    model2.apply(mb.init_weights_mnas)
    model2.cuda().train()
    opt = rmsprop.RMSprop(model2.parameters(), lr=0.004, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
    ts = engine.TrainStep(model2, opt, None, None, weight_decay=1e-5, batch_size=8, image_size=224, use_graph=True)
    ts.set_batch(torch.randn(8, 3, 224, 224, generator=g).cuda(), torch.randint(0, 1000, (8,), generator=g).cuda())
    l = []
    for _ in range(6):
        ts.step(lr=0.004)
        l.append(ts.loss[0].item())
    assert all(v == v for v in l) and l[-1] < l[0], l


def test_fold_jobs_kernel(gpu_lib):
    """atomnas_fold_jobs (csrc/reduce.hip): dst += src, src = 0 over a table of 2-D blocks; whole table and slices of it"""
    import ctypes
    from atomnas_amd import _lib, ops

    class J(ctypes.Structure):
        _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("src_ld", ctypes.c_long), ("dst_ld", ctypes.c_long),
                    ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("blk0", ctypes.c_uint), ("pad_", ctypes.c_int)]
    g = torch.Generator(device="cuda").manual_seed(3)
    src = torch.randn(40, 100, device="cuda", generator=g)
    dst = torch.randn(40, 77, device="cuda", generator=g)
    src0, dst0 = src.clone(), dst.clone()
    blocks = [(0, 3, 0, 5, 7, 13), (8, 20, 9, 30, 31, 40), (39, 99, 39, 76, 1, 1), (10, 70, 20, 0, 3, 300 // 3 // 10)]   # (sr, sc, dr, dc, rows, cols)
    arr, blk, blk0 = (J * len(blocks))(), 0, [0]
    for q, (sr, sc, dr, dc, rows, cols) in enumerate(blocks):
        arr[q] = J(src.data_ptr() + 4 * (sr * 100 + sc), dst.data_ptr() + 4 * (dr * 77 + dc), 100, 77, rows, cols, blk, 0)
        blk += (rows * cols + 255) // 256
        blk0.append(blk)
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def expect(jobs, s, d):
        for sr, sc, dr, dc, rows, cols in jobs:
            d[dr:dr + rows, dc:dc + cols] += s[sr:sr + rows, sc:sc + cols]
            s[sr:sr + rows, sc:sc + cols] = 0
    _lib.call("atomnas_fold_jobs", ctypes.c_void_p(table.data_ptr()), 1, 2, blk0[1], blk0[3] - blk0[1], st)   # jobs 1 and 2 only
    torch.cuda.synchronize()
    expect(blocks[1:3], src0, dst0)
    assert torch.equal(src, src0) and torch.equal(dst, dst0)
    _lib.call("atomnas_fold_jobs", ctypes.c_void_p(table.data_ptr()), 0, 4, 0, blk0[4], st)                    # the whole table
    torch.cuda.synchronize()
    expect(blocks, src0, dst0)
    assert torch.equal(src, src0) and torch.equal(dst, dst0)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_block_batched_weight_gradients_equal_the_per_segment_launches(gpu_lib, dtype):
    """functional._FUSED_WG_BATCH: the expand / projection weight gradients of a fused block as ONE atomnas_pw_gemm_tn per layer into
    a padded scratch matrix + atomnas_fold_jobs, against one launch per kernel-size segment -- the same products; the reduction over
    the rows is cut into chunks by the GEMM's width, so the two weight gradients agree to summation-order rounding, everything else
    bit for bit; a second backward accumulates (the scratch is left zeroed by the fold)."""
    from atomnas_amd import functional as Fn
    from atomnas_amd.models import mobilenet_base as mb

    def run(batched):
        torch.manual_seed(5)
        blk = mb.InvertedResidualChannelsFused(24, 24, 1, [30, 50, 13], [3, 5, 7], True, active_fn=mb.get_active_fn("nn.Swish"),
                                               batch_norm_kwargs={"momentum": 0.01, "eps": 1e-3}, se_ratio=0.5)
        blk.compute_dtype = dtype
        blk.cuda().train()
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn(6, 24, 14, 14, device="cuda", generator=g).requires_grad_(True)
        go = torch.randn(6, 24, 14, 14, device="cuda", generator=g)
        old, Fn._FUSED_WG_BATCH = Fn._FUSED_WG_BATCH, batched
        try:
            for _ in range(2):   # gradients accumulate over two backward passes
                blk(x).backward(go)
        finally:
            Fn._FUSED_WG_BATCH = old
        torch.cuda.synchronize()
        scratch_clean = float(blk._plan.mgr.FW.abs().max()) == 0.0
        return collections.OrderedDict((n, p.grad.detach().clone()) for n, p in blk.named_parameters()), x.grad.clone(), scratch_clean

    a, ax, _ = run(False)
    b, bx, clean = run(True)
    assert clean, "the fold must leave the scratch matrices zeroed"
    assert torch.equal(ax, bx)
    for n in a:
        if n.endswith("expand_conv.0.weight") or n.endswith("project_conv.0.weight"):
            assert torch.allclose(a[n], b[n], rtol=1e-4, atol=1e-5 * float(a[n].abs().max())), (n, float((a[n] - b[n]).abs().max()))
        else:
            assert torch.equal(a[n], b[n]), n
