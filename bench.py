#!/usr/bin/env python
"""Headline benchmark of the hot path: images/sec of the full AtomNAS-C supernet training step (forward, CE-smooth + L2 + L1,
backward, gradient all-reduce, RMSprop, EMA) at 224x224, per-GPU batch 256, bf16 activations, synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement): value = whole-job images/sec, plus
  roofline     -- the dominant kernel of the step (by summed time), its algorithmic bytes / measured time vs HBM peak,
                  timed live with events on the launch stream in an eager pass of the same step;
  cpu_baseline -- the CPU oracle (oracle/atomnas_oracle.py, the verified restatement of the reference's step) timed on the
                  host cores on a bounded sample (N=1 / rank 0 only).
Weak scaling: every rank trains its own batch of 256; gradients are averaged with one RCCL all-reduce of the flat arena.
"""
import argparse
import collections
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK = 8.0e12   # B/s, MI355X_MICROARCH.md "HBM3E peak BW"


def build(model_name, dtype, batch, seed):
    from atomnas_amd import configs, engine
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import mobilenet_supernet as ms
    from atomnas_amd.utils import model_profiling as mp
    from atomnas_amd.utils import optim as aopt
    from atomnas_amd.utils import prune as aprune
    from atomnas_amd.utils import rmsprop
    hp = configs.SEARCH_HPARAMS
    torch.manual_seed(seed)
    model = ms.Model(**configs.model_kwparams(model_name), input_size=hp['image_size'])
    model.apply(mb.init_weights_mnas)
    model.set_compute_dtype(dtype)
    mp.model_profiling(model, hp['image_size'], hp['image_size'], verbose=False)
    model.cuda().train()
    pinfo = aprune.get_bn_to_prune(model, hp['prune_params'], verbose=False)
    world = dist.get_world_size() if dist.is_initialized() else 1
    lr = hp['base_lr'] * (batch * world / hp['base_total_batch'])
    opt = rmsprop.RMSprop(model.parameters(), lr=lr, alpha=hp['alpha'], momentum=hp['momentum'], eps=hp['epsilon'],
                          eps_inside_sqrt=hp['eps_inside_sqrt'])
    decay = aopt.ExponentialMovingAverage.adjust_momentum(hp['moving_average_decay'], hp['moving_average_decay_base_batch'] / (batch * world))
    ema = aopt.ExponentialMovingAverage(decay)
    for n, p in model.named_parameters():
        ema.register(n, p)
    for n, b in model.named_buffers():
        if 'running_var' in n or 'running_mean' in n:
            ema.register(n, b)
    ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=hp['weight_decay'], wd_method=hp['weight_decay_method'],
                          label_smoothing=hp['label_smoothing'], batch_size=batch, image_size=hp['image_size'],
                          world_size=world)
    return model, ts, hp


def algorithmic_bytes(tag_name, tag, itemsize):
    """Algorithmic bytes of one launch from its shape tag, so that the sums over a step are SURVEY.md section 8(d)'s figures:
    depthwise (3|X| + 2|Y|): forward reads X writes Y, backward reads X and dY, writes dX;
    pointwise (3 MK + 2 MN) per layer: forward reads A writes C (MK + MN); backward reads dC and A, writes dA (MN + 2 MK), split
    as input-gradient GEMM = dC + dA (its own MK + MN) and weight-gradient GEMM = the one extra read of A."""
    f = {k: int(v) for k, v in re.findall(r"([A-Za-z]+)(\d+)", tag)}   # "M802816 N24 K432 pro1 st1", "N256 H56 C144 k7 s1", ...
    if tag_name.startswith("atomnas_dwconv"):
        N, H, C, k, s = f["N"], f["H"], f["C"], f["k"], f["s"]
        Ho = (H - 1) // s + 1
        x, y = N * H * H * C, N * Ho * Ho * C
        return (x + y) * itemsize if tag_name.endswith("fwd") else (2 * x + y) * itemsize
    if tag_name == "atomnas_pw_gemm_nt":
        return (f["M"] * f["K"] + f["M"] * f["N"]) * itemsize
    if tag_name == "atomnas_expand_bwd":   # input-gradient GEMM (its MK + MN) + the weight gradient's one extra read of the block input
        return (f["M"] * f["K"] + 2 * f["M"] * f["N"]) * itemsize
    if tag_name == "atomnas_project_bwd":  # input-gradient GEMM (MK + MN, K = oup, N = hid) + the weight gradient's extra read of the hidden input
        return (f["M"] * f["K"] + 2 * f["M"] * f["N"]) * itemsize
    if tag_name == "atomnas_pw_gemm_tn":
        # "pro<u>,<v>": the operand with the BatchNorm-backward prologue (2) is the gradient dC; the other one is the layer input A
        m = re.search(r"pro(\d),(\d)", tag)
        a_is_u = m is not None and m.group(2) == "2"
        return f["M"] * (f["NU"] if a_is_u else f["NV"]) * itemsize
    return 0


def kernel_profile(ts, itemsize):
    """Eager pass of the same step with HIP events around every C-ABI launch (on the launch stream)."""
    from atomnas_amd import _lib
    was = ts.use_graph
    ts.use_graph = False
    # rank 0 profiles alone while the other ranks wait at the barrier: reduce=False keeps every collective out of this pass
    # (the 1/world gradient scale of the optimizer graph is harmless here)
    ts.step(rho=1e-4, reduce=False)
    torch.cuda.synchronize()
    _lib.PROFILE = []
    ts.step(rho=1e-4, reduce=False)
    torch.cuda.synchronize()
    prof, _lib.PROFILE = _lib.PROFILE, None
    ts.use_graph = was
    agg = collections.OrderedDict()
    for name, tag, e0, e1 in prof:
        a = agg.setdefault(name, dict(launches=0, ms=0.0, bytes=0))
        a["launches"] += 1
        a["ms"] += e0.elapsed_time(e1)
        if tag and name in ("atomnas_dwconv_fwd", "atomnas_dwconv_bwd", "atomnas_pw_gemm_nt", "atomnas_pw_gemm_tn", "atomnas_expand_bwd", "atomnas_project_bwd"):
            a["bytes"] += algorithmic_bytes(name, tag, itemsize)
    return agg


def pmc_traffic(entry):
    """HBM bytes per launch of `entry` from the committed PMC pass of this same command (tools/pmc_bench.sh ->
    tools/pmc_traffic.py -> profiles/rNN_pmc_traffic.json; counters need their own rocprofv3 run).  None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return int(d["kernels"][entry]["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


_T0 = time.perf_counter()


def note(msg):
    """Progress on stderr (stdout carries exactly one JSON line)."""
    sys.stderr.write("[bench %7.1fs] %s\n" % (time.perf_counter() - _T0, msg))
    sys.stderr.flush()


class _Timeout(Exception):
    pass


def cpu_baseline_guarded(model_name, limit_s=150):
    """cpu_baseline() under a wall-clock limit: a slow host must not keep the GPU numbers from being printed."""
    import signal

    def on_alarm(signum, frame):
        raise _Timeout()
    old = signal.signal(signal.SIGALRM, on_alarm)
    signal.alarm(limit_s)
    try:
        return cpu_baseline(model_name)
    except _Timeout:
        note("cpu_baseline: gave up after %d s" % limit_s)
        return dict(value=None, unit="images/sec", cores=min(os.cpu_count() or 1, 64), kind="port",
                    sample="oracle train_step did not finish its sample within %d s on this host" % limit_s)
    finally:
        signal.alarm(0)
        signal.signal(signal.SIGALRM, old)


def cpu_baseline(model_name, seconds_budget=20.0):
    """The oracle's full training step on the host cores, bs 16, same synthetic data recipe."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import atomnas_oracle as orc
    from atomnas_amd import configs
    from atomnas_amd.models import mobilenet_base as mb
    from atomnas_amd.models import mobilenet_supernet as ms
    hp = configs.SEARCH_HPARAMS
    cores = os.cpu_count() or 1
    torch.manual_seed(1995)
    model = ms.Model(**configs.model_kwparams(model_name), input_size=hp['image_size'])   # structure + init only (CPU tensors)
    model.apply(mb.init_weights_mnas)
    spec = orc.spec_from_model(model)
    sd = collections.OrderedDict((k, v.detach().clone()) for k, v in model.state_dict().items())
    names, pen, _ = orc.prune_penalties(spec, hp['image_size'])
    bs = 16
    x = torch.randn(bs, 3, hp['image_size'], hp['image_size'])
    y = torch.randint(0, 1000, (bs,))
    h = dict(lr=0.016, rho=1e-4, weight_decay=hp['weight_decay'], wd_method='mnas', label_smoothing=hp['label_smoothing'],
             alpha=hp['alpha'], eps=hp['epsilon'], momentum=hp['momentum'], ema_decay=0.9999)
    opt_state, ema = {}, collections.OrderedDict((k, v.clone()) for k, v in sd.items() if v.is_floating_point())
    # torch's CPU kernels do not scale past ~64 threads at this size (measured in round 1: 256 threads are several times slower
    # than 64 on the bench host, so no probing of larger counts inside the bench)
    threads = max(1, min(cores, 64))
    torch.set_num_threads(threads)
    note("cpu_baseline: warm-up step on %d threads" % threads)
    orc.train_step(sd, spec, opt_state, ema, x, y, h, names, pen)
    n, t0 = 0, time.perf_counter()
    while True:
        orc.train_step(sd, spec, opt_state, ema, x, y, h, names, pen)
        n += 1
        dt = time.perf_counter() - t0
        if dt > seconds_budget or n >= 10:
            break
    note("cpu_baseline: %d steps in %.1f s" % (n, dt))
    return dict(value=round(bs * n / dt, 2), unit="images/sec", cores=threads, kind="port",
                sample="oracle/atomnas_oracle.train_step (fp32 torch CPU restatement of train.py:165-236), %s, bs %d, %d timed steps after 1 warm-up, %d threads of %d host cores" % (model_name, bs, n, threads, cores))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE metric: 256)")
    ap.add_argument("--model", default="atomnas_c_supernet")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (the measured path); gloo only to "
                    "exercise the multi-rank code on a single-GPU box together with --same-device")
    ap.add_argument("--same-device", action="store_true", help="all ranks on cuda:0 (validation of the multi-rank path only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path runs in libatomnas_hip.so only (no CPU fallback)")
    torch.cuda.set_device(0 if args.same_device else local)
    if world > 1 or os.environ.get("ATOMNAS_FORCE_ALLREDUCE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)   # "nccl" is RCCL on ROCm
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    note("building %s" % args.model)
    model, ts, hp = build(args.model, dtype, args.batch, seed=1995)
    ts.use_graph = not args.no_graph
    if world > 1:   # replicate rank 0's initialisation (reference: utils/distributed.py:183-190)
        dist.broadcast(ts.mgr.P, 0)
        dist.broadcast(ts.mgr.S, 0)
    g = torch.Generator(device="cuda").manual_seed(1995 + rank)
    x = torch.randn(args.batch, 3, hp['image_size'], hp['image_size'], device="cuda", generator=g)
    y = torch.randint(0, 1000, (args.batch,), device="cuda", generator=g)
    ts.set_batch(x, y)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    note("warm-up (graph capture)")
    for _ in range(max(args.warmup, 1)):
        ts.step(rho=1e-4)
    barrier()
    note("timing %d steps" % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts.step(rho=1e-4)
    barrier()
    dt = time.perf_counter() - t0
    note("timed: %.2f ms/step" % (dt / args.steps * 1e3))
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = ts.loss.tolist()
    topk = ts.topk.tolist()
    # the replayed graph must still be a training step: finite losses, hit counters within the batch (a step that trains on stale
    # gradients or uncleared accumulators shows up here, not in the timing)
    if not all(v == v and abs(v) < 1e6 for v in loss) or not all(0 <= t <= args.batch for t in topk):
        raise SystemExit("bench.py: the timed steps did not train (loss %s, top-k hits %s)" % (loss, topk))

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        out = collections.OrderedDict(
            metric="images/sec (whole node) AtomNAS-C supernet 224x224 bs256/GPU",
            value=round(args.batch * world * args.steps / dt, 1), unit="images/sec", n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype="bf16" if dtype == torch.bfloat16 else "f32", data="synthetic",
            config=dict(workload="%s full training step (fwd + CE-smooth/L2/L1 + bwd + grad all-reduce + RMSprop + EMA), 224x224" % args.model,
                        per_gpu_batch=args.batch, global_batch=args.batch * world, parallelism="dp%d" % world,
                        hip_graph=bool(ts.use_graph), final_loss=[round(v, 4) for v in loss]))
    if not args.no_roofline and rank == 0:
        note("per-launch profile pass")
        agg = kernel_profile(ts, 2 if dtype == torch.bfloat16 else 4)
        note("profile pass done")
        tot = sum(a["ms"] for a in agg.values())
        dom_name, dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
        ach = dom["bytes"] / (dom["ms"] * 1e-3) if dom["ms"] > 0 else 0.0
        # the committed PMC pass was taken on the default workload (bf16, per-GPU batch 256): only valid for that
        traffic, traffic_src = pmc_traffic(dom_name) if (args.batch == 256 and dtype == torch.bfloat16) else (None, None)
        out["roofline"] = dict(kernel=dom_name, bound="hbm", achieved=round(ach / 1e9, 1), peak=HBM_PEAK / 1e9, unit="GB/s",
                               frac=round(ach / HBM_PEAK, 4), traffic=traffic, traffic_source=traffic_src,
                               launches_per_step=dom["launches"],
                               algorithmic_bytes_per_launch=int(dom["bytes"] / max(dom["launches"], 1)),
                               avg_launch_us=round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
                               algorithmic_bytes_per_step=dom["bytes"], kernel_ms_per_step=round(dom["ms"], 3),
                               share_of_step=round(dom["ms"] / tot, 3))
        out["kernels"] = {k[8:]: dict(n=a["launches"], ms=round(a["ms"], 3), GBps=(round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1) if a["bytes"] else None))
                          for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:8]}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_guarded(args.model)
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
