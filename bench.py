#!/usr/bin/env python
"""Headline benchmark of the hot path: images/sec of the full AtomNAS-C supernet training step (forward, CE-smooth + L2 + L1,
backward, gradient all-reduce, RMSprop, EMA) at 224x224, per-GPU batch 256, bf16 activations, synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...            (N > 1 without a launcher environment: re-launches itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
Other BASELINE configurations (one JSON line each, same fields): --model atomnas_a_supernet (config 2), --model atomnas_a_supernet
--shrink 0.3 (config 3: a seeded 30 % of the atoms dead, shrink, steady state on the ragged network; shrink latency reported
separately), --model atomnas_c_plus --batch 128 (config 5: searched network with SE / Swish / fused blocks).

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

HBM_PEAK = 8.0e12    # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
MFMA_PEAK = 2.5e15   # flop/s, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA", dense
GEMM_ENTRIES = ("atomnas_pw_gemm_nt", "atomnas_pw_gemm_tn", "atomnas_expand_bwd", "atomnas_project_bwd")


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
    searched = model_name in ("atomnas_c", "atomnas_c_plus")
    if searched:   # retrain of a searched architecture (apps/searched/**): no prunable atoms, no L1 term
        from atomnas_amd.models import searched_network as sn
        model = sn.Model(**configs.searched_kwparams(model_name), input_size=hp['image_size'])
    else:
        model = ms.Model(**configs.model_kwparams(model_name), input_size=hp['image_size'])
    model.apply(mb.init_weights_mnas)
    model.set_compute_dtype(dtype)
    mp.model_profiling(model, hp['image_size'], hp['image_size'], verbose=False)
    model.cuda().train()
    pinfo = None if searched else aprune.get_bn_to_prune(model, hp['prune_params'], verbose=False)
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
    return model, ts, hp, opt, ema, pinfo


def forced_shrink(model, ts, opt, ema, pinfo, frac, seed):
    """SURVEY.md section 8(d) recipe of config 3: a seeded random `frac` of the atoms get gamma = gamma_EMA = 0 (the others are left
    alone), then the reference's shrink_model (train.py:27-81).  Returns (shrink wall time in ms, MACs before, MACs after)."""
    import train as T
    from atomnas_amd.utils import config
    g = torch.Generator().manual_seed(seed)
    table = dict(model.named_parameters())
    with torch.no_grad():
        for name in pinfo.weight:
            w = table[name]
            dead = (torch.rand(w.numel(), generator=g) < frac).to(w.device)
            w[dead] = 0.0
            ema.average(name)[dead] = 0.0

    class F(dict):
        __getattr__ = dict.__getitem__
    config.FLAGS.bind(F(image_size=224, use_distributed=False))
    wrapper = torch.nn.Module()
    wrapper.module = model
    macs0 = int(model.n_macs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    T.shrink_model(wrapper, ema, opt, pinfo, 1e-3, ema_only=False)
    ts.mgr.ensure()   # arenas rebuilt (parameters, optimizer state, EMA shadows gathered)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, macs0, int(model.n_macs)


def gemm_flops(tag_name, tag):
    """flops of one GEMM-type launch from its shape tag (2 M N K per product; the fused backward entry points do two products)"""
    f = {k: int(v) for k, v in re.findall(r"([A-Za-z]+)(\d+)", tag)}
    if tag_name == "atomnas_pw_gemm_nt":
        return 2 * f["M"] * f["N"] * f["K"]
    if tag_name == "atomnas_pw_gemm_tn":
        return 2 * f["M"] * f["NU"] * f["NV"]
    if tag_name in ("atomnas_expand_bwd", "atomnas_project_bwd"):
        return 4 * f["M"] * f["N"] * f["K"]
    return 0


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


def kernel_profile(ts, itemsize, lr, rho):
    """Eager pass of the same step with HIP events around every C-ABI launch (on the launch stream)."""
    from atomnas_amd import _lib
    was = ts.use_graph
    ts.use_graph = False
    # rank 0 profiles alone while the other ranks wait at the barrier: reduce=False keeps every collective out of this pass (the step
    # then scales its own gradients by 1, not by 1 / world; the ranks have finished the timed region, so the divergence of rank 0's
    # parameters is of no consequence for the numbers)
    ts.step(lr=lr, rho=rho, reduce=False)
    torch.cuda.synchronize()
    _lib.PROFILE = []
    ts.step(lr=lr, rho=rho, reduce=False)
    torch.cuda.synchronize()
    prof, _lib.PROFILE = _lib.PROFILE, None
    ts.use_graph = was
    agg = collections.OrderedDict()
    for name, tag, e0, e1 in prof:
        a = agg.setdefault(name, dict(launches=0, ms=0.0, bytes=0, flops=0))
        a["launches"] += 1
        a["ms"] += e0.elapsed_time(e1)
        if tag and name in ("atomnas_dwconv_fwd", "atomnas_dwconv_bwd") + GEMM_ENTRIES:
            a["bytes"] += algorithmic_bytes(name, tag, itemsize)
        if tag and name in GEMM_ENTRIES:
            a["flops"] += gemm_flops(name, tag)
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
        from atomnas_amd import build as _build
        if d.get("lib_src_sha") != _build.sources_digest():
            return None, "%s is from another build of the library (lib_src_sha differs): dropped" % os.path.relpath(files[-1], ROOT)
        return int(d["kernels"][entry]["hbm_bytes_per_launch"]), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def pmc_mfma_util():
    """MFMA utilisation per GEMM kernel family from the committed counter pass (profiles/rNN_pmc_mfma.json), same build only"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_mfma.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        from atomnas_amd import build as _build
        if d.get("lib_src_sha") != _build.sources_digest():
            return None
        return {k: round(v["mfma_util"], 4) for k, v in d["families"].items() if "mfma_util" in v}
    except Exception:
        return None


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


def relaunch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: one process per GPU under torch.distributed.run (the
    command the driver uses), same arguments; the children's rank 0 prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note("re-launching as: " + " ".join(cmd))
    raise SystemExit(subprocess.call(cmd))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks on this node (default: WORLD_SIZE of the launcher, else 1)")
    ap.add_argument("--steps", type=int, default=100)    # warm-up 20, time >= 100
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch (BASELINE metric: 256)")
    ap.add_argument("--model", default="atomnas_c_supernet",
                    choices=["atomnas_c_supernet", "atomnas_a_supernet", "mobilenet_v2_1.0", "atomnas_c", "atomnas_c_plus"])
    ap.add_argument("--shrink", type=float, default=0.0, help="config 3: fraction of atoms forced dead before a shrink; the timed steps "
                    "then run on the ragged network")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--input-pipeline", default="resident", choices=["resident", "uint8"], help="resident: the synthetic fp32 batch "
                    "stays in HBM (the headline protocol); uint8: every step takes a fresh batch from the GPU input pipeline -- decoded "
                    "uint8 images from pinned host memory, H2D + crop / PIL-exact resize / flip / normalize on a side stream "
                    "(utils/dataflow.py DevicePrefetcher), then set_batch.  Value then includes the hand-over; a different random "
                    "batch per step, so the `trained` check does not apply")
    ap.add_argument("--allow-untrained", action="store_true", help="print the line even when the cross entropy on the fixed batch did not "
                    "go down over a run of >= 100 steps (fatal otherwise)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="nccl = RCCL (the measured path); gloo only to "
                    "exercise the multi-rank code on a single-GPU box together with --same-device")
    ap.add_argument("--same-device", action="store_true", help="all ranks on cuda:0 (validation of the multi-rank path only)")
    args = ap.parse_args()

    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path runs in libatomnas_hip.so only (no CPU fallback)")
    torch.cuda.set_device(0 if args.same_device else local)
    if world > 1 or os.environ.get("ATOMNAS_FORCE_ALLREDUCE"):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend)   # "nccl" is RCCL on ROCm
        if dist.get_world_size() != args.gpus:
            raise SystemExit("bench.py: process group of %d ranks for --gpus %d" % (dist.get_world_size(), args.gpus))
    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    note("building %s" % args.model)
    model, ts, hp, opt, ema, pinfo = build(args.model, dtype, args.batch, seed=1995)
    ts.use_graph = not args.no_graph
    if world > 1:   # replicate rank 0's initialisation (reference: utils/distributed.py:183-190)
        dist.broadcast(ts.mgr.P, 0)
        dist.broadcast(ts.mgr.S, 0)
    g = torch.Generator(device="cuda").manual_seed(1995 + rank)
    x = torch.randn(args.batch, 3, hp['image_size'], hp['image_size'], device="cuda", generator=g)
    y = torch.randint(0, 1000, (args.batch,), device="cuda", generator=g)
    ts.set_batch(x, y)
    # learning rate of the timed iterations: the schedule's value at iteration 0.  The reference warms up from base_lr to
    # base_lr * batch / 256 over 5 epochs (utils/optim.py:252-306: lr(0) = 0.016 at every world size), so the first steps of a
    # large-batch job run at base_lr, not at the scaled rate.
    lr0 = hp['base_lr']
    rho = 1e-4 if pinfo is not None else 0.0

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    shrink_info = None
    if args.shrink > 0:
        if pinfo is None:
            raise SystemExit("bench.py: --shrink needs a supernet (prunable atoms)")
        note("config 3: %d warm steps, then %.0f %% of the atoms dead and shrink" % (max(args.warmup, 1), 100 * args.shrink))
        for _ in range(max(args.warmup, 1)):
            ts.step(lr=lr0, rho=rho)
        barrier()
        ms_shrink, macs0, macs1 = forced_shrink(model, ts, opt, ema, pinfo, args.shrink, seed=11)   # same seed on every rank
        shrink_info = dict(fraction=args.shrink, shrink_ms=round(ms_shrink, 1), macs_before=macs0, macs_after=macs1)
        note("shrink: %.0f ms, MACs %d -> %d" % (ms_shrink, macs0, macs1))

    pipeline = None
    if args.input_pipeline == "uint8":
        from atomnas_amd.utils import dataflow as DF
        nsteps = max(args.warmup, 1) + args.steps
        pipeline = iter(DF.DevicePrefetcher(DF.SyntheticDecodedImages(args.batch, nsteps + 1, image_size=hp['image_size'], seed=1995 + rank),
                                            image_size=hp['image_size']))

    def one_step():
        if pipeline is not None:
            xb, yb = next(pipeline)
            ts.set_batch(xb, yb)
        ts.step(lr=lr0, rho=rho)

    note("warm-up (graph capture)")
    first_loss = None
    for i in range(max(args.warmup, 1)):
        one_step()
        if i == 0:
            first_loss = float(ts.loss[0].item())
    barrier()
    note("timing %d steps" % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    barrier()
    dt = time.perf_counter() - t0
    note("timed: %.2f ms/step" % (dt / args.steps * 1e3))
    rank_ms = [dt / args.steps * 1e3]
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        every = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(every, t)
        rank_ms = [float(v.item()) / args.steps * 1e3 for v in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = ts.loss.tolist()
    topk = ts.topk.tolist()
    # the replayed graph must still be a training step: finite losses, hit counters within the batch, and a cross entropy below the
    # one of the first step on this (fixed) batch -- a step that trains on stale gradients, uncleared accumulators or a broken
    # kernel shows up here, not in the timing
    if not all(v == v and abs(v) < 1e6 for v in loss) or not all(0 <= t <= args.batch for t in topk):
        raise SystemExit("bench.py: the timed steps did not train (loss %s, top-k hits %s)" % (loss, topk))
    # A cross entropy that did not go down on the fixed batch: fatal for a run of at least 100 steps (the default protocol: with
    # that many RMSprop updates on ONE batch a working step always gets below its first loss), unless --allow-untrained says the
    # caller knows why (e.g. the first steps after a forced shrink).  Shorter runs (10..99 steps: a handful of small updates need not
    # outweigh the dropout-mask noise of the loss) are reported; `trained` is in the headline object, not buried in config.
    trained = (loss[0] < first_loss) if (args.steps + args.warmup >= 10 and pipeline is None) else None
    if trained is False:
        msg = ("the cross entropy did not go down over %d steps on a fixed batch (%.4f -> %.4f)"
               % (args.steps + max(args.warmup, 1), first_loss, loss[0]))
        if args.steps + args.warmup >= 100 and not args.allow_untrained:
            raise SystemExit("bench.py: " + msg + "; pass --allow-untrained to print the line anyway")
        note("WARNING: " + msg)

    out = None
    if rank == 0:
        ms_step = dt / args.steps * 1e3
        metric = ("images/sec (whole node) AtomNAS-C supernet 224x224 bs256/GPU" if args.model == "atomnas_c_supernet" and not args.shrink
                  else "images/sec (whole node) %s%s 224x224 bs%d/GPU" % (args.model, " after shrink" if args.shrink else "", args.batch))
        backend = dist.get_backend() if dist.is_initialized() else None
        out = collections.OrderedDict(
            metric=metric,
            value=round(args.batch * world * args.steps / dt, 1), unit="images/sec", n_gpus=world, steps=args.steps,
            warmup=args.warmup, ms_per_step=round(ms_step, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
            dtype="bf16" if dtype == torch.bfloat16 else "f32",
            data="synthetic" if pipeline is None else "synthetic uint8 images through the GPU input pipeline (H2D + crop / resize / flip / normalize per step)",
            config=dict(workload="%s full training step (fwd + CE-smooth/L2/L1 + bwd + grad all-reduce + RMSprop + EMA), 224x224" % args.model,
                        per_gpu_batch=args.batch, global_batch=args.batch * world, parallelism="dp%d" % world,
                        hip_graph=bool(ts.use_graph), lr=lr0, first_loss=round(first_loss, 4), final_loss=[round(v, 4) for v in loss],
                        ),
            trained=trained,   # cross entropy on the fixed batch below the first step's (None: fewer than 10 steps; false is fatal from 100 steps on)
            rccl_ranks=(world if backend == "nccl" else 0), comm_backend=backend, comm_mode=ts.comm_mode,
            rank_ms_per_step=[round(v, 3) for v in rank_ms])
        if shrink_info:
            out["shrink"] = shrink_info
    if not args.no_roofline and rank == 0:
        note("per-launch profile pass")
        agg = kernel_profile(ts, 2 if dtype == torch.bfloat16 else 4, lr0, rho)
        note("profile pass done")
        tot = sum(a["ms"] for a in agg.values())
        dom_name, dom = max(agg.items(), key=lambda kv: kv[1]["ms"])
        ach = dom["bytes"] / (dom["ms"] * 1e-3) if dom["ms"] > 0 else 0.0
        # the committed PMC pass was taken on the default workload (bf16, per-GPU batch 256): only valid for that
        default_wl = args.batch == 256 and dtype == torch.bfloat16 and args.model == "atomnas_c_supernet" and not args.shrink
        traffic, traffic_src = pmc_traffic(dom_name) if default_wl else (None, None)
        out["roofline"] = dict(kernel=dom_name, bound="hbm", achieved=round(ach / 1e9, 1), peak=HBM_PEAK / 1e9, unit="GB/s",
                               frac=round(ach / HBM_PEAK, 4), traffic=traffic, traffic_source=traffic_src,
                               launches_per_step=dom["launches"],
                               algorithmic_bytes_per_launch=int(dom["bytes"] / max(dom["launches"], 1)),
                               avg_launch_us=round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
                               algorithmic_bytes_per_step=dom["bytes"], kernel_ms_per_step=round(dom["ms"], 3),
                               share_of_step=round(dom["ms"] / tot, 3))
        if dom_name == "atomnas_dwconv_bwd":
            # SURVEY 8(d) charges the depthwise backward three streams (2|x| + |y|); fused with the BatchNorm backward of the next BN and
            # the activation mask of the previous one it MOVES four (g, yraw and x in, h out: xhat is needed where the ReLU mask is
            # zero), so the algorithmic fraction cannot pass 3/4 of whatever the memory system streams: 0.75 of the 8 TB/s spec, 0.50
            # of the 5.3 TB/s a 3-read-1-write probe reaches on this chip (profiles/r02_membw_probe.txt)
            out["roofline"]["four_stream"] = dict(moved_GBps=round(ach * 4 / 3 / 1e9, 1), frac_of_peak_moved=round(ach * 4 / 3 / HBM_PEAK, 4),
                                                  algorithmic_ceiling_at_spec_peak=0.75, algorithmic_ceiling_at_measured_3r1w=round(0.75 * 5.3e12 / HBM_PEAK, 3),
                                                  frac_of_measured_ceiling=round((ach * 4 / 3) / 5.3e12, 4))
        # the pointwise (1x1) layers against BOTH of their rooflines -- all GEMM-type entries of the step
        pw = [a for k, a in agg.items() if k in GEMM_ENTRIES]
        pw_ms, pw_fl, pw_by = sum(a["ms"] for a in pw), sum(a["flops"] for a in pw), sum(a["bytes"] for a in pw)
        if pw_ms > 0:
            out["pointwise"] = dict(ms_per_step=round(pw_ms, 3), flops_per_step=pw_fl, algorithmic_bytes_per_step=pw_by,
                                    tflops=round(pw_fl / (pw_ms * 1e-3) / 1e12, 1), mfma_peak_tflops=MFMA_PEAK / 1e12,
                                    frac_mfma=round(pw_fl / (pw_ms * 1e-3) / MFMA_PEAK, 4),
                                    GBps=round(pw_by / (pw_ms * 1e-3) / 1e9, 1), frac_hbm=round(pw_by / (pw_ms * 1e-3) / HBM_PEAK, 4),
                                    mfma_util=pmc_mfma_util() if default_wl else None)
        dwk = [a for k, a in agg.items() if k.startswith("atomnas_dwconv")]
        dw_ms, dw_by = sum(a["ms"] for a in dwk), sum(a["bytes"] for a in dwk)
        if dw_ms > 0:
            bwd = agg.get("atomnas_dwconv_bwd")
            moved = dw_by + (bwd["bytes"] / 3 if bwd else 0)   # the backward moves a fourth stream (see roofline.four_stream)
            out["depthwise"] = dict(ms_per_step=round(dw_ms, 3), algorithmic_bytes_per_step=dw_by,
                                    GBps=round(dw_by / (dw_ms * 1e-3) / 1e9, 1), frac_hbm=round(dw_by / (dw_ms * 1e-3) / HBM_PEAK, 4),
                                    moved_GBps=round(moved / (dw_ms * 1e-3) / 1e9, 1), frac_of_measured_streaming=round(moved / (dw_ms * 1e-3) / 5.3e12, 4))
        out["kernels"] = {k[8:]: dict(n=a["launches"], ms=round(a["ms"], 3), GBps=(round(a["bytes"] / (a["ms"] * 1e-3) / 1e9, 1) if a["bytes"] else None))
                          for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ms"])[:8]}
    if world > 1:
        dist.barrier()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_guarded(args.model if args.model in ("atomnas_c_supernet", "atomnas_a_supernet", "mobilenet_v2_1.0")
                                                    else "atomnas_c_supernet")
    if rank == 0:
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
