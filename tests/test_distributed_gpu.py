"""Data-parallel training step on the GPU with a process group: two ranks share cuda:0 and reduce through gloo (the box
has one GPU; RCCL refuses two ranks on one device).  What is checked is what the RCCL run relies on: the two captured graphs
with the collective on the flat gradient arena between them, the 1/world scaling inside the optimizer graph, rank-identical
parameters after the steps although every rank sees its own batch and keeps its own BatchNorm statistics."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from atomnas_amd import engine
        from atomnas_amd.models import mobilenet_base as mb
        from atomnas_amd.models import mobilenet_supernet as ms
        from atomnas_amd.utils import model_profiling as mp_
        from atomnas_amd.utils import optim as aopt
        from atomnas_amd.utils import prune as aprune
        from atomnas_amd.utils import rmsprop
        torch.manual_seed(7)   # same initialisation on every rank (bench.py broadcasts rank 0's instead)
        model = ms.Model(num_classes=10, input_size=64, input_channel=16, last_channel=64, dropout_ratio=0.0, batch_norm_momentum=0.01,
                         batch_norm_epsilon=1e-3, active_fn="nn.ReLU",
                         inverted_residual_setting=[[1, 8, 1, 1, [3]], [6, 16, 2, 2, [3, 5, 7]], [6, 24, 1, 2, [3, 5, 7]], [6, 32, 1, 2, [3, 5, 7]],
                                                    [6, 40, 1, 2, [3, 5, 7]]])
        model.apply(mb.init_weights_mnas)
        mp_.model_profiling(model, 64, 64, verbose=False)
        model.cuda().train()
        pinfo = aprune.get_bn_to_prune(model, {'bn_prune_filter': 'expansion_only_skip_expand1'}, verbose=False)
        opt = rmsprop.RMSprop(model.parameters(), lr=0.01, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
        ema = aopt.ExponentialMovingAverage(0.99)
        for n, p in model.named_parameters():
            ema.register(n, p)
        for n, b in model.named_buffers():
            if "running" in n:
                ema.register(n, b)
        ts = engine.TrainStep(model, opt, ema, pinfo, batch_size=8, image_size=64, use_graph=True, world_size=world)
        g = torch.Generator().manual_seed(100 + rank)   # every rank its own batch
        ts.set_batch(torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 10, (8,), generator=g).cuda())
        p0 = ts.mgr.P.clone()
        losses = []
        for _ in range(3):
            ts.step(lr=0.003, rho=1e-4)
            losses.append(float(ts.loss[0]))
        torch.cuda.synchronize()
        assert all(l == l for l in losses)
        assert float((ts.mgr.P - p0).abs().max()) > 0, "parameters did not move"
        assert ts.comm_mode == "host"   # gloo: one blocking all-reduce between the two graphs
        # the reduced gradient arena is the SUM of the per-rank gradients (the 1/world scale rides in the optimizer kernel)
        ts._fwd_bwd()
        torch.cuda.synchronize()
        local = ts.mgr.G.detach().cpu().clone()
        every = [torch.zeros_like(local) for _ in range(world)]
        dist.all_gather(every, local)
        dist.all_reduce(ts.mgr.G)
        torch.cuda.synchronize()
        assert not torch.equal(every[0], every[1])
        assert torch.equal(ts.mgr.G.cpu(), every[0] + every[1]), float((ts.mgr.G.cpu() - every[0] - every[1]).abs().max())
        mine = ts.mgr.P.detach().cpu()
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        assert all(torch.equal(q, parts[0]) for q in parts), "ranks diverged: max diff %g" % float((parts[0] - parts[1]).abs().max())
        lo = torch.tensor(losses)
        both = [torch.zeros_like(lo) for _ in range(world)]
        dist.all_gather(both, lo)
        assert not torch.equal(both[0], both[1]), "ranks were supposed to see different batches"
        # allreduce_bn (utils/distributed.py:164-169, train.py:213-226): the running statistics are averaged over the ranks
        # BEFORE their EMA, so the shadows of the statistics are rank-identical too
        from atomnas_amd import ops

        def gathered(t):
            mine_ = t.detach().cpu().clone()
            allr = [torch.zeros_like(mine_) for _ in range(world)]
            dist.all_gather(allr, mine_)
            return allr
        s_before = gathered(ts.mgr.S)
        assert not torch.equal(s_before[0], s_before[1]), "per-rank BN statistics expected before allreduce_bn"
        ts.allreduce_bn = True
        sema0 = ts.mgr.SEMA.detach().clone()
        ts.step(lr=0.003, rho=1e-4)
        torch.cuda.synchronize()
        s_after = gathered(ts.mgr.S)
        assert torch.equal(s_after[0], s_after[1]), "allreduce_bn left rank-local statistics"
        d = torch.tensor(float(ts.mgr.hyper_host[ops.HYP_EMA_DECAY]), dtype=torch.float32)
        want = sema0.cpu() * d + (1.0 - d) * s_after[0]
        assert torch.allclose(ts.mgr.SEMA.cpu(), want, rtol=1e-6, atol=1e-7), float((ts.mgr.SEMA.cpu() - want).abs().max())
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_two_ranks_one_gpu_graph_step(gpu_lib):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(out.keys()) == list(range(world))


def _nccl_worker(port, out):
    """One rank, RCCL backend, collective forced on: the bucketed all-reduce issued from inside backward on the side stream is
    captured into the step's hipGraph ("graph" mode).  With a single rank the sum is the identity, so the run must be bit-identical
    to the same steps without any collective."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ATOMNAS_FORCE_ALLREDUCE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        from trainstep_diag import setup
        res = []
        for force in (True, False):
            model, sd, spec, pinfo, opt, ema, engine = setup(torch.bfloat16, 64)
            ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-3, batch_size=8, image_size=64, use_graph=True)
            g = torch.Generator().manual_seed(5)
            for step in range(3):
                ts.set_batch(torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 10, (8,), generator=g).cuda())
                ts.step(lr=0.002, rho=1e-3, reduce=force)
            torch.cuda.synchronize()
            if force:
                assert ts.comm_mode == "graph" and ts.g_all is not None, (ts.comm_mode, ts.g_all)
                assert len(ts._buckets) >= 1 and ts._fired == len(ts._buckets)
                lo = sorted(b[1] for b in ts._buckets)
                assert lo[0] == 0 and sum(b[2] - b[1] for b in ts._buckets) == ts.mgr.nP   # the buckets tile the arena
            res.append((ts.mgr.P.clone(), ts.mgr.SQ.clone(), ts.loss.clone()))
        for a, b in zip(*res):
            assert torch.equal(a, b)
        out["ok"] = 1
    finally:
        dist.destroy_process_group()


def test_rccl_allreduce_is_captured_in_the_step_graph(gpu_lib):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), out))
    p.start()
    p.join(timeout=300)
    if p.is_alive():
        p.terminate()
    assert p.exitcode == 0 and out.get("ok") == 1


def _bucket_worker(rank, world, port, out):
    """Two gloo ranks on one device drive the BUCKETED all-reduce of engine.TrainStep eagerly (ATOMNAS_OVERLAP_ALLREDUCE=force lifts
    the backend gate; gloo collectives cannot be captured, so use_graph is off): the collectives are issued from inside backward,
    bucket by bucket on the side stream.  Against one blocking all-reduce of the whole arena over the same per-rank gradients the
    result must be bit-identical, the buckets must tile the arena, and every bucket must have fired."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.environ.update(RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      ATOMNAS_OVERLAP_ALLREDUCE="force")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from trainstep_diag import setup
        model, sd, spec, pinfo, opt, ema, engine = setup(torch.bfloat16, 64)   # same seed -> same initialisation on both ranks
        ts = engine.TrainStep(model, opt, ema, pinfo, weight_decay=1e-3, batch_size=8, image_size=64, use_graph=False, world_size=world)
        g = torch.Generator().manual_seed(300 + rank)
        ts.set_batch(torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 10, (8,), generator=g).cuda())
        # reference: the per-rank gradients of this state, summed by ONE all-reduce of the whole arena
        ts._fwd_bwd()
        torch.cuda.synchronize()
        ref = ts.mgr.G.detach().clone()
        dist.all_reduce(ref)
        # the bucketed form on the same state (BN statistics are per rank and do not enter the gradients of a train-mode step)
        ts.step(lr=0.0, rho=0.0)   # lr 0: parameters stay where they are; comm mode is decided here
        assert ts.comm_mode == "graph", ts.comm_mode
        assert len(ts._buckets) >= 1 and ts._fired == len(ts._buckets)
        lo = sorted(b[1] for b in ts._buckets)
        assert lo[0] == 0 and sum(b[2] - b[1] for b in ts._buckets) == ts.mgr.nP
        edges = sorted((b[1], b[2]) for b in ts._buckets)
        assert all(edges[i][1] == edges[i + 1][0] for i in range(len(edges) - 1)), edges   # no gap, no overlap
        ts._fwd_bwd_overlapped()
        torch.cuda.synchronize()
        assert ts._fired == len(ts._buckets)
        assert torch.equal(ts.mgr.G, ref), float((ts.mgr.G - ref).abs().max())
        # small buckets: several collectives per backward, same result
        ts._build_buckets(target_floats=1 << 12)
        assert len(ts._buckets) > 2
        ts._fwd_bwd_overlapped()
        torch.cuda.synchronize()
        assert ts._fired == len(ts._buckets)
        assert torch.equal(ts.mgr.G, ref), float((ts.mgr.G - ref).abs().max())
        # a step without the collective between reducing steps scales its own gradient by 1, not by 1 / world
        from atomnas_amd import ops
        p0 = ts.mgr.P.clone()
        ts.step(lr=0.001, rho=0.0, reduce=False)
        torch.cuda.synchronize()
        assert float(ts.mgr.hyper[ops.HYP_GRAD_SCALE]) == 1.0 and float(ts.mgr.hyper[4]) == 1.0
        assert float((ts.mgr.P - p0).abs().max()) > 0
        ts.step(lr=0.001, rho=0.0)
        torch.cuda.synchronize()
        assert float(ts.mgr.hyper[ops.HYP_GRAD_SCALE]) == 0.5 and float(ts.mgr.hyper[4]) == 2.0
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_two_ranks_matches_one_shot(gpu_lib):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(out.keys()) == list(range(world))


def test_bench_gpus_2_launches_itself(gpu_lib):
    """`python bench.py --gpus 2` without a launcher environment re-launches itself under torch.distributed.run, runs the full-size
    supernet step on two ranks (both on cuda:0, gloo: the box has one GPU), the rank-0 profile pass included, and prints ONE JSON
    line whose rank count matches --gpus."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--batch", "8",
                        "--steps", "2", "--warmup", "2", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 16 and j["config"]["parallelism"] == "dp2"
    assert len(j["rank_ms_per_step"]) == 2 and j["comm_backend"] == "gloo" and j["comm_mode"] == "host" and j["rccl_ranks"] == 0
    assert j["value"] > 0 and "roofline" in j and "pointwise" in j and j["pointwise"]["frac_mfma"] > 0
    # a launcher / flag mismatch is an error, not a mislabelled 1-GPU run
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "8", "--steps", "1"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "must agree" in (bad.stdout + bad.stderr)


# --------------------------------------------------------------------------------------------------- real RCCL, >= 2 GPUs
def _rccl_worker(rank, world, port, out):
    """One rank per GPU, backend "nccl" (RCCL over xGMI): the bucketed all-reduce issued from inside backward is captured into the
    step graph; the reduced arena is the one-shot all-reduce bit for bit; parameters stay rank-identical over three steps."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world)
    try:
        from atomnas_amd import engine
        from atomnas_amd.models import mobilenet_base as mb
        from atomnas_amd.models import mobilenet_supernet as ms
        from atomnas_amd.utils import model_profiling as mp_
        from atomnas_amd.utils import optim as aopt
        from atomnas_amd.utils import prune as aprune
        from atomnas_amd.utils import rmsprop
        torch.manual_seed(7)
        model = ms.Model(num_classes=10, input_size=64, input_channel=16, last_channel=64, dropout_ratio=0.0, batch_norm_momentum=0.01,
                         batch_norm_epsilon=1e-3, active_fn="nn.ReLU",
                         inverted_residual_setting=[[1, 8, 1, 1, [3]], [6, 16, 2, 2, [3, 5, 7]], [6, 24, 1, 2, [3, 5, 7]], [6, 32, 1, 2, [3, 5, 7]],
                                                    [6, 40, 1, 2, [3, 5, 7]]])
        model.apply(mb.init_weights_mnas)
        mp_.model_profiling(model, 64, 64, verbose=False)
        model.cuda().train()
        pinfo = aprune.get_bn_to_prune(model, {'bn_prune_filter': 'expansion_only_skip_expand1'}, verbose=False)
        opt = rmsprop.RMSprop(model.parameters(), lr=0.01, alpha=0.9, momentum=0.9, eps=1e-3, eps_inside_sqrt=True)
        ema = aopt.ExponentialMovingAverage(0.99)
        for n, p in model.named_parameters():
            ema.register(n, p)
        for n, b in model.named_buffers():
            if "running" in n:
                ema.register(n, b)
        ts = engine.TrainStep(model, opt, ema, pinfo, batch_size=8, image_size=64, use_graph=True, world_size=world)
        g = torch.Generator().manual_seed(100 + rank)
        ts.set_batch(torch.randn(8, 3, 64, 64, generator=g).cuda(), torch.randint(0, 10, (8,), generator=g).cuda())
        for _ in range(3):
            ts.step(lr=0.003, rho=1e-4)
        torch.cuda.synchronize()
        assert ts.comm_mode == "graph", ts.comm_mode      # (i) the collectives were captured into the step graph
        # (ii) the bucketed reduction issued from inside backward == ONE all-reduce of the whole arena over the same per-rank gradients
        ts._fwd_bwd()
        torch.cuda.synchronize()
        ref = ts.mgr.G.detach().clone()
        dist.all_reduce(ref)
        ts._fwd_bwd_overlapped()
        torch.cuda.synchronize()
        assert len(ts._buckets) >= 1 and ts._fired == len(ts._buckets)
        assert torch.equal(ts.mgr.G, ref), float((ts.mgr.G - ref).abs().max())
        # (iii) rank-identical parameters although every rank saw its own batch
        mine = ts.mgr.P.detach().clone()
        parts = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine)
        assert all(torch.equal(q, parts[0]) for q in parts), "ranks diverged: max diff %g" % float((parts[0] - parts[-1]).abs().max())
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (real RCCL between devices); the 1-GPU box skips")
def test_rccl_two_devices_graph_step(gpu_lib):
    world = 2
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    for p in procs:
        if p.is_alive():
            p.terminate()
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(out.keys()) == list(range(world))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (real RCCL between devices); the 1-GPU box skips")
def test_bench_gpus_2_rccl(gpu_lib):
    """`python bench.py --gpus 2` as the driver's scaling run starts it (one rank per GPU, RCCL): rccl_ranks == 2, graph-mode collectives."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "32", "--steps", "3", "--warmup", "2",
                        "--no-cpu-baseline", "--no-roofline"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["comm_backend"] == "nccl" and j["comm_mode"] == "graph", j
    assert len(j["rank_ms_per_step"]) == 2 and j["value"] > 0
