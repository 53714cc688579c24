"""The training iteration of the hot path as one replayable hipGraph (train.py:165-236 of the reference, per process).

    zero_grad -> forward -> CE-smooth mean -> backward -> [gradient all-reduce] -> L1 / L2 gradients + RMSprop + EMA

Everything between the barriers is device work issued from Python once, captured with torch.cuda.graph (the HIP kernels
are launched on the capturing stream through the C ABI, so they become graph nodes) and replayed every iteration.  Values
that change per iteration (lr, rho, EMA decay) are staged through a 4-float device vector, inputs through static buffers.

Data parallelism (utils/distributed.py:131-139, 155-161 of the reference: one blocking all-reduce of all gradients after
backward): the gradient arena is summed over the ranks with RCCL.  With the nccl backend the collective is issued from INSIDE
backward, bucket by bucket on a side stream as soon as the blocks that own a bucket have finished (the arena is laid out in
module order, backward runs it from the end), and is captured into the same hipGraph as the kernels ("graph" mode); the 1/world
scale rides in the optimizer kernel.  Other backends (gloo in the tests) and a failed capture use one blocking all-reduce between
two graphs ("host" mode).
"""
import os

import torch
import torch.distributed as dist

from . import ops, runtime
from .utils import optim as aopt
from .utils import prune as aprune

_DEFER_REDUCE = bool(int(os.environ.get("ATOMNAS_DEFER_REDUCE", "1")))   # experiment switch: batched weight-gradient reductions


HYP_SUMMED_RANKS = 4   # engine-owned slot of the per-step scalar vector (runtime.ArenaManager.hyper)


class TrainStep:

    def __init__(self, model, optimizer, ema=None, prune_info=None, weight_decay=1e-5, wd_method='mnas', label_smoothing=0.1,
                 batch_size=256, image_size=224, use_graph=True, process_group=None, world_size=1, allreduce_bn=False):
        self.model, self.optimizer, self.ema, self.prune_info = model, optimizer, ema, prune_info
        self.weight_decay, self.wd_method = weight_decay, wd_method
        self.use_graph = use_graph
        self.world_size = world_size
        self.pg = process_group
        # average the BN running statistics over the ranks every step, BEFORE their EMA (the reference's order: optimizer.step,
        # allreduce_bn, ema -- train.py:213-226); the optimizer does not read them, so they travel with the gradient collectives
        self.allreduce_bn = bool(allreduce_bn)
        dev = next(model.parameters()).device
        self.mgr = runtime.manager_of(model)
        self.mgr.attach_optimizer(optimizer)
        optimizer._mgr = self.mgr
        if ema is not None:
            ema.attach(self.mgr)
        self.mgr.ensure()
        self.label_smoothing = float(label_smoothing)
        self.x = torch.zeros(batch_size, 3, image_size, image_size, dtype=torch.float32, device=dev)
        self.y = torch.zeros(batch_size, dtype=torch.int64, device=dev)
        self._scal = torch.zeros(8, dtype=torch.float32, device=dev)   # per-step scalars, cleared by ONE memset per step
        self.loss = self._scal[0:3]                                    # CE (mean), L2, L1 of the last step
        self.topk = self._scal[4:6].view(torch.int32)                  # top-1 / top-5 hits of the last step (common.py:73-79)
        self.loss_vec = torch.zeros(batch_size, dtype=torch.float32, device=dev)   # per-sample CE of the last step
        self._tables_version = -1
        self._comm = None          # side stream of the bucketed all-reduce
        self._buckets = []         # (first plan of the bucket, lo, hi) in backward order
        self._fired = 0
        self.comm_mode = None      # decided at the first step: "graph" | "host" | None (no collective)
        self.g_all = None
        self.g_fwd_bwd = self.g_opt = None
        self._version = -1
        self.global_step = 0
        self._seed_grad = None
        self._agree = torch.ones(1, dtype=torch.float32, device=dev)   # capture outcome, MIN-reduced over the ranks

    # ---- pieces
    def _prune_weights(self):
        if self.prune_info is None or len(self.prune_info.weight) == 0:
            return [], []
        table = dict(self.model.named_parameters())
        return [table[n] for n in self.prune_info.weight], self.prune_info.penalty

    def _tables(self):
        """Per-arena-version device tables: weight-decay coefficient per 256-element chunk (cal_l2_loss, utils/optim.py:226-243:
        every conv / fc weight and the classifier bias for 'mnas'; dense conv and fc weights for 'slimmable') and the L1 job
        table (offset, count, penalty) of the prunable gammas (utils/prune.py:161-167)."""
        mgr = self.mgr
        if self._tables_version == mgr.version:
            return
        kinds = {'mnas': ('dense', 'dw', 'fc', 'fcbias'), 'slimmable': ('dense', 'fc')}.get(self.wd_method)
        if kinds is None:
            raise ValueError('Unknown weight_decay method: {}'.format(self.wd_method))
        wd = torch.zeros(mgr.nP // runtime.ALIGN, dtype=torch.float32)
        for kind, off, n in mgr.reg_slots:
            if kind in kinds:
                assert off % runtime.ALIGN == 0
                wd[off // runtime.ALIGN:(off + n + runtime.ALIGN - 1) // runtime.ALIGN] = self.weight_decay
        self._wd_chunk = wd.to(mgr.P.device)
        w, pen = self._prune_weights()
        self._l1 = mgr.reg_table([(p._atomnas_off, p.numel(), c) for p, c in zip(w, pen)]) if w else None
        self._ws = torch.empty(4096 + 64 * max(len(w), 1), dtype=torch.float32, device=mgr.P.device)
        self._tables_version = mgr.version

    # ---- gradient all-reduce overlapped with backward
    def _build_buckets(self, target_floats=4 << 20):
        """Cuts the gradient arena into ~16 MiB buckets at plan boundaries, from the end (backward order).  A bucket is complete
        when the plan with the LOWEST offset in it has finished its backward."""
        plans = sorted((p for p in self.mgr.plans.values() if getattr(p, "nb", 1) != 0), key=lambda p: p.g_lo)   # pruned-away blocks never run
        self._buckets = []
        hi = self.mgr.nP
        acc_lo = None
        for pl in reversed(plans):
            acc_lo = pl.g_lo
            if hi - acc_lo >= target_floats:
                self._buckets.append((pl, acc_lo, hi))
                hi = acc_lo
        if hi > 0:
            first = plans[0] if plans else None
            if self._buckets and self._buckets[-1][0] is first:
                pl, lo, bhi = self._buckets.pop()
                self._buckets.append((pl, 0, bhi))
            else:
                self._buckets.append((first, 0, hi))

    def _on_grad_done(self, pl):
        if self._fired < len(self._buckets) and self._buckets[self._fired][0] is pl:
            _, lo, hi = self._buckets[self._fired]
            self._fired += 1
            ops.reduce_flush()   # the bucket's weight gradients are complete only after their recorded reductions
            cur = torch.cuda.current_stream()
            self._comm.wait_stream(cur)
            with torch.cuda.stream(self._comm):
                dist.all_reduce(self.mgr.G[lo:hi], group=self.pg)

    def _reduce_bn(self):
        """utils/distributed.py:164-169 on the statistics arena: one collective + one launch (x 1/world from the hyper vector)"""
        mgr = self.mgr
        dist.all_reduce(mgr.S, group=self.pg)
        ops.scale_by(mgr.S, mgr.nS, mgr.hyper, ops.HYP_GRAD_SCALE)

    def _fwd_bwd_overlapped(self):
        mgr = self.mgr
        self._fired = 0
        mgr.grad_done_cb = self._on_grad_done
        try:
            self._fwd_bwd()
        finally:
            mgr.grad_done_cb = None
        cur = torch.cuda.current_stream()
        while self._fired < len(self._buckets):   # plans that took no part in this backward (none in the networks of this path)
            _, lo, hi = self._buckets[self._fired]
            self._fired += 1
            self._comm.wait_stream(cur)
            with torch.cuda.stream(self._comm):
                dist.all_reduce(mgr.G[lo:hi], group=self.pg)
        if self.allreduce_bn:
            self._comm.wait_stream(cur)
            with torch.cuda.stream(self._comm):
                self._reduce_bn()
        cur.wait_stream(self._comm)

    def _fwd_bwd(self):
        """zero_grad -> forward -> label-smoothed CE (mean) -> backward, HIP launches only.  The regularisers do not go through
        autograd here: their gradients are added in _opt (L1: one launch on the gamma job table after the all-reduce; L2: inside
        the optimizer kernel), which is the same arithmetic as the reference's `loss + l2 + l1` backward because both terms are
        identical on every rank (train.py:171-185)."""
        mgr = self.mgr
        mgr.zero_grad()
        ops.zero_(self._scal)
        loss = self.model(self.x, loss_args=(self.y, self.label_smoothing, self.loss_vec, self.topk, self.loss[0:1]))
        if self._seed_grad is None or self._seed_grad.shape != loss.shape:
            self._seed_grad = torch.ones_like(loss)   # allocated once: backward() would fill a fresh ones tensor every step
        # the fixed-order sums of the weight-gradient partials are recorded during backward and run as a few batched launches
        # (csrc/reduce.hip): ~110 graph nodes less per supernet step
        ops.reduce_defer(_DEFER_REDUCE)
        try:
            loss.backward(self._seed_grad)
        finally:
            ops.reduce_defer(False)   # flushes on the current stream

    def _opt(self):
        """[gradients summed over ranks] -> + world * rho * penalty * sign(gamma) -> RMSprop on g / world + wd * p (+ EMA of the
        parameters, L2 value) -> L1 value -> EMA of the BN statistics."""
        mgr = self.mgr
        rho_ptr = mgr.hyper[ops.HYP_RHO:ops.HYP_RHO + 1]
        group = self.optimizer.param_groups[0]
        if self._l1 is not None:
            table, njobs = self._l1
            ops.reg_value(mgr.P, table, njobs, 1, rho_ptr, 1.0, self.loss[2:3], ws=self._ws[4096:])
            ops.reg_grad(mgr.P, mgr.G, table, njobs, 1, rho_ptr, mgr.hyper[HYP_SUMMED_RANKS:HYP_SUMMED_RANKS + 1])
        ops.fused_rmsprop_ema(mgr.P, mgr.G, mgr.SQ, mgr.BUF if group['momentum'] > 0 else None,
                              mgr.EMA if self.ema is not None else None, self._wd_chunk, mgr.nP, mgr.hyper, group['alpha'],
                              group['eps'], group['eps_inside_sqrt'], group['momentum'], l2_value=self.loss[1:2], ws=self._ws)
        if self.ema is not None:
            ops.ema_update(mgr.SEMA, mgr.S, mgr.nS, mgr.hyper)
        ops.add_i64(mgr.step_counter, 1)

    def _capture(self, want_overlapped=False):
        """Captures what is missing: the one-graph form with the in-graph collectives (want_overlapped, comm mode "graph") and / or
        the forward-backward + optimizer pair that runs without a collective or around a blocking one.  Graphs that exist are kept:
        a step(reduce=False) between reducing steps must not drop the overlapped graph."""
        mgr = self.mgr
        mgr.ensure()
        self._tables()
        # warm-up outside capture (allocator pools, lazy initialisation); it must not leave a trace in the training state:
        # BN running statistics / counters are snapshotted and restored, gradients are re-zeroed by the step itself
        keep_s, keep_c = mgr.S.clone(), mgr.CNT.clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        mgr.S.copy_(keep_s)
        mgr.CNT.copy_(keep_c)
        # thread_local: the RCCL watchdog thread of a process group polls events while we capture; in the default "global" mode
        # such a call from another thread invalidates the capture
        if want_overlapped and self.comm_mode == "graph" and self.g_all is None:
            # one graph: kernels, bucketed RCCL all-reduces on the side stream, optimizer tail
            ok = True
            try:
                # one real collective on the side stream first: communicator set-up / lazy connections must not happen inside capture
                self._comm.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._comm):
                    dist.all_reduce(self._agree, group=self.pg)
                torch.cuda.current_stream().wait_stream(self._comm)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._fwd_bwd_overlapped()
                    self._opt()
                self.g_all = g
            except Exception as e:   # a runtime that cannot capture the collective: fall back to the blocking form
                import logging
                logging.warning("capturing the gradient all-reduce failed (%s); using one blocking all-reduce between two graphs", e)
                torch.cuda.synchronize()
                ok = False
            # The mode is a property of the JOB: a rank that replays N bucketed collectives next to a rank that issues one whole-arena
            # collective hangs or corrupts gradients.  Every rank reports, and one failure switches all of them to the blocking form.
            self._agree.fill_(1.0 if ok else 0.0)
            dist.all_reduce(self._agree, op=dist.ReduceOp.MIN, group=self.pg)
            if float(self._agree.item()) < 0.5:
                self.comm_mode = "host"
                self.g_all = None
        if self.g_fwd_bwd is None and not (want_overlapped and self.g_all is not None):
            self.g_fwd_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fwd_bwd, capture_error_mode="thread_local"):
                self._fwd_bwd()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, capture_error_mode="thread_local"):
                self._opt()
        self._version = mgr.version

    # ---- public
    def set_batch(self, x, y):
        self.x.copy_(x, non_blocking=True)
        self.y.copy_(y, non_blocking=True)

    def _wants_reduce(self):
        # the collective can be forced on a single rank to exercise RCCL next to graph replay on a one-GPU box
        return self.world_size > 1 or (bool(os.environ.get("ATOMNAS_FORCE_ALLREDUCE")) and dist.is_initialized())

    def step(self, lr=None, rho=0.0, ema_decay=None, reduce=None):
        """One iteration on the current static batch.  lr defaults to the optimizer's group lr.  reduce: None = all-reduce the
        gradient arena when world_size > 1; False = never (bench.py's single-rank profiling pass: the other ranks wait at a
        barrier, so no collective may be issued)."""
        mgr = self.mgr
        do_reduce = self._wants_reduce() if reduce is None else bool(reduce)
        if mgr.dirty or self._version != mgr.version:
            mgr.ensure()
            self.g_fwd_bwd = self.g_opt = self.g_all = None
            self._buckets = []
        if do_reduce and self.comm_mode is None:
            backend = dist.get_backend(self.pg) if dist.is_initialized() else None
            # ATOMNAS_OVERLAP_ALLREDUCE: "0" never, "force" with any backend (tests drive the bucketed path eagerly with gloo ranks)
            env = os.environ.get("ATOMNAS_OVERLAP_ALLREDUCE", "1")
            overlap_ok = env == "force" or (backend == "nccl" and env != "0")
            self.comm_mode = "graph" if overlap_ok else "host"
        if do_reduce and self.comm_mode == "graph" and not self._buckets:
            self._comm = self._comm or torch.cuda.Stream()
            self._build_buckets()
        h = mgr.hyper_host
        self._tables()
        h[ops.HYP_LR] = float(self.optimizer.param_groups[0]['lr'] if lr is None else lr)
        h[ops.HYP_RHO] = float(rho)
        # the scales follow what THIS step does: a step without the collective (bench.py's single-rank profile pass) trains on its own
        # gradients, not on gradients shrunk by 1 / world
        eff_world = float(max(self.world_size, 1)) if do_reduce else 1.0
        h[ops.HYP_GRAD_SCALE] = 1.0 / eff_world
        h[HYP_SUMMED_RANKS] = eff_world
        if self.ema is not None:
            h[ops.HYP_EMA_DECAY] = float(self.ema.momentum_at(self.global_step + 1) if ema_decay is None else ema_decay)
        else:
            h[ops.HYP_EMA_DECAY] = -1.0
        mgr.push_hyper()
        overlapped = do_reduce and self.comm_mode == "graph"
        if self.use_graph:
            if overlapped and self.g_all is None:
                self._capture(want_overlapped=True)   # may agree on "host" over the ranks
                overlapped = do_reduce and self.comm_mode == "graph"
            if overlapped and self.g_all is not None:
                self.g_all.replay()
            else:
                if self.g_fwd_bwd is None:
                    self._capture()
                self.g_fwd_bwd.replay()
                if do_reduce:
                    dist.all_reduce(mgr.G, group=self.pg)
                    if self.allreduce_bn:
                        self._reduce_bn()
                self.g_opt.replay()
        elif overlapped:
            self._fwd_bwd_overlapped()
            self._opt()
        else:
            self._fwd_bwd()
            if do_reduce:
                dist.all_reduce(mgr.G, group=self.pg)
                if self.allreduce_bn:
                    self._reduce_bn()
            self._opt()
        # optimizer / regulariser launches outside TrainStep (RMSprop.step(), cal_bn_l1_loss) read the same vector: leave the neutral
        # scales behind (host copy only: the device vector is re-staged by whoever launches next)
        h[ops.HYP_GRAD_SCALE] = 1.0
        h[HYP_SUMMED_RANKS] = 1.0
        self.global_step += 1
        if self.ema is not None:   # bookkeeping the reference keeps per variable (utils/optim.py:62-63); checkpointed
            self.ema.note_updates(1, float(h[ops.HYP_EMA_DECAY]))
