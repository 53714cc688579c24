"""Device-resident flat storage ("arenas") behind the reference's nn.Module / optimizer / EMA objects.

The reference keeps 497 separate parameter tensors, 302 BN buffers, 2x497 RMSprop state tensors and 799 EMA shadows and
walks them in Python loops every step (utils/optim.py:226-243, utils/rmsprop.py:70-132, utils/optim.py:54-65,
utils/distributed.py:131-139).  Here every one of those tensors is a *view* into a handful of flat fp32 arenas:

    P    parameters            G    gradients (p.grad)       SQ / BUF   RMSprop square_avg / momentum_buffer
    EMA  EMA shadows of P      S    BN running statistics    SEMA       EMA shadows of S         CNT  int64 BN counters

so that the optimizer tail, gradient all-reduce, broadcast and regularisers are O(1) launches on one pointer, while
`state_dict()` names, shapes and `nn.Parameter` identities stay those of the reference.

Inside an atomic block the three branches are laid out *fused*: expand weights of all branches form one [HT, inp] matrix,
projection weights one [oup, HT] matrix, BN vectors one [HT] vector, where each branch owns a segment whose length is
rounded up to 16 channels, one slab of the slab-major hidden tensors (HT = sum of padded segments).  Padding entries are zeros that belong to no Parameter; they
stay zero under training (their gradients are identically zero) and let every kernel use aligned 16-byte accesses after a
shrink has produced ragged channel counts (13, 139, ...).
"""
import collections
import ctypes

import torch
from torch import nn

from . import ops

ALIGN = 256  # arena slots start on 256-element (1 KiB) boundaries


def pad8(c):
    return (c + 7) // 8 * 8


def pads(c):
    """width of a branch segment inside a block's hidden tensor: whole 16-channel slabs (slab-major layout, ops.Slab)"""
    return (c + 15) // 16 * 16


def pad32(c):
    return (c + 31) // 32 * 32


def pad64(c):
    return (c + 63) // 64 * 64


def _align(n, a=ALIGN):
    return (n + a - 1) // a * a


class _Layout:
    """Assigns offsets inside one arena."""

    def __init__(self):
        self.size = 0

    def take(self, numel):
        off = self.size
        self.size = _align(off + max(int(numel), 1))
        return off


class PwPack:
    """Packed 1x1 weight in both orientations (see atomnas_pack_weights)."""
    __slots__ = ("w", "wt", "n", "k")


class BlockPlan:
    """Everything the executor needs for one InvertedResidualChannels block: views into arenas and pack buffers."""

    def __init__(self):
        self.valid = False

    def __deepcopy__(self, memo):
        return None   # plans point into the owner's arenas: a copied module gets its own at its first use (plan_of)


class SimplePlan:
    def __deepcopy__(self, memo):
        return None


class ArenaManager:
    """Owns the arenas of one model (attached as model._arena)."""

    def __deepcopy__(self, memo):
        return None   # copy.deepcopy(model) must not drag the arenas, optimizers and EMA of the source along (common.get_ema_model)

    def __init__(self, model):
        self.model = model
        self.dirty = True
        self.optimizers = []
        self.emas = []
        self.P = self.G = self.SQ = self.BUF = self.EMA = self.S = self.SEMA = self.CNT = None
        self._stats_ws, self._stats_off, self._stats_need = None, 0, 0
        self.version = 0
        self.param_slots = collections.OrderedDict()  # name -> (off, shape, strides) of every nn.Parameter
        self.compute_dtype = getattr(model, "compute_dtype", torch.bfloat16)

    # ------------------------------------------------------------------ registration
    def attach_optimizer(self, opt):
        if not any(o is opt for o in self.optimizers):
            self.optimizers.append(opt)
            self.dirty = True

    def attach_ema(self, ema):
        if not any(e is ema for e in self.emas):
            self.emas.append(ema)
            self.dirty = True

    def mark_dirty(self):
        self.dirty = True

    def ensure(self):
        if self.dirty:
            self.materialize()

    @property
    def device(self):
        return next(self.model.parameters()).device

    # ------------------------------------------------------------------ layout
    def materialize(self):
        """(Re)builds all arenas from the current module tree, migrating values, optimizer state and EMA shadows."""
        from .models import mobilenet_base as mb  # local import: avoids a cycle

        model = self.model
        dev = self.device
        if dev.type != "cuda":
            raise ops._lib.AtomnasHipError("the AtomNAS arenas live in HBM: move the model to the GPU first (model.cuda())")
        self.compute_dtype = getattr(model, "compute_dtype", self.compute_dtype)
        T = self.compute_dtype

        lp, ls, lc = _Layout(), _Layout(), _Layout()   # params / bn stats / counters
        lpk, lpf = _Layout(), _Layout()                # packed weights (T) / packed depthwise taps (fp32)
        binds = []       # (tensor_getter, arena_name, off, shape, strides, setter)
        reg_slots = []   # (kind, offset, numel) of weight slots for the L2 regulariser: dense / dw / fc / fcbias
        pack_jobs_t, pack_jobs_f = [], []
        plans = []

        def bind_param(mod, attr, off, shape, strides=None):
            binds.append(("P", mod, attr, off, tuple(shape), strides))

        def bind_buf(mod, attr, arena, off, shape):
            binds.append((arena, mod, attr, off, tuple(shape), None))

        def pw_pack(src_off, rows, cols, src_ld):
            """registers pack jobs for W[rows][cols] and its transpose; returns offsets/shapes in the T pack buffer"""
            ldw, ldt = pad32(cols), pad32(rows)
            o_w = lpk.take(pad64(rows) * ldw)
            o_t = lpk.take(pad64(cols) * ldt)
            pack_jobs_t.append((src_off, o_w, rows, cols, src_ld, ldw, 0, 0))
            pack_jobs_t.append((src_off, o_t, rows, cols, src_ld, ldt, 0, 1))
            return (o_w, pad64(rows), ldw), (o_t, pad64(cols), ldt)

        def bn_slots(C):
            return dict(g=lp.take(C), b=lp.take(C), rm=ls.take(C), rv=ls.take(C))

        # ---- walk the module tree
        handled = set()
        for name, m in model.named_modules():
            if isinstance(m, mb.InvertedResidualChannels):
                pl = BlockPlan()
                pl.g_lo = lp.size
                pl.name = name
                pl.module = m
                nb = len(m.ops)
                pl.inp, pl.oup, pl.stride, pl.expand = m.input_dim, m.output_dim, m.stride, m.expand
                pl.res = m.use_res_connect
                pl.ks = list(m.kernel_sizes)
                pl.hid = list(m.channels)
                pl.nb = nb
                for sub in m.modules():
                    handled.add(id(sub))
                if nb == 0:
                    pl.HT = 0
                    # an all-pruned block keeps only its (unused) pw_bn; give it ordinary slots
                    sl = bn_slots(pl.oup)
                    self._bind_bn(bind_param, bind_buf, lc, m.pw_bn, sl, 0, pl.oup)
                    pl.g_hi = lp.size; plans.append((m, pl, {}))
                    continue
                # expanding blocks keep their hidden tensors slab-major (segments = whole 16-channel slabs); the first block of
                # the network (no expansion: hidden = input, narrow) stays plain with 8-channel padding
                sp = pl.segpad = pads if m.expand else pad8
                pl.seg = []
                o = 0
                for h in pl.hid:
                    pl.seg.append(o)
                    o += sp(h)
                HT = pl.HT = o
                so = {}
                if m.expand:
                    so["We"] = lp.take(HT * pl.inp)
                    reg_slots.append(("dense", so["We"], HT * pl.inp))
                    so["bne"] = bn_slots(HT)
                so["Wd"] = [lp.take(sp(h) * k * k) for h, k in zip(pl.hid, pl.ks)]
                for o_, h_, k_ in zip(so["Wd"], pl.hid, pl.ks):
                    reg_slots.append(("dw", o_, sp(h_) * k_ * k_))
                so["bnd"] = bn_slots(HT)
                so["Wp"] = lp.take(pl.oup * HT)
                reg_slots.append(("dense", so["Wp"], pl.oup * HT))
                so["bnp"] = bn_slots(pl.oup)
                idx_depth = 1 if m.expand else 0
                for i, op in enumerate(m.ops):
                    ch = list(op.children())
                    s, h, k = pl.seg[i], pl.hid[i], pl.ks[i]
                    if m.expand:
                        conv, bn, _ = list(ch[0].children())
                        bind_param(conv, "weight", so["We"] + s * pl.inp, (h, pl.inp, 1, 1))
                        self._bind_bn(bind_param, bind_buf, lc, bn, so["bne"], s, h)
                    conv, bn, _ = list(ch[idx_depth].children())
                    bind_param(conv, "weight", so["Wd"][i], (h, 1, k, k))
                    self._bind_bn(bind_param, bind_buf, lc, bn, so["bnd"], s, h)
                    proj = ch[idx_depth + 1]
                    bind_param(proj, "weight", so["Wp"] + s, (pl.oup, h, 1, 1), (HT, 1, 1, 1))
                self._bind_bn(bind_param, bind_buf, lc, m.pw_bn, so["bnp"], 0, pl.oup)
                # pack buffers
                pk = {}
                if m.expand:
                    pk["We"], pk["WeT"] = pw_pack(so["We"], HT, pl.inp, pl.inp)
                pk["Wp"], pk["WpT"] = pw_pack(so["Wp"], pl.oup, HT, HT)
                pk["taps"] = []
                for i, (h, k) in enumerate(zip(pl.hid, pl.ks)):
                    o_f = lpf.take(k * k * sp(h))
                    pack_jobs_f.append((so["Wd"][i], o_f, sp(h), k * k, k * k, sp(h), 0, 2))
                    pk["taps"].append((o_f, k * k, sp(h)))
                pl.g_hi = lp.size; plans.append((m, pl, dict(so=so, pk=pk)))
            elif isinstance(m, mb.InvertedResidualChannelsFused):
                # Fused block (models/mobilenet_base.py:145-274): ONE expand conv / BN over all `total` hidden channels, Narrow +
                # depthwise per kernel size, optional SE, ONE projection conv + BN.  The kernels run on the same padded-segment
                # layout as the branch block (HT = sum of segments padded to whole slabs); the total-wide parameters (expand
                # weight / BN, projection weight, SE weights) are CONTIGUOUS masters in the arena, as the reference's state_dict
                # has them, and reach the padded layout through per-segment pack jobs / per-segment finalize launches.
                pl = BlockPlan()
                pl.g_lo = lp.size
                pl.name, pl.module, pl.fused = name, m, True
                pl.inp, pl.oup, pl.stride, pl.expand = m.input_dim, m.output_dim, m.stride, m.expand
                pl.res = m.use_res_connect
                pl.ks, pl.hid = list(m.kernel_sizes), list(m.channels)
                nb = pl.nb = len(m.depth_ops)
                for sub in m.modules():
                    handled.add(id(sub))
                sp = pl.segpad = pads if m.expand else pad8
                pl.seg, pl.start = [], []
                o = st = 0
                for h in pl.hid:
                    pl.seg.append(o)
                    pl.start.append(st)
                    o += sp(h)
                    st += h
                HT, total = o, st
                pl.HT, pl.total = HT, total
                so, pk = {}, {}
                ldw, ldt = pad32(pl.inp), pad32(HT)
                if m.expand:
                    conv, bn, _ = list(m.expand_conv.children())
                    so["We"] = lp.take(total * pl.inp)
                    reg_slots.append(("dense", so["We"], total * pl.inp))
                    bind_param(conv, "weight", so["We"], (total, pl.inp, 1, 1))
                    so["bne"] = bn_slots(total)
                    self._bind_bn(bind_param, bind_buf, lc, bn, so["bne"], 0, total)
                    o_w, o_t = lpk.take(pad64(HT) * ldw), lpk.take(pad64(pl.inp) * ldt)
                    for sg, stt, h in zip(pl.seg, pl.start, pl.hid):
                        pack_jobs_t.append((so["We"] + stt * pl.inp, o_w + sg * ldw, h, pl.inp, pl.inp, ldw, 0, 0))
                        pack_jobs_t.append((so["We"] + stt * pl.inp, o_t + sg, h, pl.inp, pl.inp, ldt, 0, 1))
                    pk["We"], pk["WeT"] = (o_w, pad64(HT), ldw), (o_t, pad64(pl.inp), ldt)
                so["Wd"] = [lp.take(sp(h) * k * k) for h, k in zip(pl.hid, pl.ks)]
                so["bnd"] = bn_slots(HT)
                pk["taps"] = []
                idx = 1 if m.expand else 0
                for i, op in enumerate(m.depth_ops):
                    conv, bn, _ = list(list(op.children())[idx].children())
                    h, k = pl.hid[i], pl.ks[i]
                    reg_slots.append(("dw", so["Wd"][i], sp(h) * k * k))
                    bind_param(conv, "weight", so["Wd"][i], (h, 1, k, k))
                    self._bind_bn(bind_param, bind_buf, lc, bn, so["bnd"], pl.seg[i], h)
                    o_f = lpf.take(k * k * sp(h))
                    pack_jobs_f.append((so["Wd"][i], o_f, sp(h), k * k, k * k, sp(h), 0, 2))
                    pk["taps"].append((o_f, k * k, sp(h)))
                pconv, pbn = list(m.project_conv.children())
                so["Wp"] = lp.take(pl.oup * total)
                reg_slots.append(("dense", so["Wp"], pl.oup * total))
                bind_param(pconv, "weight", so["Wp"], (pl.oup, total, 1, 1))
                so["bnp"] = bn_slots(pl.oup)
                self._bind_bn(bind_param, bind_buf, lc, pbn, so["bnp"], 0, pl.oup)
                ldp, ldpt = pad32(HT), pad32(pl.oup)
                o_w, o_t = lpk.take(pad64(pl.oup) * ldp), lpk.take(pad64(HT) * ldpt)
                for sg, stt, h in zip(pl.seg, pl.start, pl.hid):
                    pack_jobs_t.append((so["Wp"] + stt, o_w, pl.oup, h, total, ldp, sg, 0))
                    pack_jobs_t.append((so["Wp"] + stt, o_t, pl.oup, h, total, ldpt, sg, 1))
                pk["Wp"], pk["WpT"] = (o_w, pad64(pl.oup), ldp), (o_t, pad64(HT), ldpt)
                pl.se = isinstance(m.se_op, mb.SqueezeAndExcitation)
                if pl.se:
                    se = m.se_op
                    pl.se_hid = se.n_hidden
                    for conv_, key in ((se.se_reduce, "se1"), (se.se_expand, "se2")):
                        so[key + "w"] = lp.take(conv_.weight.numel())
                        so[key + "b"] = lp.take(conv_.bias.numel())
                        reg_slots.append(("dense", so[key + "w"], conv_.weight.numel()))
                        bind_param(conv_, "weight", so[key + "w"], tuple(conv_.weight.shape))
                        bind_param(conv_, "bias", so[key + "b"], tuple(conv_.bias.shape))
                    # fp32 copies of the gate's dense layers over the padded channel layout, both with the channels contiguous:
                    # W1p[j][sg + c] = W1[j][st + c], W2t[j][sg + c] = W2[st + c][j], b2p[sg + c] = b2[st + c] (csrc/se.hip k_se_mlp)
                    o1, o2, o3 = lpf.take(pl.se_hid * HT), lpf.take(pl.se_hid * HT), lpf.take(HT)
                    for sg, stt, h in zip(pl.seg, pl.start, pl.hid):
                        pack_jobs_f.append((so["se1w"] + stt, o1, pl.se_hid, h, total, HT, sg, 0))
                        pack_jobs_f.append((so["se2w"] + stt * pl.se_hid, o2, h, pl.se_hid, pl.se_hid, HT, sg, 2))
                        pack_jobs_f.append((so["se2b"] + stt, o3, 1, h, total, HT, sg, 0))
                    pk["se"] = (o1, o2, o3)
                pl.g_hi = lp.size; plans.append((m, pl, dict(so=so, pk=pk)))
            elif isinstance(m, mb.ConvBNReLU) and id(m) not in handled:
                conv, bn, _ = list(m.children())
                for sub in m.modules():
                    handled.add(id(sub))
                pl = SimplePlan()
                pl.g_lo = lp.size
                pl.name, pl.module = name, m
                pl.cin, pl.cout, pl.k, pl.stride, pl.groups = conv.in_channels, conv.out_channels, conv.kernel_size[0], conv.stride[0], conv.groups
                kk = conv.kernel_size[0] * conv.kernel_size[1]
                cols = (conv.in_channels // conv.groups) * kk
                so = dict(W=lp.take(pad8(conv.out_channels) * cols), bn=bn_slots(pad8(conv.out_channels)))
                reg_slots.append(("dense" if conv.groups == 1 else "dw", so["W"], conv.out_channels * cols))
                bind_param(conv, "weight", so["W"], tuple(conv.weight.shape))
                self._bind_bn(bind_param, bind_buf, lc, bn, so["bn"], 0, conv.out_channels)
                pk = {}
                if conv.groups == 1:
                    pk["W"], pk["WT"] = pw_pack(so["W"], conv.out_channels, cols, cols)
                elif conv.groups == conv.in_channels == conv.out_channels:
                    o_f = lpf.take(kk * pad8(conv.out_channels))
                    pack_jobs_f.append((so["W"], o_f, pad8(conv.out_channels), kk, kk, pad8(conv.out_channels), 0, 2))
                    pk["taps"] = (o_f, kk, pad8(conv.out_channels))
                else:
                    raise NotImplementedError("grouped convolution other than depthwise")
                pl.g_hi = lp.size; plans.append((m, pl, dict(so=so, pk=pk)))
            elif isinstance(m, nn.Linear) and id(m) not in handled:
                handled.add(id(m))
                pl = SimplePlan()
                pl.g_lo = lp.size
                pl.name, pl.module = name, m
                pl.cin, pl.cout = m.in_features, m.out_features
                so = dict(W=lp.take(m.out_features * m.in_features), b=lp.take(pad8(m.out_features)))
                reg_slots.append(("fc", so["W"], m.out_features * m.in_features))
                if m.bias is not None:
                    reg_slots.append(("fcbias", so["b"], m.out_features))
                bind_param(m, "weight", so["W"], (m.out_features, m.in_features))
                if m.bias is not None:
                    bind_param(m, "bias", so["b"], (m.out_features,))
                pk = {}
                pk["W"], pk["WT"] = pw_pack(so["W"], m.out_features, m.in_features, m.in_features)
                pl.g_hi = lp.size; plans.append((m, pl, dict(so=so, pk=pk)))
        # anything else that owns parameters (e.g. SE convs with bias) gets plain slots
        for name, m in model.named_modules():
            if id(m) in handled:
                continue
            for attr, p in list(m._parameters.items()):
                if p is None:
                    continue
                bind_param(m, attr, lp.take(p.numel()), tuple(p.shape))
            if isinstance(m, nn.BatchNorm2d):
                raise NotImplementedError("BatchNorm2d outside ConvBNReLU / InvertedResidualChannels: %s" % name)

        # ---- allocate
        nP, nS, nC = max(lp.size, ALIGN), max(ls.size, ALIGN), max(lc.size, 8)
        f32 = dict(dtype=torch.float32, device=dev)
        newP = torch.zeros(nP, **f32)
        newG = torch.zeros(nP, **f32)
        newS = torch.zeros(nS, **f32)
        newC = torch.zeros(nC, dtype=torch.int64, device=dev)
        has_opt = len(self.optimizers) > 0
        newSQ = torch.zeros(nP, **f32) if has_opt else None
        need_buf = has_opt and any(g["momentum"] > 0 for o in self.optimizers for g in o.param_groups)
        newBUF = torch.zeros(nP, **f32) if need_buf else None
        has_ema = len(self.emas) > 0
        newEMA = torch.zeros(nP, **f32) if has_ema else None
        newSEMA = torch.zeros(nS, **f32) if has_ema else None

        def view(arena, off, shape, strides):
            if strides is None:
                n = 1
                for s in shape:
                    n *= s
                return arena[off:off + n].view(shape)
            return torch.as_strided(arena, shape, strides, off)

        # ---- migrate values and re-point tensors
        name_of = {id(p): n for n, p in model.named_parameters()}
        name_of.update({id(b): n for n, b in model.named_buffers()})
        self.param_slots = collections.OrderedDict()
        self.buffer_slots = collections.OrderedDict()
        # The value migration is ~2,500 small copies per rebuild (every parameter, its RMSprop state, its EMA shadow, every BatchNorm
        # buffer): recorded and run as ONE launch of the shrink's job-table kernel (ops.copy_job -> atomnas_gather_jobs) where the pair
        # is fp32 on this device; anything else (the first build from host tensors, int64 counters) is a torch copy as before.
        def move(dst, src):
            if not ops.copy_job(dst, src):
                dst.copy_(src.to(dev))
        # A rebuild inside a deferral window (a forward, profiling pass or optimizer call between a shrink's gathers and their flush):
        # the pending gathers WRITE the module tensors the migration below READS, and one job-table launch has no order between its
        # workgroups -- run them first.  The migration's own jobs are flushed before anything reads the new arenas, and the deferral
        # state is restored even when the walk raises.
        was_deferring = ops.gather_deferring()
        if was_deferring:
            ops.gather_flush()
        else:
            ops.gather_defer(True)
        try:
            with torch.no_grad():
                for arena_name, mod, attr, off, shape, strides in binds:
                    if arena_name == "P":
                        p = mod._parameters[attr]
                        nm = name_of.get(id(p))
                        newv = view(newP, off, shape, strides)
                        move(newv, p.data)
                        # optimizer state and EMA shadows follow the parameter
                        for opt in self.optimizers:
                            st = opt.state.get(p)
                            if st:
                                for key, arena in (("square_avg", newSQ), ("momentum_buffer", newBUF)):
                                    if key in st and arena is not None:
                                        nv = view(arena, off, shape, strides)
                                        move(nv, st[key])
                                        st[key] = nv
                        for ema in self.emas:
                            if nm is not None and nm in ema._shadow:
                                nv = view(newEMA, off, shape, strides)
                                move(nv, ema._shadow[nm])
                                ema._shadow[nm] = nv
                        p.data = newv
                        p.grad = view(newG, off, shape, strides)
                        p._atomnas_off = off
                        p._atomnas_mgr = self
                        if nm is not None:
                            self.param_slots[nm] = (off, shape, strides)
                    else:
                        b = mod._buffers[attr]
                        nm = name_of.get(id(b))
                        if arena_name == "S":
                            newv = view(newS, off, shape, None)
                            move(newv, b)
                            for ema in self.emas:
                                if nm is not None and nm in ema._shadow:
                                    nv = view(newSEMA, off, shape, None)
                                    move(nv, ema._shadow[nm])
                                    ema._shadow[nm] = nv
                        else:
                            newv = newC[off:off + 1].view(shape)
                            newv.copy_(b.to(dev))
                        mod._buffers[attr] = newv
                        if nm is not None:
                            self.buffer_slots[nm] = (arena_name, off, shape)

        finally:
            if not was_deferring:
                ops.gather_defer(False)   # the migration runs here (one launch), before anything reads the new arenas
            else:
                ops.gather_flush()
        self.P, self.G, self.S, self.CNT = newP, newG, newS, newC
        self.SQ, self.BUF, self.EMA, self.SEMA = newSQ, newBUF, newEMA, newSEMA
        self.nP, self.nS = nP, nS
        self.reg_slots = reg_slots

        # running_var padding = 1 so that eval-mode coefficients stay finite on padding channels
        # (padding gamma is 0, so the value never matters)

        # ---- pack buffers and job tables
        self.packT = torch.zeros(max(lpk.size, ALIGN), dtype=T, device=dev)
        self.packF = torch.zeros(max(lpf.size, ALIGN), dtype=torch.float32, device=dev)
        self.jobsT = self._job_table(pack_jobs_t, dev)
        self.jobsF = self._job_table(pack_jobs_f, dev)
        self.njobsT, self.njobsF = len(pack_jobs_t), len(pack_jobs_f)

        # ---- finish the plans (views)
        self.plans = {}
        self._fold_build = []   # fold jobs of the fused blocks: (scratch offset, gradient-arena offset, src_ld, dst_ld, rows, cols)
        self._fold_size = 0
        for m, pl, info in plans:
            pl.mgr = self
            self._finish_plan(m, pl, info)
            object.__setattr__(m, "_plan", pl)
            self.plans[pl.name] = pl
        self._build_fold_table(dev)

        # regulariser job tables (L2 'mnas' over conv/fc weights + classifier bias; L1 filled by the prune module)
        self._build_reg_tables()
        for opt in self.optimizers:
            opt._on_materialize(self)
        for ema in self.emas:
            ema._on_materialize(self)
        self.anchor = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)
        # lr, rho, ema decay, grad scale (atomnas_hip.h: the kernels read slots 0..3); slot 4 is the engine's own: the number of
        # ranks the gradients of this step were summed over (multiplier of the per-rank L1 sub-gradient, engine.TrainStep._opt)
        self.hyper = torch.zeros(8, dtype=torch.float32, device=dev)
        self.hyper[3] = 1.0
        self.hyper[4] = 1.0
        self.hyper_host = torch.zeros(8, dtype=torch.float32)
        self.hyper_host[3] = 1.0
        self.hyper_host[4] = 1.0
        self._hyper_ring = torch.zeros(64, 8, dtype=torch.float32).pin_memory()  # staging slots: the host may run ahead
        self._hyper_slot = 0
        self.step_counter = torch.zeros(1, dtype=torch.int64, device=dev)  # decorrelates dropout masks across iterations
        self.bn_trained = False
        self._depth = 0
        self.dirty = False
        self.version += 1
        self.pack()

    # gradient-completion hook: every executor calls grad_done(plan) at the end of its backward; engine.TrainStep uses it to
    # all-reduce the finished part of the gradient arena while the rest of backward is still running
    grad_done_cb = None

    def grad_done(self, pl):
        cb = self.grad_done_cb
        if cb is not None:
            cb(pl)

    def _bind_bn(self, bind_param, bind_buf, lc, bn, slots, seg, c):
        if bn.affine:
            bind_param(bn, "weight", slots["g"] + seg, (c,))
            bind_param(bn, "bias", slots["b"] + seg, (c,))
        if bn.track_running_stats:
            bind_buf(bn, "running_mean", "S", slots["rm"] + seg, (c,))
            bind_buf(bn, "running_var", "S", slots["rv"] + seg, (c,))
            bind_buf(bn, "num_batches_tracked", "CNT", self._take_counter(lc), ())

    @staticmethod
    def _take_counter(lc):
        off = lc.size
        lc.size += 1
        return off

    @staticmethod
    def _job_table(jobs, dev):
        if not jobs:
            return None

        class J(ctypes.Structure):
            _fields_ = [("src_off", ctypes.c_long), ("dst_off", ctypes.c_long), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                        ("src_ld", ctypes.c_int), ("dst_ld", ctypes.c_int), ("c_off", ctypes.c_int), ("mode", ctypes.c_int)]

        arr = (J * len(jobs))(*[J(*j) for j in jobs])
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        return raw.to(dev)

    def _vec(self, arena, off, n):
        return arena[off:off + n]

    def _finish_plan(self, m, pl, info):
        from .functional import act_code
        P, G, S = self.P, self.G, self.S
        if isinstance(pl, BlockPlan) and getattr(pl, "fused", False):
            self._finish_fused_plan(m, pl, info)
            return
        if isinstance(pl, BlockPlan):
            pl.valid = True
            pl.fused = pl.se = False
            if pl.nb == 0:
                return
            so, pk = info["so"], info["pk"]
            HT = pl.HT

            def bnv(sl, C, mods):
                d = dict(gamma=P[sl["g"]:sl["g"] + C], beta=P[sl["b"]:sl["b"] + C], dgamma=G[sl["g"]:sl["g"] + C],
                         dbeta=G[sl["b"]:sl["b"] + C], rm=S[sl["rm"]:sl["rm"] + C], rv=S[sl["rv"]:sl["rv"] + C], C=C, mods=mods,
                         mgr=self)
                return d

            idx_depth = 1 if pl.expand else 0
            chs = [list(op.children()) for op in m.ops]
            pl.act = act_code(list(chs[0][idx_depth].children())[2])
            if pl.expand:
                pl.We_grad = G[so["We"]:so["We"] + HT * pl.inp]
                pl.bne = bnv(so["bne"], HT, [list(c[0].children())[1] for c in chs])
                pl.We_pack = self._packview(pk["We"])
                pl.WeT_pack = self._packview(pk["WeT"])
            pl.Wd_grad = [G[o:o + pl.segpad(h) * k * k] for o, h, k in zip(so["Wd"], pl.hid, pl.ks)]
            pl.bnd = bnv(so["bnd"], HT, [list(c[idx_depth].children())[1] for c in chs])
            pl.Wp_grad = G[so["Wp"]:so["Wp"] + pl.oup * HT]
            pl.bnp = bnv(so["bnp"], pl.oup, [m.pw_bn])
            pl.Wp_pack = self._packview(pk["Wp"])
            pl.WpT_pack = self._packview(pk["WpT"])
            pl.taps = [self.packF[o:o + kk * c].view(kk, c) for (o, kk, c) in pk["taps"]]
            # running_var of padding channels: 1
            with torch.no_grad():
                for sl, C in ((so.get("bne"), HT), (so["bnd"], HT)):
                    if sl is None:
                        continue
                    rv = S[sl["rv"]:sl["rv"] + C]
                    for s, h in zip(pl.seg, pl.hid):
                        rv[s + h:s + pl.segpad(h)] = 1.0
        else:
            so, pk = info["so"], info["pk"]
            if isinstance(m, nn.Linear):
                pl.W_grad = G[so["W"]:so["W"] + pl.cout * pl.cin]
                pl.bias = P[so["b"]:so["b"] + pad8(pl.cout)] if m.bias is not None else None
                pl.bias_grad = G[so["b"]:so["b"] + pl.cout] if m.bias is not None else None
                pl.W_pack = self._packview(pk["W"])
                pl.WT_pack = self._packview(pk["WT"])
            else:
                conv, bn, actm = list(m.children())
                pl.act = act_code(actm)
                C = pl.cout
                sl = so["bn"]
                pl.bn = dict(gamma=P[sl["g"]:sl["g"] + pad8(C)], beta=P[sl["b"]:sl["b"] + pad8(C)], dgamma=G[sl["g"]:sl["g"] + pad8(C)],
                             dbeta=G[sl["b"]:sl["b"] + pad8(C)], rm=S[sl["rm"]:sl["rm"] + pad8(C)], rv=S[sl["rv"]:sl["rv"] + pad8(C)],
                             C=C, mods=[bn], mgr=self)
                pl.W_grad = G[so["W"]:so["W"] + conv.weight.numel()]
                if "W" in pk:
                    pl.W_pack = self._packview(pk["W"])
                    pl.WT_pack = self._packview(pk["WT"])
                else:
                    o, kk, c = pk["taps"]
                    pl.taps = self.packF[o:o + kk * c].view(kk, c)
                with torch.no_grad():
                    pl.bn["rv"][C:] = 1.0

    def _finish_fused_plan(self, m, pl, info):
        from .functional import act_code
        P, G, S = self.P, self.G, self.S
        so, pk = info["so"], info["pk"]
        HT, total = pl.HT, pl.total
        pl.valid = True
        segs = [(sg, st, h) for sg, st, h in zip(pl.seg, pl.start, pl.hid)]   # (padded offset, contiguous offset, channels)
        # padded kernel channel -> index into the module's contiguous [total] vectors, -1 for the padding between branch segments
        cmap = torch.full((HT,), -1, dtype=torch.int32)
        for sg, st, h in segs:
            cmap[sg:sg + h] = torch.arange(st, st + h, dtype=torch.int32)
        pl.cmap = cmap.to(P.device)

        def bnv(sl, C, mods, segmented):
            d = dict(gamma=P[sl["g"]:sl["g"] + C], beta=P[sl["b"]:sl["b"] + C], dgamma=G[sl["g"]:sl["g"] + C],
                     dbeta=G[sl["b"]:sl["b"] + C], rm=S[sl["rm"]:sl["rm"] + C], rv=S[sl["rv"]:sl["rv"] + C], C=C, mods=mods, mgr=self)
            if segmented:   # contiguous [total] vectors, padded [HT] kernel layout
                d["cmap"], d["Cpad"] = pl.cmap, HT
            return d

        idx = 1 if pl.expand else 0
        dws = [list(op.children())[idx] for op in m.depth_ops]
        pl.act = act_code(list(dws[0].children())[2])
        if pl.expand:
            pl.We_grad = G[so["We"]:so["We"] + total * pl.inp]
            pl.bne = bnv(so["bne"], total, [list(m.expand_conv.children())[1]], True)
            pl.We_pack, pl.WeT_pack = self._packview(pk["We"]), self._packview(pk["WeT"])
        pl.Wd_grad = [G[o:o + pl.segpad(h) * k * k] for o, h, k in zip(so["Wd"], pl.hid, pl.ks)]
        pl.bnd = bnv(so["bnd"], HT, [list(d.children())[1] for d in dws], False)
        pl.Wp_grad = G[so["Wp"]:so["Wp"] + pl.oup * total]
        # one weight-gradient GEMM per layer into a padded scratch matrix, folded into the contiguous tensors per segment
        pl.fold_first = len(self._fold_build)
        o_p = self._fold_take(pl.oup * HT)
        pl.Wp_scratch_off = o_p
        for sg, st, h in segs:
            self._fold_build.append((o_p + sg, so["Wp"] + st, HT, total, pl.oup, h))
        pl.fold_np = len(self._fold_build) - pl.fold_first
        pl.fold_ne = 0
        if pl.expand:
            o_e = self._fold_take(HT * pl.inp)
            pl.We_scratch_off = o_e
            for sg, st, h in segs:
                self._fold_build.append((o_e + sg * pl.inp, so["We"] + st * pl.inp, h * pl.inp, h * pl.inp, 1, h * pl.inp))
            pl.fold_ne = len(self._fold_build) - pl.fold_first - pl.fold_np
        pl.bnp = bnv(so["bnp"], pl.oup, [list(m.project_conv.children())[1]], False)
        pl.Wp_pack, pl.WpT_pack = self._packview(pk["Wp"]), self._packview(pk["WpT"])
        pl.taps = [self.packF[o:o + kk * c].view(kk, c) for (o, kk, c) in pk["taps"]]
        with torch.no_grad():
            rv = S[so["bnd"]["rv"]:so["bnd"]["rv"] + HT]
            for sg, h in zip(pl.seg, pl.hid):
                rv[sg + h:sg + pl.segpad(h)] = 1.0
        if pl.se:
            n1, n2 = pl.se_hid * total, total * pl.se_hid
            pl.se_w1, pl.se_b1 = P[so["se1w"]:so["se1w"] + n1], P[so["se1b"]:so["se1b"] + pl.se_hid]
            pl.se_w2, pl.se_b2 = P[so["se2w"]:so["se2w"] + n2], P[so["se2b"]:so["se2b"] + total]
            pl.se_dw1, pl.se_db1 = G[so["se1w"]:so["se1w"] + n1], G[so["se1b"]:so["se1b"] + pl.se_hid]
            pl.se_dw2, pl.se_db2 = G[so["se2w"]:so["se2w"] + n2], G[so["se2b"]:so["se2b"] + total]
            pl.se_act = act_code(m.se_op.active_fn)
            o1, o2, o3 = pk["se"]
            nh = pl.se_hid * HT
            pl.se_w1p, pl.se_w2t, pl.se_b2p = self.packF[o1:o1 + nh], self.packF[o2:o2 + nh], self.packF[o3:o3 + HT]

    # ---- fused blocks: padded scratch matrices of the expand / projection weight gradients and the table that folds them into the
    # contiguous gradient tensors (ops.fold_jobs, csrc/reduce.hip k_fold_jobs)
    def _fold_take(self, n):
        off = self._fold_size
        self._fold_size = _align(off + n)
        return off

    def _build_fold_table(self, dev):
        jobs = self._fold_build
        self.FW = torch.zeros(max(self._fold_size, ALIGN), dtype=torch.float32, device=dev)   # zero once: every fold clears what it read
        self.fold_table, self.fold_blk0 = None, [0]
        if not jobs:
            return

        class J(ctypes.Structure):
            _fields_ = [("src", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("src_ld", ctypes.c_long), ("dst_ld", ctypes.c_long),
                        ("rows", ctypes.c_int), ("cols", ctypes.c_int), ("blk0", ctypes.c_uint), ("pad_", ctypes.c_int)]

        arr = (J * len(jobs))()
        blk = 0
        for q, (so, do, sld, dld, rows, cols) in enumerate(jobs):
            arr[q] = J(self.FW.data_ptr() + 4 * so, self.G.data_ptr() + 4 * do, sld, dld, rows, cols, blk, 0)
            blk += (rows * cols + 255) // 256
            self.fold_blk0.append(blk)
        self.fold_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone().to(dev)

    def _packview(self, t):
        off, rows, ld = t
        return self.packT[off:off + rows * ld].view(rows, ld)

    # ------------------------------------------------------------------ regulariser tables
    def _build_reg_tables(self):
        """Job tables (offset, count, coefficient) for cal_l2_loss 'mnas'/'slimmable' (utils/optim.py:210-249)."""
        self.l2_tables = {}

    def reg_table(self, entries):
        """entries: list of (offset, count, coef) -> device table for atomnas_reg_value / atomnas_reg_grad"""
        class J(ctypes.Structure):
            _fields_ = [("off", ctypes.c_long), ("count", ctypes.c_int), ("coef", ctypes.c_float)]

        arr = (J * len(entries))(*[J(int(o), int(c), float(k)) for o, c, k in entries])
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        return raw.to(self.device), len(entries)

    # ------------------------------------------------------------------ per-step helpers
    def pack(self):
        """Refreshes the packed (kernel-layout, compute-dtype) copies of all weights from the fp32 master arena."""
        if self.njobsT:
            ops.pack_weights(self.P, self.packT, self.jobsT, self.njobsT, self.compute_dtype)
        if self.njobsF:
            ops.pack_weights(self.P, self.packF, self.jobsF, self.njobsF, torch.float32)

    def zero_grad(self):
        ops.zero_(self.G)

    def push_hyper(self):
        """host -> device copy of (lr, rho, EMA decay, grad scale), stream-ordered before the kernels that read them"""
        slot = self._hyper_ring[self._hyper_slot % 64]
        self._hyper_slot += 1
        slot.copy_(self.hyper_host)
        self.hyper.copy_(slot, non_blocking=True)

    # Outermost module call: refresh packed weights on entry; on exit bump every BN's num_batches_tracked once if any BN
    # ran with batch statistics (all BNs of the model share their mode in the reference's train / calibration phases).
    # ---- per-step statistics workspace: every BatchNorm use of a step (forward and backward) takes a distinct slice of partial
    # rows.  The producing kernels write every row (include/atomnas_hip.h), so nothing is ever cleared.  A slice is dead as soon
    # as its finalize kernel has run, so re-using the workspace from the next top-level forward on is safe even when a
    # backward is still pending.
    def take_stats(self, n):
        n = (n + 63) // 64 * 64
        off = self._stats_off
        self._stats_off = off + n
        self._stats_need = max(self._stats_need, self._stats_off)
        ws = self._stats_ws
        if ws is None or off + n > ws.numel():
            return None    # first steps: the caller allocates; the workspace is sized at the next top-level forward
        return ws[off:off + n]

    def _reset_stats(self):
        ws = self._stats_ws
        if self._stats_need > (ws.numel() if ws is not None else 0):
            self._stats_ws = torch.empty(int(self._stats_need * 1.25) // 64 * 64 + 64, dtype=torch.float32, device=self.device)
        self._stats_off = 0

    def enter(self):
        self.ensure()
        if self._depth == 0:
            self.pack()
            self.bn_trained = False
            self._reset_stats()
        self._depth += 1

    def leave(self):
        self._depth -= 1
        if self._depth == 0 and self.bn_trained:
            ops.add_i64(self.CNT, 1)
            self.bn_trained = False


def manager_of(module):
    """The arena manager that owns `module`: the one of the enclosing model once that has been materialised, otherwise a
    private one created on demand (a block or ConvBNReLU used on its own, as the reference's unit tests do)."""
    pl = getattr(module, "_plan", None)
    if pl is not None and getattr(pl, "mgr", None) is not None and pl.mgr.model is not None:
        return pl.mgr
    mgr = getattr(module, "_arena", None)
    if mgr is None:
        mgr = ArenaManager(module)
        object.__setattr__(module, "_arena", mgr)
    return mgr


def plan_of(module):
    """Plan (arena views) of `module`, (re)materialising its arenas if the structure changed since."""
    mgr = manager_of(module)
    mgr.ensure()
    pl = getattr(module, "_plan", None)
    if pl is None or pl.mgr is not mgr:
        mgr.mark_dirty()
        mgr.ensure()
        pl = module._plan
    return pl
