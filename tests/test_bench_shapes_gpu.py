"""Bench-size sweep (GPU): EVERY distinct launch of the timed training step, at the bench's own sizes, every element.

tests/golden/bench_shapes.json (tools/make_bench_shapes.py: one eager step of the AtomNAS-C and AtomNAS-A supernets at batch 256 /
224 x 224 / bf16 with atomnas_amd.ops.RECORD on) lists the step's launches by entry point, sizes, prologue / epilogue / statistics modes,
activation layouts, pitches and workspace sizes.  Every row is replayed here on seeded random data:

  * outputs pre-filled with NaN (accumulating outputs with a constant), statistics rows and workspaces pre-filled with NaN;
  * THREE back-to-back launches into three separate output sets on a busy stream (the launches queue behind a large fill: round 5's
    store hazard of k_gemm_nt_st needed a backed-up store queue and hit different elements in every launch);
  * compared with torch on the GPU (fp32 / fp64) element by element and through the statistics rows.

Why: the reference has no numerics tests for conv / BN; this repository's kernel tests ran at M <= 70 k while the
7 x 7 expand GEMM of the bench wrote garbage into 0.01 % of its outputs for two rounds (profiles/r05_st_store_hazard.txt).
Bounds: one bf16 rounding of the output (2^-8 relative) plus an absolute floor of 1-2 % of the output's rms for operand roundings
(prologue results and matrix-core operands are rounded to bf16 / fp16 inside the kernels); statistics against the sums of the values
the kernel itself stored.
"""
import json
import os
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
TABLE = os.path.join(HERE, "golden", "bench_shapes.json")
ROWS = json.load(open(TABLE)) if os.path.exists(TABLE) else []
REPEATS = 3


def _id(r):
    skip = ("entry", "nets", "dt", "ws_floats", "stat_rows", "part_rows", "ldw", "stat_ld", "count", "momentum")
    parts = [r["entry"]]
    for k in sorted(r):
        if k in skip or r[k] is None or r[k] is False:
            continue
        v = r[k]
        if isinstance(v, list):
            v = "+".join(str(e) for e in v)
        parts.append("%s%s" % (k, "" if v is True else v))
    return "-".join(str(p) for p in parts)


def pad(c, a):
    return (c + a - 1) // a * a


class Ctx:
    def __init__(self, seed):
        from atomnas_amd import ops
        self.ops = ops
        self.g = torch.Generator(device="cuda").manual_seed(seed)

    def randn(self, *s, scale=1.0):
        return torch.randn(*s, device="cuda", generator=self.g) * scale

    def rand(self, *s):
        return torch.rand(*s, device="cuda", generator=self.g)

    def T(self, dt):
        return torch.bfloat16 if dt == 1 else torch.float32

    # ---- activations in the recorded layout
    def act_in(self, val, lay, dtype):
        """val: fp32 [M, C] -> (kernel argument, the values as stored: fp32 [M, C])"""
        M, C = val.shape
        q = val.to(dtype)
        if lay == "slab":
            buf = torch.zeros(M, pad(C, 16), dtype=dtype, device="cuda")
            buf[:, :C] = q
            return self.ops.Slab.from_plain(buf, C), q.float()
        buf = torch.zeros(M, int(lay), dtype=dtype, device="cuda")
        buf[:, :C] = q
        return buf, q.float()

    def act_out(self, M, C, lay, dtype):
        if lay == "slab":
            s = self.ops.Slab(M, C, dtype, "cuda")
            s.t.fill_(float("nan"))
            return s
        buf = torch.zeros(M, int(lay), dtype=dtype, device="cuda")
        buf[:, :C] = float("nan")
        return buf

    @staticmethod
    def read(t, C):
        from atomnas_amd.ops import Slab
        p = t.to_plain() if isinstance(t, Slab) else t
        if p.shape[1] > C:   # padding channels of a tensor must stay zero
            cp = pad(C, 8)
            assert float(p[:, C:cp].float().abs().max() if cp > C else 0.0) == 0.0
        return p[:, :C].float()

    def cvec(self, v):
        out = torch.zeros(pad(v.numel(), 8) + 8, dtype=torch.float32, device="cuda")
        out[:v.numel()] = v
        return out

    def stats(self, rows, ld):
        return torch.full((rows, 2, ld), float("nan"), dtype=torch.float32, device="cuda")

    def weights(self, W, ldw, dtype, rows_to=64):
        """[N, K] fp32 -> packed [N padded][ldw] in the storage dtype, padding zero; returns (packed, the values as stored)"""
        n, k = W.shape
        buf = torch.zeros(pad(n, rows_to), ldw, dtype=dtype, device="cuda")
        buf[:n, :k] = W.to(dtype)
        return buf, buf[:n, :k].float()


def busy():
    """a large fill queued in front of the launches: they are then issued back to back behind work in flight"""
    b = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    b.fill_(1.0)
    return b


def act_fwd(pre, code):
    return {0: pre, 1: torch.relu(pre), 2: pre.clamp(0, 6), 3: pre * torch.sigmoid(pre)}[code]


def act_grad(pre, code):
    if code == 0:
        return torch.ones_like(pre)
    if code == 1:
        return (pre > 0).float()
    if code == 2:
        return ((pre > 0) & (pre < 6)).float()
    s = torch.sigmoid(pre)
    return s * (1 + pre * (1 - s))


def margin(ctx, M, C, scale_vec, shift_vec, dtype, amp=1.0, gap=0.05):
    """raw tensor z [M, C] (as stored in dtype) whose pre-activation z*scale+shift stays `gap` away from zero: a ReLU mask compared
    element by element must not depend on how a multiply-add is contracted"""
    t = ctx.randn(M, C) * amp
    t = torch.where(t >= 0, t + gap, t - gap)
    z = ((t - shift_vec) / scale_vec).to(dtype)
    return z.float()


# ATOMNAS_SWEEP_DIGESTS=<file>: every checked output / statistics row of every replayed launch is also written there as a SHA-1 of its
# bytes (the inputs are seeded per row), so that two builds of the library can be compared BIT FOR BIT over the step's launches:
#   ATOMNAS_HIP_LIB=old.so ATOMNAS_SWEEP_DIGESTS=a.txt pytest tests/test_bench_shapes_gpu.py -m gpu; ... new ...; diff a.txt b.txt
DIGESTS = os.environ.get("ATOMNAS_SWEEP_DIGESTS")
_CUR = [""]


def _digest(name, t):
    if DIGESTS:
        import hashlib
        with open(DIGESTS, "a") as f:
            f.write("%s %s %s\n" % (_CUR[0], name, hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()))


def check(name, got, ref, rtol=1.2e-2, afrac=1e-2, outliers=0):
    """every element: |got - ref| <= rtol |ref| + afrac * rms(ref)"""
    _digest(name, got)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert bool(torch.isfinite(got).all()), "%s: non-finite outputs (unwritten or corrupted elements)" % name
    rms = float(ref.float().pow(2).mean().sqrt())
    bad = ~((got - ref).abs() <= rtol * ref.abs() + afrac * rms)
    nbad = int(bad.sum())
    if nbad > outliers:
        idx = torch.nonzero(bad)[:4].tolist()
        raise AssertionError("%s: %d of %d elements off (rms %.4g), first %s got %s ref %s" % (
            name, nbad, bad.numel(), rms, idx, [float(got[tuple(i)]) for i in idx], [float(ref[tuple(i)]) for i in idx]))


def check_sums(name, stat_row, terms, rtol=1e-4, afrac=1e-5):
    """a statistics value against the fp64 column sums of `terms` (computed from what the kernel stored): error budget relative to
    the sum of the absolute terms (fp32 partial sums in a fixed order)"""
    _digest(name, stat_row)
    ref = terms.double().sum(0)
    mag = terms.double().abs().sum(0)
    bad = ~((stat_row.double() - ref).abs() <= rtol * ref.abs() + afrac * mag + 1e-30)
    assert int(bad.sum()) == 0, "%s: %d of %d channel sums off, first %s got %s ref %s" % (
        name, int(bad.sum()), bad.numel(), torch.nonzero(bad)[:3].flatten().tolist(), stat_row[bad][:3].tolist(), ref[bad][:3].tolist())


def stat_sum(st):
    assert not bool(torch.isnan(st).any()), "statistics rows left unwritten"
    return st.sum(0)


# ------------------------------------------------------------------------------------------------------------------ entries
def run_pw_gemm_nt(ctx, r):
    ops = ctx.ops
    M, N, K = r["M"], r["N"], r["K"]
    T = ctx.T(r["dt"])
    A = ctx.randn(M, K)
    Aarg, Aq = ctx.act_in(A, r["a"], T)
    W = ctx.randn(N, K) / K ** 0.5
    Wp, Wq = ctx.weights(W, r["ldw"], T)
    kw = {}
    mode = r["a_mode"]
    if mode == 1:
        c1, c2 = ctx.rand(K) + 0.5, ctx.randn(K) * 0.3
        Aeff = act_fwd(Aq * c1 + c2, r["a_relu"])
        kw.update(a_mode=1, ac1=ctx.cvec(c1), ac2=ctx.cvec(c2), a_relu=r["a_relu"])
    elif mode == 2:
        c1, c2, c3 = ctx.rand(K) + 0.5, ctx.randn(K) * 0.2, ctx.randn(K) * 0.2
        A2arg, A2q = ctx.act_in(ctx.randn(M, K), r["a2"], T)
        Aeff = c1 * Aq + c2 * A2q + c3
        kw.update(a_mode=2, a2=A2arg, ac1=ctx.cvec(c1), ac2=ctx.cvec(c2), ac3=ctx.cvec(c3))
    else:
        Aeff = Aq
    Aeff = Aeff.to(T).float()   # the prologue result is rounded to the storage type before the MFMA
    ref = (Aeff.double() @ Wq.double().t()).float() if K > 512 else Aeff @ Wq.t()
    if r["bias"]:
        b = ctx.randn(N)
        ref = ref + b
        kw["bias"] = ctx.cvec(b)
    if r["add"] is not None:
        addarg, addq = ctx.act_in(ctx.randn(M, N), r["add"], T)
        ref = ref + addq
        kw["add"] = addarg
    Zq = None
    if r["z"] is not None:
        zs, zh = ctx.rand(N) + 0.5, ctx.randn(N) * 0.3
        Zq = margin(ctx, M, N, zs, zh, T) if r["mask"] else ctx.randn(M, N).to(T).float()
        Zarg, _ = ctx.act_in(Zq, r["z"], T)
        kw.update(z=Zarg, zscale=ctx.cvec(zs), zshift=ctx.cvec(zh), mask=r["mask"])
        if r["mask"]:
            ref = ref * act_grad(Zq * zs + zh, r["mask"])
    out_T = torch.float32 if r["out_f32"] else T
    outs, sts = [], []
    hold = busy()
    for _ in range(REPEATS):
        C = ctx.act_out(M, N, r["c"], out_T)
        st = ctx.stats(r["stat_rows"], N) if r["stat_mode"] else None
        ops.gemm_nt(Aarg, Wp, C, M, N, K, stats=st, stat_mode=r["stat_mode"], stat_rows=r["stat_rows"] if r["stat_mode"] else None, **kw)
        outs.append(C)
        sts.append(st)
    torch.cuda.synchronize()
    del hold
    for i, (C, st) in enumerate(zip(outs, sts)):
        got = ctx.read(C, N)
        check("C[%d]" % i, got, ref, rtol=1.2e-2 if out_T == torch.bfloat16 else 1e-3, afrac=1e-2 if T == torch.bfloat16 else 1e-4)
        if st is not None:
            s = stat_sum(st)
            check_sums("stat0[%d]" % i, s[0], got)
            check_sums("stat1[%d]" % i, s[1], got * got if r["stat_mode"] == 1 else got * Zq)
    assert all(torch.equal(ctx.read(outs[0], N), ctx.read(o, N)) for o in outs[1:]), "repeated launches differ"


def run_pw_gemm_tn(ctx, r):
    ops = ctx.ops
    M, NU, NV = r["M"], r["NU"], r["NV"]
    T = ctx.T(r["dt"])

    def operand(n, mode, lay, lay2, relu):
        X = ctx.randn(M, n)
        arg, q = ctx.act_in(X, lay, T)
        kw = {}
        if mode == 1:
            c1, c2 = ctx.rand(n) + 0.5, ctx.randn(n) * 0.3
            eff = act_fwd(q * c1 + c2, relu)
            kw = dict(c1=ctx.cvec(c1), c2=ctx.cvec(c2), relu=relu)
        elif mode == 2:
            c1, c2, c3 = ctx.rand(n) + 0.5, ctx.randn(n) * 0.2, ctx.randn(n) * 0.2
            arg2, q2 = ctx.act_in(ctx.randn(M, n), lay2, T)
            eff = c1 * q + c2 * q2 + c3
            kw = dict(c1=ctx.cvec(c1), c2=ctx.cvec(c2), c3=ctx.cvec(c3), x2=arg2)
        else:
            eff = q
        return arg, eff.to(T).double(), kw

    U, Ue, ku = operand(NU, r["u_mode"], r["u"], r["u2"], r["u_relu"])
    V, Ve, kv = operand(NV, r["v_mode"], r["v"], r["v2"], r["v_relu"])
    ref = (Ue.t() @ Ve).float()
    si, sj = r["si"], r["sj"]
    n_out = (NU - 1) * si + (NV - 1) * sj + 1
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        out = torch.full((n_out,), 0.5, dtype=torch.float32, device="cuda")   # the entry point ACCUMULATES into the gradient arena
        ws = torch.full((r["ws_floats"],), float("nan"), dtype=torch.float32, device="cuda") if r["ws_floats"] else False
        ops.gemm_tn(U, NU, V, NV, out, si, sj, M, u_mode=r["u_mode"], u2=ku.get("x2"), uc1=ku.get("c1"), uc2=ku.get("c2"), uc3=ku.get("c3"),
                    u_relu=ku.get("relu", 0), v_mode=r["v_mode"], v2=kv.get("x2"), vc1=kv.get("c1"), vc2=kv.get("c2"), vc3=kv.get("c3"),
                    v_relu=kv.get("relu", 0), ws=ws)
        outs.append(out)
    torch.cuda.synchronize()
    del hold
    for i, out in enumerate(outs):
        got = torch.as_strided(out, (NU, NV), (si, sj)) - 0.5
        check("dW[%d]" % i, got, ref, rtol=2e-3, afrac=3e-3)
    assert all(torch.equal(outs[0], o) for o in outs[1:]), "repeated launches differ (fixed-order partial sums)"


def _dw_reference(ctx, r, direction):
    """inputs and the torch reference of a depthwise row.  Returns a dict of tensors."""
    N, H, W, C, k, s = r["N"], r["H"], r["W"], r["C"], r["k"], r["stride"]
    T = ctx.T(r["dt"])
    P = (k - 1) // 2
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    M, M2 = N * H * W, N * Ho * Wo
    d = dict(N=N, H=H, W=W, C=C, k=k, s=s, T=T, Ho=Ho, Wo=Wo, M=M, M2=M2)
    w = ctx.randn(C, 1, k, k) * 0.3
    taps = torch.zeros(k * k, r["ldw"], dtype=torch.float32, device="cuda")
    taps[:, :C] = w.reshape(C, k * k).t()
    d["w"], d["taps"] = w, taps
    nchw = lambda t2, h_, w_: t2.reshape(N, h_, w_, C).permute(0, 3, 1, 2)
    d["nchw"] = nchw
    if r["fused_in"]:
        sc, sh = ctx.rand(C) + 0.5, ctx.randn(C) * 0.3
        xq = margin(ctx, M, C, sc, sh, T) if direction == "bwd" else ctx.randn(M, C).to(T).float()
        d["sc"], d["sh"] = ctx.cvec(sc), ctx.cvec(sh)
        pre = nchw(xq, H, W) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)
    else:
        xq = ctx.randn(M, C).to(T).float()
        d["sc"] = d["sh"] = None
        pre = nchw(xq, H, W)
    d["xq"], d["pre"] = xq, pre
    d["xarg"], _ = ctx.act_in(xq, r["x"], T)
    return d


def run_dwconv_fwd(ctx, r):
    ops = ctx.ops
    d = _dw_reference(ctx, r, "fwd")
    N, H, W, C, k, s, T = d["N"], d["H"], d["W"], d["C"], d["k"], d["s"], d["T"]
    xa = act_fwd(d["pre"], r["act"]) if r["fused_in"] else d["pre"]
    yref = F.conv2d(xa, d["w"], None, s, (k - 1) // 2, 1, C).permute(0, 2, 3, 1).reshape(d["M2"], C)
    outs, sts = [], []
    hold = busy()
    for _ in range(REPEATS):
        y = ctx.act_out(d["M2"], C, r["y"], T)
        st = ctx.stats(r["stat_rows"], r["stat_ld"]) if r["stats"] else None
        ops.dwconv_fwd(d["xarg"], d["sc"], d["sh"], r["act"], d["taps"], y, st, r["stat_ld"], N, H, W, C, k, s,
                       stat_rows=r["stat_rows"] if r["stats"] else None)
        outs.append(y)
        sts.append(st)
    torch.cuda.synchronize()
    del hold
    for i, (y, st) in enumerate(zip(outs, sts)):
        got = ctx.read(y, C)
        check("y[%d]" % i, got, yref, rtol=1.2e-2, afrac=1e-2)
        if st is not None:
            sm = stat_sum(st[:, :, :C])
            check_sums("sum y[%d]" % i, sm[0], got)
            check_sums("sum y^2[%d]" % i, sm[1], got * got)
    assert all(torch.equal(ctx.read(outs[0], C), ctx.read(o, C)) for o in outs[1:]), "repeated launches differ"


def _dw_bwd_case(ctx, r):
    """inputs, kernel arguments and torch references of one depthwise backward (a launch of its own or a branch of a fused launch)"""
    d = _dw_reference(ctx, r, "bwd")
    C, k, s, T = d["C"], d["k"], d["s"], d["T"]
    M, M2 = d["M"], d["M2"]
    gq = (ctx.randn(M2, C) * 1e-2).to(T).float()
    d["garg"], _ = ctx.act_in(gq, r["g"], T)
    if r["yraw"] is not None:
        yq = ctx.randn(M2, C).to(T).float()
        d["yarg"], _ = ctx.act_in(yq, r["yraw"], T)
        c1, c2, c3 = ctx.rand(C) + 0.5, ctx.randn(C) * 1e-3, ctx.randn(C) * 1e-3
        dy = c1 * gq + c2 * yq + c3
        d["cc"] = [ctx.cvec(c1), ctx.cvec(c2), ctx.cvec(c3)]
    else:
        d["yarg"], dy, d["cc"] = None, gq, [None, None, None]
    pre = d["pre"].detach().requires_grad_(True)
    xa = act_fwd(pre, r["act"]) if r["fused_in"] else pre
    wr = d["w"].clone().requires_grad_(True)
    y = F.conv2d(xa, wr, None, s, (k - 1) // 2, 1, C)
    (y * d["nchw"](dy, d["Ho"], d["Wo"])).sum().backward()
    d["href"] = pre.grad.permute(0, 2, 3, 1).reshape(M, C)     # = dwconv^T(dY) * act'(pre)
    d["dwref"] = wr.grad.reshape(C, k * k)
    del d["pre"]
    return d


def _dw_bwd_outputs(ctx, r, d):
    C, k = d["C"], d["k"]
    h = ctx.act_out(d["M"], C, r["h"], d["T"])
    dw = torch.full((C * k * k,), 0.25, dtype=torch.float32, device="cuda") if r["dw"] else None
    st = ctx.stats(r["part_rows"], r["stat_ld"]) if r["stats"] else None
    ws = torch.full((r["part_rows"] * C * k * k,), float("nan"), dtype=torch.float32, device="cuda") if r["dw"] else None
    return h, dw, st, ws


def _dw_bwd_check(ctx, d, outs, tag=""):
    C, k = d["C"], d["k"]
    for i, (h, dw, st) in enumerate(outs):
        got = ctx.read(h, C)
        # the matrix-core backward (k = 7, row-ring tiles) rounds dYraw and the taps to bf16: single large terms put the extreme
        # elements of 1e8 at 2.2 % of the rms (measured: 61 of 115 M beyond 2 %); a corrupted store is off by >= the rms itself
        check("%sh[%d]" % (tag, i), got, d["href"], rtol=1.2e-2, afrac=4e-2)
        if dw is not None:
            check("%sdw[%d]" % (tag, i), dw.view(C, k * k) - 0.25, d["dwref"], rtol=3e-3, afrac=5e-3)
        if st is not None:
            sm = stat_sum(st[:, :, :C])
            check_sums("%ssum h[%d]" % (tag, i), sm[0], got)
            check_sums("%ssum h*x[%d]" % (tag, i), sm[1], got * d["xq"])
    assert all(torch.equal(ctx.read(outs[0][0], C), ctx.read(o[0], C)) for o in outs[1:]), "repeated launches differ"
    if outs[0][1] is not None:
        assert all(torch.equal(outs[0][1], o[1]) for o in outs[1:]), "weight gradients of repeated launches differ"


def run_dwconv_bwd(ctx, r):
    ops = ctx.ops
    d = _dw_bwd_case(ctx, r)
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        h, dw, st, ws = _dw_bwd_outputs(ctx, r, d)
        ops.dwconv_bwd(d["garg"], d["yarg"], d["cc"][0], d["cc"][1], d["cc"][2], d["xarg"], d["sc"], d["sh"], r["act"], d["taps"], h, dw, st,
                       r["stat_ld"], d["N"], d["H"], d["W"], d["C"], d["k"], d["s"], stat_rows=r["part_rows"], dw_ws=ws)
        outs.append((h, dw, st))
    torch.cuda.synchronize()
    del hold
    _dw_bwd_check(ctx, d, outs)


def run_expand_bwd(ctx, r):
    """atomnas_expand_bwd (the expand backward without E): gx = (c1*h) We (+ x mp^T + vb) (+ add), dwe += (c1*h)^T x"""
    ops = ctx.ops
    M, inp, hid = r["M"], r["inp"], r["hid"]
    T = ctx.T(r["dt"])
    harg, hq = ctx.act_in(ctx.randn(M, hid), r["h"], T)
    xarg, xq = ctx.act_in(ctx.randn(M, inp), r["x"], T)
    We = ctx.randn(hid, inp) / hid ** 0.5
    wt, wtq = ctx.weights(We.t().contiguous(), r["ldw"], T)       # We^T packed [inp pad64][ldw]
    c1 = ctx.rand(hid) + 0.5
    dE = c1 * hq
    ref_gx = (dE.double() @ wtq.double().t()).float()
    kw = {}
    if r["mp"]:
        Mm = ctx.randn(inp, inp) / inp ** 0.5
        mp, mpq = ctx.weights(Mm, r["ldm"], T)
        vb = ctx.randn(inp) * 0.1
        ref_gx = ref_gx + xq @ mpq.t() + vb
        kw.update(mp=mp, vb=ctx.cvec(vb))
    addarg = None
    if r["add"] is not None:
        addarg, addq = ctx.act_in(ctx.randn(M, inp), r["add"], T)
        ref_gx = ref_gx + addq
    ref_dw = (dE.double().t() @ xq.double()).float()              # [hid, inp]
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        gx = ctx.act_out(M, inp, r["gx"], T)
        dwe = torch.full((hid * inp,), 0.25, dtype=torch.float32, device="cuda")
        ws = torch.full((r["ws_floats"],), float("nan"), dtype=torch.float32, device="cuda")
        ops.expand_bwd(harg, ctx.cvec(c1), xarg, wt, addarg, gx, dwe, M, inp, hid, ws=ws, **kw)
        outs.append((gx, dwe))
    torch.cuda.synchronize()
    del hold
    for i, (gx, dwe) in enumerate(outs):
        check("gx[%d]" % i, ctx.read(gx, inp), ref_gx, rtol=1.2e-2, afrac=1.5e-2)
        check("dwe[%d]" % i, dwe.view(hid, inp) - 0.25, ref_dw, rtol=3e-3, afrac=5e-3)
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:]), "repeated launches differ"


def run_project_bwd(ctx, r):
    """atomnas_project_bwd on the materialised dP: gh = act'(z*zs+zh) * (dP Wp), statistics [sum gh, sum gh*z], dwp += dP^T act(z*zs+zh)"""
    ops = ctx.ops
    M, oup, hid = r["M"], r["oup"], r["hid"]
    T = ctx.T(r["dt"])
    garg, gq = ctx.act_in(ctx.randn(M, oup), r["g"], T)
    Wp = ctx.randn(oup, hid) / oup ** 0.5
    wpt, wptq = ctx.weights(Wp.t().contiguous(), r["ldw"], T)      # Wp^T packed [hid pad64][ldw]
    zs, zh = ctx.rand(hid) + 0.5, ctx.randn(hid) * 0.3
    zq = margin(ctx, M, hid, zs, zh, T)
    zarg, _ = ctx.act_in(zq, r["z"], T)
    pre = zq * zs + zh
    ref_gh = (gq @ wptq.t()) * act_grad(pre, r["act"])
    A = act_fwd(pre, r["act"]).to(T).double()
    ref_dw = (gq.double().t() @ A).float()                          # [oup, hid]
    si, sj = r["si"], r["sj"]
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        gh = ctx.act_out(M, hid, r["gh"], T)
        st = ctx.stats(r["stat_rows"], hid)
        dwp = torch.full(((oup - 1) * si + (hid - 1) * sj + 1,), 0.25, dtype=torch.float32, device="cuda")
        ws = torch.full((r["ws_floats"],), float("nan"), dtype=torch.float32, device="cuda")
        ops.project_bwd(garg, wpt, zarg, ctx.cvec(zs), ctx.cvec(zh), r["act"], gh, st, dwp, si, sj, M, oup, hid,
                        stat_rows=r["stat_rows"], ws=ws)
        outs.append((gh, st, dwp))
    torch.cuda.synchronize()
    del hold
    for i, (gh, st, dwp) in enumerate(outs):
        got = ctx.read(gh, hid)
        check("gh[%d]" % i, got, ref_gh, rtol=1.2e-2, afrac=1e-2)
        sm = stat_sum(st)
        check_sums("sum gh[%d]" % i, sm[0], got)
        check_sums("sum gh*z[%d]" % i, sm[1], got * zq)
        check("dwp[%d]" % i, torch.as_strided(dwp, (oup, hid), (si, sj)) - 0.25, ref_dw, rtol=3e-3, afrac=5e-3)
    assert all(torch.equal(ctx.read(outs[0][0], hid), ctx.read(o[0], hid)) and torch.equal(outs[0][2], o[2]) for o in outs[1:]), "repeated launches differ"


def run_gram(ctx, r):
    ops = ctx.ops
    M, inp = r["M"], r["inp"]
    T = ctx.T(r["dt"])
    xarg, xq = ctx.act_in(ctx.randn(M, inp), r["x"], T)
    refG = (xq.double().t() @ xq.double()).float()
    refs = xq.double().sum(0).float()
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        G = torch.full((inp * inp,), float("nan"), dtype=torch.float32, device="cuda")
        sx = torch.full((inp,), float("nan"), dtype=torch.float32, device="cuda")
        ws = torch.full((r["ws_floats"],), float("nan"), dtype=torch.float32, device="cuda")
        ops.gram(xarg, M, inp, G, sx, ws=ws)
        outs.append((G, sx))
    torch.cuda.synchronize()
    del hold
    for i, (G, sx) in enumerate(outs):
        check("G[%d]" % i, G.view(inp, inp), refG, rtol=1e-3, afrac=1e-3)
        check("sx[%d]" % i, sx, refs, rtol=1e-3, afrac=1e-3)
    assert all(torch.equal(outs[0][0], o[0]) for o in outs[1:])


def run_xb_coeffs(ctx, r):
    """mp = bf16(We^T diag(c2) We), vb = c3^T We, dwe += diag(c2) We G + c3 sx^T  (include/atomnas_hip.h)"""
    ops = ctx.ops
    inp, C = r["inp"], r["C"]
    T = torch.bfloat16
    We = ctx.randn(C, inp) / inp ** 0.5
    wexp, Wq = ctx.weights(We, r["ldwe"], T)
    c2, c3 = ctx.randn(C) * 0.2, ctx.randn(C) * 0.2
    G = ctx.randn(inp, inp)
    G = G @ G.t()
    sx = ctx.randn(inp)
    Wd = Wq.double()
    refM = (Wd.t() * c2.double()) @ Wd
    refv = c3.double() @ Wd
    refdw = c2.double().view(-1, 1) * (Wd @ G.double()) + c3.double().view(-1, 1) * sx.double().view(1, -1)
    outs = []
    for _ in range(REPEATS):
        mp = torch.zeros(pad(inp, 64), r["ldm"], dtype=T, device="cuda")
        mp[:inp, :inp] = float("nan")
        vb = torch.full((pad(inp, 8),), float("nan"), dtype=torch.float32, device="cuda")
        dwe = torch.full((C * inp,), 0.25, dtype=torch.float32, device="cuda")
        ops.xb_coeffs(ctx.cvec(c2), ctx.cvec(c3), wexp, G.contiguous(), sx, inp, C, mp, vb, dwe)
        outs.append((mp, vb, dwe))
    torch.cuda.synchronize()
    for i, (mp, vb, dwe) in enumerate(outs):
        check("M[%d]" % i, mp[:inp, :inp].float(), refM.float(), rtol=1.2e-2, afrac=1e-2)
        check("v[%d]" % i, vb[:inp], refv.float(), rtol=1e-3, afrac=1e-3)
        check("dwe[%d]" % i, dwe.view(C, inp) - 0.25, refdw.float(), rtol=1e-3, afrac=1e-3)


def run_bn_finalize_fwd(ctx, r):
    ops = ctx.ops
    C, rows, ld, count = r["C"], r["stat_rows"], r["stat_ld"], r["count"]
    assert not r["cmap"]
    eps = 1e-3
    mean_t, std_t = ctx.randn(C) * 0.5, ctx.rand(C) + 0.5
    # partial rows whose totals give the target mean / variance, spread unevenly over the rows
    wts = ctx.rand(rows, 1) + 0.1
    wts = wts / wts.sum()
    st = torch.full((rows, 2, ld), float("nan"), dtype=torch.float32, device="cuda")
    st[:, 0, :C] = wts * (mean_t * count)
    st[:, 1, :C] = wts * ((std_t ** 2 + mean_t ** 2) * count)
    if ld > C:
        st[:, :, C:] = 0.0
    gamma, beta = ctx.rand(C) + 0.5, ctx.randn(C) * 0.2
    tot = st[:, :, :C].double().sum(0)
    mean = tot[0] / count
    var = (tot[1] / count - mean * mean).clamp(min=0)
    invstd = 1.0 / torch.sqrt(var + eps)
    m = r["momentum"]
    rm0, rv0 = ctx.randn(C) * 0.1, ctx.rand(C) + 0.5
    Cp = pad(C, 8)
    for _ in range(REPEATS):
        rm, rv = rm0.clone(), rv0.clone()
        nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
        sc, sh, sm, si = (torch.full((Cp,), float("nan"), device="cuda") for _ in range(4))
        ops.bn_finalize_fwd(st, count, ctx.cvec(gamma), ctx.cvec(beta), eps, None if m < 0 else m, rm if r["running"] else None,
                            rv if r["running"] else None, nbt if r["running"] else None, sc, sh, sm, si, C, stat_rows=rows, stat_ld=ld)
        torch.cuda.synchronize()
        check("scale", sc[:C], (gamma.double() * invstd).float(), rtol=1e-4, afrac=1e-4)
        check("shift", sh[:C], (beta.double() - mean * gamma.double() * invstd).float(), rtol=1e-4, afrac=1e-4)
        check("mean", sm[:C], mean.float(), rtol=1e-4, afrac=1e-4)
        check("invstd", si[:C], invstd.float(), rtol=1e-4, afrac=1e-4)
        if r["running"]:
            mm = 1.0 if m < 0 else m
            check("running_mean", rm, ((1 - mm) * rm0.double() + mm * mean).float(), rtol=1e-4, afrac=1e-4)
            check("running_var", rv, ((1 - mm) * rv0.double() + mm * var * count / (count - 1)).float(), rtol=1e-4, afrac=1e-4)


def run_bn_finalize_bwd(ctx, r):
    ops = ctx.ops
    C, rows, ld, count = r["C"], r["stat_rows"], r["stat_ld"], r["count"]
    assert not r["cmap"]
    st = torch.full((rows, 2, ld), float("nan"), dtype=torch.float32, device="cuda")
    st[:, :, :C] = ctx.randn(rows, 2, C)
    if ld > C:
        st[:, :, C:] = 0.0
    gamma, mean, invstd = ctx.rand(C) + 0.5, ctx.randn(C) * 0.5, ctx.rand(C) + 0.5
    tot = st[:, :, :C].double().sum(0)
    g, mu, rr = gamma.double(), mean.double(), invstd.double()
    dg = rr * (tot[1] - mu * tot[0])
    db = tot[0]
    ref = dict(c1=g * rr, c2=-g * rr * rr * dg / count, c3=g * rr * (mu * rr * dg - db) / count, dgamma=dg + 0.25, dbeta=db + 0.25)
    Cp = pad(C, 8)
    for _ in range(REPEATS):
        c1, c2, c3 = (torch.full((Cp,), float("nan"), device="cuda") for _ in range(3))
        dgamma, dbeta = torch.full((C,), 0.25, device="cuda"), torch.full((C,), 0.25, device="cuda")
        ops.bn_finalize_bwd(st, count, ctx.cvec(gamma), ctx.cvec(mean), ctx.cvec(invstd), None, None, dgamma, dbeta, c1, c2, c3, C, stat_rows=rows,
                            stat_ld=ld)
        torch.cuda.synchronize()
        mag = float(st[:, :, :C].double().abs().sum(0).max())
        for name, got in (("c1", c1[:C]), ("c2", c2[:C]), ("c3", c3[:C]), ("dgamma", dgamma), ("dbeta", dbeta)):
            rf = ref[name].float()
            assert bool(torch.isfinite(got).all()), name
            # sums of `rows` zero-mean terms: the error budget is relative to the magnitude of the summed terms
            scale = {"c1": 1.0, "c2": mag / count * 4, "c3": mag / count * 4, "dgamma": mag * 2, "dbeta": mag}[name]
            assert float((got - rf).abs().max()) <= 1e-4 * float(rf.abs().max()) + 3e-6 * scale, (name, float((got - rf).abs().max()), scale)


def run_bn_apply(ctx, r):
    ops = ctx.ops
    M, C = r["M"], r["C"]
    T = ctx.T(r["dt"])
    xarg, xq = ctx.act_in(ctx.randn(M, C), r["x"], T)
    sc, sh = ctx.rand(C) + 0.5, ctx.randn(C) * 0.3
    ref = act_fwd(xq * sc + sh, r["act"])
    res = None
    if r["res"] is not None:
        res, rq = ctx.act_in(ctx.randn(M, C), r["res"], T)
        ref = ref + rq
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        y = ctx.act_out(M, C, r["y"], T)
        ops.bn_apply(xarg, ctx.cvec(sc), ctx.cvec(sh), r["act"], res, y, M, C)
        outs.append(y)
    torch.cuda.synchronize()
    del hold
    for i, y in enumerate(outs):
        check("y[%d]" % i, ctx.read(y, C), ref, rtol=1e-2, afrac=1e-3)


def run_bnbwd_apply(ctx, r):
    ops = ctx.ops
    M, C = r["M"], r["C"]
    T = ctx.T(r["dt"])
    garg, gq = ctx.act_in(ctx.randn(M, C), r["g"], T)
    xarg, xq = ctx.act_in(ctx.randn(M, C), r["x"], T)
    c1, c2, c3 = ctx.rand(C) + 0.5, ctx.randn(C) * 0.2, ctx.randn(C) * 0.2
    ref = c1 * gq + c2 * xq + c3
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        y = ctx.act_out(M, C, r["y"], T)
        ops.bnbwd_apply(garg, xarg, ctx.cvec(c1), ctx.cvec(c2), ctx.cvec(c3), y, M, C)
        outs.append(y)
    torch.cuda.synchronize()
    del hold
    for i, y in enumerate(outs):
        check("dP[%d]" % i, ctx.read(y, C), ref, rtol=1e-2, afrac=2e-3)


def run_act_bwd_stats(ctx, r):
    ops = ctx.ops
    M, C = r["M"], r["C"]
    T = ctx.T(r["dt"])
    dyarg, dq = ctx.act_in(ctx.randn(M, C), r["dy"], T)
    if r["masked"]:
        sc, sh = ctx.rand(C) + 0.5, ctx.randn(C) * 0.3
        zq = margin(ctx, M, C, sc, sh, T)
        ref = dq * act_grad(zq * sc + sh, r["act"])
        scv, shv = ctx.cvec(sc), ctx.cvec(sh)
    else:
        zq = ctx.randn(M, C).to(T).float()
        ref, scv, shv = dq, None, None
    zarg, _ = ctx.act_in(zq, r["z"], T)
    outs = []
    hold = busy()
    for _ in range(REPEATS):
        g = ctx.act_out(M, C, r["g"], T) if r["g"] is not None else None
        st = ctx.stats(r["stat_rows"], C)
        ops.act_bwd_stats(dyarg, zarg, scv, shv, r["act"], g, st, M, C, stat_rows=r["stat_rows"])
        outs.append((g, st))
    torch.cuda.synchronize()
    del hold
    for i, (g, st) in enumerate(outs):
        got = ctx.read(g, C) if g is not None else ref
        if g is not None:
            check("g[%d]" % i, got, ref, rtol=1e-2, afrac=1e-3)
        sm = stat_sum(st)
        check_sums("sum g[%d]" % i, sm[0], got)
        check_sums("sum g*z[%d]" % i, sm[1], got * zq)


def run_im2col_stem(ctx, r):
    ops = ctx.ops
    N, H, W = r["N"], r["H"], r["W"]
    T = ctx.T(r["dt"])
    img = ctx.randn(N, 3, H, W)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ref = F.unfold(img, 3, padding=1, stride=2)                      # [N, 27, Ho*Wo], column order ci*9 + ky*3 + kx
    ref = ref.permute(0, 2, 1).reshape(N * Ho * Wo, 27).to(T).float()
    for _ in range(REPEATS):
        col = torch.full((N * Ho * Wo, int(r["col"])), float("nan"), dtype=T, device="cuda")
        ops.im2col_stem(img, col, N, H, W)
        torch.cuda.synchronize()
        assert torch.equal(col[:, :27].float(), ref)
        assert float(col[:, 27:].float().abs().max()) == 0.0


def run_bn_act_pool(ctx, r):
    ops = ctx.ops
    N, HW, C = r["N"], r["HW"], r["C"]
    T = ctx.T(r["dt"])
    xarg, xq = ctx.act_in(ctx.randn(N * HW, C), r["x"], T)
    sc, sh = ctx.rand(C) + 0.5, ctx.randn(C) * 0.3
    mean = act_fwd(xq * sc + sh, r["act"]).view(N, HW, C).mean(1)
    p = r["drop_p"]
    step = torch.full((1,), 7, dtype=torch.int64, device="cuda")
    keeps = []
    for _ in range(REPEATS):
        pooled = torch.full((N, C), float("nan"), dtype=T, device="cuda")
        keep = torch.full((N, C), 3, dtype=torch.uint8, device="cuda") if r["keep"] else None
        ops.bn_act_pool(xarg, ctx.cvec(sc), ctx.cvec(sh), r["act"], pooled, keep, p, 1995, step, N, HW, C)
        torch.cuda.synchronize()
        if keep is not None:
            assert int((keep > 1).sum()) == 0
            frac = float(keep.float().mean())
            assert abs(frac - (1 - p)) < 0.02, frac
            ref = mean * keep.float() / (1 - p)
            keeps.append(keep)
        else:
            ref = mean
        check("pooled", pooled.float(), ref, rtol=1e-2, afrac=2e-3)
    assert all(torch.equal(keeps[0], k) for k in keeps[1:])   # same seed and step: same mask


def run_pool_act_bwd(ctx, r):
    ops = ctx.ops
    N, HW, C = r["N"], r["HW"], r["C"]
    T = ctx.T(r["dt"])
    sc, sh = ctx.rand(C) + 0.5, ctx.randn(C) * 0.3
    xq = margin(ctx, N * HW, C, sc, sh, T)
    xarg, _ = ctx.act_in(xq, r["x"], T)
    dp = ctx.randn(N, C).to(T)
    p = r["drop_p"]
    keep = (ctx.rand(N, C) >= p).to(torch.uint8) if r["keep"] else None
    gg = dp.float() / HW
    if keep is not None:
        gg = gg * keep.float() / (1 - p)
    ref = gg.view(N, 1, C) * act_grad(xq * sc + sh, r["act"]).view(N, HW, C)
    ref = ref.reshape(N * HW, C)
    for _ in range(REPEATS):
        g = torch.full((N * HW, C), float("nan"), dtype=T, device="cuda")
        st = ctx.stats(r["stat_rows"], C)
        ops.pool_act_bwd(dp, keep, p, xarg, ctx.cvec(sc), ctx.cvec(sh), r["act"], g, st, N, HW, C, stat_rows=r["stat_rows"])
        torch.cuda.synchronize()
        got = g.float()
        check("g", got, ref, rtol=1e-2, afrac=1e-3)
        sm = stat_sum(st)
        check_sums("sum g", sm[0], got)
        check_sums("sum g*x", sm[1], got * xq)


def run_colsum(ctx, r):
    ops = ctx.ops
    M, C = r["M"], r["C"]
    T = ctx.T(r["dt"])
    xarg, xq = ctx.act_in(ctx.randn(M, C), r["x"], T)
    out = torch.full((C,), 0.25, dtype=torch.float32, device="cuda")
    ops.colsum(xarg, out, M, C)
    torch.cuda.synchronize()
    ref = xq.double().sum(0).float()
    # colsum writes or accumulates: accept either convention consistently
    d0, d1 = float((out - ref).abs().max()), float((out - 0.25 - ref).abs().max())
    assert min(d0, d1) <= 1e-3 * float(ref.abs().max()) + 1e-3, (d0, d1)


def run_ce_smooth(ctx, r):
    ops = ctx.ops
    B, K = r["B"], r["K"]
    T = ctx.T(r["dt"])
    logits = torch.zeros(B, int(r["ldl"]), dtype=torch.float32, device="cuda")
    logits[:, :K] = ctx.randn(B, K) * 3
    target = torch.randint(0, K, (B,), device="cuda", generator=ctx.g)
    eps = 0.1
    lp = torch.log_softmax(logits[:, :K].double(), 1)
    tgt = torch.full((B, K), eps / K, dtype=torch.float64, device="cuda")
    tgt[torch.arange(B), target] += 1 - eps
    ref_loss = -(tgt * lp).sum(1)
    ref_dl = (lp.exp() - tgt) / B
    loss = torch.full((B,), float("nan"), device="cuda")
    dl = torch.full((B, int(r["dl"])), float("nan"), dtype=T, device="cuda")
    topk = torch.zeros(2, dtype=torch.int32, device="cuda")
    ops.ce_smooth(logits, target, eps, B, K, loss, dl, 1.0, topk)
    torch.cuda.synchronize()
    check("loss", loss, ref_loss.float(), rtol=1e-4, afrac=1e-4)
    check("dlogits", dl[:, :K].float(), ref_dl.float(), rtol=1e-2, afrac=1e-2)
    assert float(dl[:, K:].float().abs().max() if dl.shape[1] > K else 0) == 0.0
    top5 = logits[:, :K].topk(5, 1).indices
    assert int(topk[0]) == int((top5[:, 0] == target).sum()) and int(topk[1]) == int((top5 == target.view(-1, 1)).any(1).sum())


RUNNERS = dict(pw_gemm_nt=run_pw_gemm_nt, pw_gemm_tn=run_pw_gemm_tn, dwconv_fwd=run_dwconv_fwd, dwconv_bwd=run_dwconv_bwd,
               expand_bwd=run_expand_bwd, project_bwd=run_project_bwd, gram=run_gram, xb_coeffs=run_xb_coeffs,
               bn_finalize_fwd=run_bn_finalize_fwd, bn_finalize_bwd=run_bn_finalize_bwd, bn_apply=run_bn_apply, bnbwd_apply=run_bnbwd_apply,
               act_bwd_stats=run_act_bwd_stats, im2col_stem=run_im2col_stem, bn_act_pool=run_bn_act_pool, pool_act_bwd=run_pool_act_bwd,
               colsum=run_colsum, ce_smooth=run_ce_smooth)


def test_table_is_present_and_covers_the_step():
    """the table exists, names only entry points this file replays, and is as large as the step is varied (>= 100 distinct launches)"""
    assert ROWS, "tests/golden/bench_shapes.json is missing: run tools/make_bench_shapes.py on the GPU box"
    assert len(ROWS) >= 100, len(ROWS)
    unknown = sorted({r["entry"] for r in ROWS} - set(RUNNERS))
    assert not unknown, unknown
    nets = {n for r in ROWS for n in r["nets"]}
    assert {"atomnas_c_supernet", "atomnas_a_supernet"} <= nets


@pytest.mark.parametrize("row", ROWS, ids=[_id(r) for r in ROWS])
def test_bench_size_launch(gpu_lib, row):
    seed = zlib.crc32(json.dumps({k: v for k, v in row.items() if k != "nets"}, sort_keys=True).encode()) & 0x7FFFFFFF
    ctx = Ctx(seed)
    _CUR[0] = _id(row)
    RUNNERS[row["entry"]](ctx, row)
    torch.cuda.empty_cache()


def test_whole_step_intermediates_are_finite_and_variances_positive(gpu_lib):
    """One eager batch-256 training step of the AtomNAS-C supernet (bf16): every intermediate tensor the block / stem / tail executors
    produce is finite, every BatchNorm batch variance is > 0 (a saturated sum of squares shows as var = fmaxf(NaN, 0) = 0, which is
    how round 5's corrupted 7x7 stage hid), every gradient, parameter and optimizer quantity is finite after the step."""
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    from atomnas_amd import functional as Fn
    model, ts, hp, opt, ema, pinfo = bench.build("atomnas_c_supernet", torch.bfloat16, 256, 1995)
    ts.use_graph = False
    g = torch.Generator(device="cuda").manual_seed(7)
    ts.set_batch(torch.randn(256, 3, 224, 224, device="cuda", generator=g), torch.randint(0, 1000, (256,), device="cuda", generator=g))
    ts.step(rho=1e-5)
    Fn.STEP_TAP = tap = []
    try:
        ts.step(rho=1e-5)
    finally:
        Fn.STEP_TAP = None
    torch.cuda.synchronize()
    assert len(tap) > 300, len(tap)
    n_bn = 0
    eps = 1e-3
    for name, t in tap:
        p = t.to_plain() if hasattr(t, "to_plain") else t
        assert bool(torch.isfinite(p.float()).all()), "non-finite values in %s" % name
        if name.endswith(".invstd"):
            n_bn += 1
            var = 1.0 / p.double() ** 2 - eps
            live = p != 0     # padding channels of the fused hidden layout carry invstd = 0
            assert bool((var[live] > 1e-12).all()), "zero batch variance in %s" % name
    assert n_bn >= 60
    mgr = ts.mgr
    for nm in ("P", "G", "SQ", "BUF", "EMA", "S"):
        assert bool(torch.isfinite(getattr(mgr, nm)).all()), nm
    assert bool(torch.isfinite(ts.loss).all())
