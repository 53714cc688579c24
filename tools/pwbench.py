"""Micro-benchmark of the pointwise GEMM entry points through the C ABI on the step's shapes (experiments; not a test).

    python tools/pwbench.py [lib.so] [expand|mask|project|dgrad|all] [N]
expand: gemm_nt of the expand forward (x[M,inp] -> E[M,hid] slab-major, sum / sum-of-squares statistics)
mask  : gemm_nt of the projection input gradient through the depthwise activation (dP[M,oup] -> g[M,hid], z = D, statistics sum g, sum g*z)
project: gemm_nt of the projection forward (act(bn(D))[M,hid] -> P[M,oup], sum / sum-of-squares statistics)
dgrad : gemm_nt of the expand input gradient (BatchNorm-backward prologue over h, E [M,hid] -> Gx[M,inp], + residual gradient)
Tensor sets rotate so that the 256 MiB Infinity Cache does not serve re-runs.  GB/s counts the wide tensors only (written / read once).
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import _lib
args = sys.argv[1:]
if args and args[0].endswith(".so"):
    _lib.LIB_PATH = args.pop(0)
from atomnas_amd import ops
from atomnas_amd.ops import Slab, STAT_SQ, STAT_Z
which = args[0] if args else "all"
N = int(args[1]) if len(args) > 1 else 256
_lib.load()
NSET = int(os.environ.get("PWBENCH_SETS", "3"))
ITERS = int(os.environ.get("PWBENCH_ITERS", "12"))


WARM = int(os.environ.get("PWBENCH_WARMX", "0"))   # 1: the narrow operand is rewritten right before every call (as the producer of
                                                    # the block input does in the training step): it is then in the Infinity Cache


def bench(fn, n, pre=None):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in ev:
        if pre is not None: pre()
        e0.record(); fn(); e1.record()
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in ev) / n


def pack_w(w):
    n, k = w.shape
    buf = torch.zeros((n + 63) // 64 * 64, (k + 31) // 32 * 32, dtype=torch.bfloat16, device="cuda")
    buf[:n, :k] = w.bfloat16()
    return buf


# (H, narrow channels, hidden channels): the AtomNAS-C supernet stages at 224x224
CASES = [(112, 16, 288), (56, 24, 432), (28, 40, 720), (14, 80, 1440), (14, 96, 1728), (7, 192, 3456)]
if os.environ.get("PWBENCH_CASES"):
    CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ["PWBENCH_CASES"].split(";")]
print(os.path.basename(_lib.LIB_PATH), "N", N)
tot = {"expand": 0.0, "mask": 0.0}
for (H, inp, hid) in CASES:
    M = N * H * H
    nset = NSET if M * hid * 2 < (1 << 30) else 2
    x = [torch.randn(M, inp, device="cuda").bfloat16() for _ in range(nset)]
    wide = [Slab(M, hid, torch.bfloat16, "cuda") for _ in range(nset)]
    W = pack_w(torch.randn(hid, inp, device="cuda") / inp ** 0.5)
    rows = ops.stat_rows_for(hid)
    st = torch.empty(rows * 2 * hid, device="cuda")
    cnt = [0]
    line = "H%-3d M%-8d %4d -> %-4d:" % (H, M, inp, hid)
    if which in ("expand", "all"):
        def f():
            i = cnt[0] % nset; cnt[0] += 1
            ops.gemm_nt(x[i], W, wide[i], M, hid, inp, stats=st, stat_mode=STAT_SQ, stat_rows=rows)
        xs = torch.randn(M, inp, device="cuda").bfloat16()
        t = bench(f, ITERS, (lambda: x[cnt[0] % nset].copy_(xs)) if WARM else None); tot["expand"] += t
        line += "  expand %.3f ms %5.0f GB/s" % (t, M * hid * 2 / t / 1e6)
    if which in ("mask", "all"):
        z = [Slab.from_plain(torch.randn(M, hid, device="cuda").bfloat16()) for _ in range(nset)]
        zs, zh = torch.rand(hid, device="cuda") + 0.5, torch.randn(hid, device="cuda") * 0.3
        def f():
            i = cnt[0] % nset; cnt[0] += 1
            ops.gemm_nt(x[i], W, wide[i], M, hid, inp, z=z[i], zscale=zs, zshift=zh, mask=1, stats=st, stat_mode=STAT_Z, stat_rows=rows)
        t = bench(f, ITERS); tot["mask"] += t
        line += "  mask %.3f ms %5.0f GB/s" % (t, 2 * M * hid * 2 / t / 1e6)
        del z
    if which in ("project", "dgrad", "all"):
        from atomnas_amd.ops import PRO_BNRELU, PRO_BNBWD
        a = [Slab.from_plain(torch.randn(M, hid, device="cuda").bfloat16()) for _ in range(nset)]
        WT = pack_w(torch.randn(inp, hid, device="cuda") / hid ** 0.5)
        out = torch.empty(M, inp, dtype=torch.bfloat16, device="cuda")
        c1, c2, c3 = torch.rand(hid, device="cuda") + 0.5, torch.randn(hid, device="cuda") * 0.2, torch.randn(hid, device="cuda") * 0.2
        rows2 = ops.stat_rows_for(inp)
        st2 = torch.empty(rows2 * 2 * inp, device="cuda")
        if which in ("project", "all"):
            def f():
                i = cnt[0] % nset; cnt[0] += 1
                ops.gemm_nt(a[i], WT, out, M, inp, hid, a_mode=PRO_BNRELU, ac1=c1, ac2=c2, a_relu=1, stats=st2, stat_mode=STAT_SQ, stat_rows=rows2)
            t = bench(f, ITERS); tot["project"] = tot.get("project", 0.0) + t
            line += "  project %.3f ms %5.0f GB/s" % (t, M * hid * 2 / t / 1e6)
        if which in ("dgrad", "all"):
            res = torch.randn(M, inp, device="cuda").bfloat16()
            def f():
                i = cnt[0] % nset; cnt[0] += 1
                ops.gemm_nt(a[i], WT, out, M, inp, hid, a_mode=PRO_BNBWD, a2=wide[i], ac1=c1, ac2=c2, ac3=c3, add=res)
            t = bench(f, ITERS); tot["dgrad"] = tot.get("dgrad", 0.0) + t
            line += "  dgrad %.3f ms %5.0f GB/s" % (t, 2 * M * hid * 2 / t / 1e6)
        del a
    print(line, flush=True)
    del x, wide
print("sum " + "  ".join("%s %.3f ms" % kv for kv in tot.items()))
