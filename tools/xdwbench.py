"""Micro-benchmark of the E-elimination kernels (csrc/xdw.hip) against the stand-alone depthwise kernels, through the C ABI.

    python tools/xdwbench.py [N]
Per branch shape of the AtomNAS supernet's 56x56 and 28x28 stages: depthwise forward / backward reading E from HBM
(atomnas_dwconv_fwd / _bwd) next to the fused forms that recompute it from the block input (atomnas_xdw_fwd / _bwd), plus the
expand GEMM the fused forward retires and the Gram-statistics launches it adds.  Tensor sets rotate so that the 256 MiB Infinity
Cache does not serve re-runs.  Experiments; not a test.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import ops  # noqa: E402
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))
import xdw_ops  # noqa: E402
if not xdw_ops.available():
    raise SystemExit("load the experiment library: tools/build_xdw_experiment.sh; ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libxdw.so")
from atomnas_amd.ops import Slab  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BF = torch.bfloat16
NSET = int(os.environ.get("DWBENCH_SETS", "3"))
ITERS = int(os.environ.get("DWBENCH_ITERS", "10"))


def bench(fn, n=ITERS):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def pad(n, m):
    return (n + m - 1) // m * m


CASES = [(56, 24, 144, 3), (56, 24, 144, 5), (56, 24, 144, 7), (28, 40, 240, 3), (28, 40, 240, 5), (28, 40, 240, 7)]
if os.environ.get("XDWBENCH_CASES"):
    CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ["XDWBENCH_CASES"].split(";")]
print("N", N, "S", os.environ.get("ATOMNAS_XDW_S", "3"))
tot = dict(f0=0.0, f1=0.0, b0=0.0, b1=0.0)
for (H, inp, C, k) in CASES:
    M = N * H * H
    if not xdw_ops.xdw_supported(N, H, H, inp, C, k, 1, BF):
        print("H%d inp%d C%d k%d: no instance" % (H, inp, C, k))
        continue
    mk = lambda: Slab.from_plain(torch.randn(M, C, device="cuda").to(BF))
    sets = [(torch.randn(M, inp, device="cuda").to(BF), mk(), mk(), mk(), mk()) for _ in range(NSET)]   # x, E, D, g, h
    wexp = torch.zeros(pad(C, 64), pad(inp, 32), dtype=BF, device="cuda")
    wexp[:C, :inp] = (torch.randn(C, inp, device="cuda") / inp ** 0.5).to(BF)
    w = torch.randn(k * k, C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5
    sh = torch.randn(C, device="cuda")
    c1, c2, c3 = torch.rand(C, device="cuda"), torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    rows = ops.stat_rows_for(C)
    st = torch.empty(rows * 2 * C, device="cuda")
    dw = torch.zeros(C * k * k, device="cuda")
    ws = torch.empty(rows * C * k * k, device="cuda")
    cnt = [0]

    def nxt():
        s = sets[cnt[0] % NSET]
        cnt[0] += 1
        return s

    def expand():
        x, E, D, g, h = nxt()
        ops.gemm_nt(x, wexp, E, M, C, inp, stats=st, stat_mode=ops.STAT_SQ, stat_rows=rows)

    def fwd0():
        x, E, D, g, h = nxt()
        ops.dwconv_fwd(E, sc, sh, True, w, D, st, C, N, H, H, C, k, 1, stat_rows=rows)

    def fwd1():
        x, E, D, g, h = nxt()
        xdw_ops.xdw_fwd(x, inp, wexp, sc, sh, 1, w, D, st, C, N, H, H, C, k, stat_rows=rows)

    def bwd0():
        x, E, D, g, h = nxt()
        ops.dwconv_bwd(g, D, c1, c2, c3, E, sc, sh, True, w, h, dw, st, C, N, H, H, C, k, 1, stat_rows=rows, dw_ws=ws)

    def bwd1():
        x, E, D, g, h = nxt()
        xdw_ops.xdw_bwd(g, D, c1, c2, c3, x, inp, wexp, sc, sh, 1, w, h, dw, st, C, N, H, H, C, k, stat_rows=rows, dw_ws=ws)

    te, f0, f1, b0, b1 = bench(expand), bench(fwd0), bench(fwd1), bench(bwd0), bench(bwd1)
    tot["f0"] += f0 + te; tot["f1"] += f1; tot["b0"] += b0; tot["b1"] += b1
    nb = M * C * 2
    print("H%-3d inp%-3d C%-4d k%d:  expand %.3f  fwd %.3f -> fused %.3f ms (D write at %4.0f GB/s)   bwd %.3f -> fused %.3f ms (%4.0f GB/s on g,D,h)"
          % (H, inp, C, k, te, f0, f1, nb / f1 / 1e6, b0, b1, 3 * nb / b1 / 1e6), flush=True)
    del sets
# Gram statistics of the two stages
for (H, inp, HT) in [(56, 24, 432), (28, 40, 720)]:
    M = N * H * H
    x = torch.randn(M, inp, device="cuda").to(BF)
    wexp = torch.zeros(pad(HT, 64), pad(inp, 32), dtype=BF, device="cuda")
    gram = torch.zeros(inp * inp, device="cuda")
    stats = torch.empty(2 * HT, device="cuda")
    sx = torch.empty(inp, device="cuda")
    gws = torch.empty(512 * (inp * inp + inp), device="cuda")

    def gram_path():
        ops.gram(x, M, inp, gram, sx, ws=gws)
        xdw_ops.gram_stats(gram, sx, wexp, inp, HT, stats, HT)

    print("H%-3d inp%-3d HT%-4d: Gram statistics (partials + reduce + quadratic forms) %.3f ms" % (H, inp, HT, bench(gram_path)))
print("sum: expand + fwd %.3f -> %.3f ms;  bwd %.3f -> %.3f ms" % (tot["f0"], tot["f1"], tot["b0"], tot["b1"]))
