"""Micro-benchmark of atomnas_expand_bwd in its one-stream form (e = NULL) on the supernet's early-stage shapes.

    python tools/xbbench.py            (ATOMNAS_XB_STREAM=0: the register-prefetch kernel k_expand_bwd)
Tensor sets rotate so that the Infinity Cache does not serve re-runs.  Experiments; not a test.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import ops  # noqa: E402
from atomnas_amd.ops import Slab  # noqa: E402

BF = torch.bfloat16


def pad(n, m):
    return (n + m - 1) // m * m


def bench(fn, n=10):
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


print("ATOMNAS_XB_STREAM", os.environ.get("ATOMNAS_XB_STREAM", "1"))
for (N, H, inp, hid) in [(256, 112, 16, 288), (256, 56, 24, 432)]:
    M = N * H * H
    sets = [(Slab.from_plain(torch.randn(M, hid, device="cuda").to(BF)), torch.randn(M, inp, device="cuda").to(BF),
             torch.randn(M, inp, device="cuda").to(BF), torch.empty(M, inp, dtype=BF, device="cuda")) for _ in range(3)]
    wt = torch.zeros(pad(inp, 64), pad(hid, 32), dtype=BF, device="cuda")
    wt[:inp, :hid] = (torch.randn(inp, hid, device="cuda") / inp ** 0.5).to(BF)
    mp = torch.zeros(pad(inp, 64), pad(inp, 32), dtype=BF, device="cuda")
    mp[:inp, :inp] = (torch.randn(inp, inp, device="cuda") * 0.1).to(BF)
    vb = torch.randn(pad(inp, 8), device="cuda")
    c1 = torch.rand(pad(hid, 8), device="cuda") + 0.5
    dwe = torch.zeros(hid, inp, device="cuda")
    ws = ops.expand_bwd_workspace(inp, hid, "cuda")
    cnt = [0]

    def run():
        h, x, add, gx = sets[cnt[0] % 3]
        cnt[0] += 1
        ops.expand_bwd(h, c1, x, wt, add, gx, dwe, M, inp, hid, ws=ws, mp=mp, vb=vb)

    t = bench(run)
    print("M%-8d inp%-3d hid%-4d: %.3f ms  (h at %4.0f GB/s)" % (M, inp, hid, t, M * hid * 2 / t / 1e6), flush=True)
