"""Micro-benchmark of the Squeeze-and-Excitation entry points on the block shapes of AtomNAS-C+ (BASELINE config 5), through the C ABI.

    python tools/sebench.py [N]
Per SE block: squeeze, dense layers forward, gating, gate backward (pool + dense backward + weight gradients), apply.
Tensor sets rotate so that the Infinity Cache does not serve re-runs.  Experiments; not a test.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import configs, ops  # noqa: E402
from atomnas_amd.ops import Slab  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
BF = torch.bfloat16
ITERS = int(os.environ.get("SEBENCH_ITERS", "10"))


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
    return e0.elapsed_time(e1) / n * 1e3   # us


def shapes():
    kw = configs.searched_kwparams("atomnas_c_plus")
    H = 112
    out = []
    inp = kw["input_channel"]
    for (oup, n, s, ks, chans, _) in [(r[0], r[1], r[2], r[3], r[4], r[5]) for r in kw["inverted_residual_setting"]]:
        Ho = (H - 1) // s + 1
        out.append((Ho, list(chans), inp))
        H, inp = Ho, oup
    return out


tot = dict(sq=0.0, mlp=0.0, sc=0.0, bg=0.0, ap=0.0)
print("N", N)
for (Ho, chans, inp) in shapes():
    hid = max(1, int(inp * 0.5))
    segs, o, st = [], 0, 0
    for h in chans:
        segs.append((o, st, h))
        o += (h + 15) // 16 * 16
        st += h
    HT, total = o, st
    cmap = torch.full((HT,), -1, dtype=torch.int32)
    for sg, s0, h in segs:
        cmap[sg:sg + h] = torch.arange(s0, s0 + h, dtype=torch.int32)
    cmap = cmap.cuda()
    HW = Ho * Ho
    M = N * HW
    nset = max(2, min(4, int(600e6 / (M * HT * 2 * 4)) + 1))
    mk = lambda: Slab.from_plain(torch.randn(M, HT, device="cuda").to(BF))
    sets = [(mk(), mk(), mk(), mk()) for _ in range(nset)]   # D, S, dS, g
    sc, sh = torch.rand(HT, device="cuda") + 0.5, torch.randn(HT, device="cuda")
    w1p, w2t = torch.randn(hid, HT, device="cuda") * 0.05, torch.randn(hid, HT, device="cuda") * 0.05
    b1, b2p = torch.randn(hid, device="cuda"), torch.randn(HT, device="cuda")
    pooled, gate, dz2, dpooled = (torch.zeros(N, HT, device="cuda") for _ in range(4))
    parts = ops.se_pool_parts(N, HW, HT)
    pparts, dgate = torch.zeros(parts, N, HT, device="cuda"), torch.zeros(parts, N, HT, device="cuda")
    hpre, dz1 = torch.zeros(N, hid, device="cuda"), torch.zeros(N, hid, device="cuda")
    dw1, db1, dw2, db2 = torch.zeros(hid * total, device="cuda"), torch.zeros(hid, device="cuda"), torch.zeros(total * hid, device="cuda"), torch.zeros(total, device="cuda")
    rows = ops.stat_rows_for(HT)
    st2 = torch.empty(rows * 2 * HT, device="cuda")
    cnt = [0]

    def nxt():
        cnt[0] += 1
        return sets[cnt[0] % nset]

    def f_sq():
        D, S, dS, g = nxt()
        ops.se_squeeze(D, sc, sh, 3, pparts, N, HW, HT)

    def f_mlp():
        ops.se_mlp_fwd(pparts, pooled, cmap, w1p, b1, w2t, b2p, 3, hpre, gate, N, HT, hid)

    def f_sc():
        D, S, dS, g = nxt()
        ops.se_scale(D, sc, sh, 3, gate, S, M, HW, HT)

    def f_bg():
        D, S, dS, g = nxt()
        ops.se_bwd_gate(dS, D, sc, sh, 3, gate, pooled, cmap, w1p, w2t, hpre, dgate, dz2, dz1, dpooled, dw1, db1, dw2, db2, N, HW, HT, total, hid, se_act=3)

    def f_ap():
        D, S, dS, g = nxt()
        ops.se_bwd_apply(dS, D, sc, sh, 3, gate, dpooled, g, st2, M, HW, HT, stat_rows=rows)

    t = dict(sq=bench(f_sq), mlp=bench(f_mlp), sc=bench(f_sc), bg=bench(f_bg), ap=bench(f_ap))
    for k in tot:
        tot[k] += t[k]
    nb = M * HT * 2 / 1e3   # KB of one hidden tensor
    print("H%-3d HT%-5d hid%-3d: squeeze %6.1f us (%4.0f GB/s)  dense fwd %6.1f  gating %6.1f (%4.0f GB/s)  gate bwd %6.1f  apply %6.1f (%4.0f GB/s)"
          % (Ho, HT, hid, t["sq"], nb / t["sq"], t["mlp"], t["sc"], 2 * nb / t["sc"], t["bg"], t["ap"], 3 * nb / t["ap"]), flush=True)
    del sets
print("sums (ms): squeeze %.3f  dense fwd %.3f  gating %.3f  gate bwd %.3f  apply %.3f   total %.3f"
      % tuple([tot[k] / 1e3 for k in ("sq", "mlp", "sc", "bg", "ap")] + [sum(tot.values()) / 1e3]))
