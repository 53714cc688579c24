"""Prototype check + timing of the late-stage wide-input GEMM on the LDS-DMA queue (csrc/experimental/nt_swg.hip, NOT in the product
library) against the product's LDS-weights kernel (atomnas_pw_gemm_nt) on the supernet's projection shapes.

    tools/build_ntswg_experiment.sh && ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libntswg.so python tools/experiments/ntswg_bench.py
Results agree to the rounding of the bf16 output (other summation order); statistics to 1e-3 relative.
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from atomnas_amd import _lib, ops  # noqa: E402
from atomnas_amd.ops import Slab, _p, _stream  # noqa: E402

BF = torch.bfloat16
lib = _lib.load()
if not hasattr(lib, "atomnas_exp_nt_swg"):
    raise SystemExit("load the prototype library: tools/build_ntswg_experiment.sh; ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libntswg.so")
vp, i32, i64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lib.atomnas_exp_nt_swg.argtypes = [vp, i64, vp, vp, i32, vp, i32, vp, i32, vp, i32, i64, i32, i32, vp]
lib.atomnas_exp_nt_swg.restype = i32


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
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K) in [(200704, 40, 720), (50176, 80, 1440), (50176, 96, 1728), (12544, 192, 3456), (50000, 96, 1000)]:
    torch.manual_seed(M + N + K)
    sets = [Slab.from_plain(torch.randn(M, pad(K, 16), device="cuda").to(BF), K) for _ in range(3)]
    Wp = torch.zeros(pad(N, 64), pad(K, 32), dtype=BF, device="cuda")
    Wp[:N, :K] = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF)
    sc, sh = torch.rand(pad(K, 8), device="cuda") + 0.5, torch.randn(pad(K, 8), device="cuda") * 0.3
    rows = ops.stat_rows_for(N)
    C0, C1 = torch.zeros(M, N, dtype=BF, device="cuda"), torch.zeros(M, N, dtype=BF, device="cuda")
    st0, st1 = torch.empty(rows, 2, N, device="cuda"), torch.empty(rows, 2, N, device="cuda")
    cnt = [0]

    def ref():
        cnt[0] += 1
        ops.gemm_nt(sets[cnt[0] % 3], Wp, C0, M, N, K, a_mode=ops.PRO_BNRELU, ac1=sc, ac2=sh, a_relu=1, stats=st0, stat_mode=ops.STAT_SQ, stat_rows=rows)

    def new():
        cnt[0] += 1
        a = sets[cnt[0] % 3]
        rc = lib.atomnas_exp_nt_swg(_p(a), a.ss, _p(sc), _p(sh), 1, _p(Wp), Wp.stride(0), _p(C1), C1.stride(0), _p(st1), rows, M, N, K, _stream())
        if rc:
            raise RuntimeError(lib.atomnas_last_error().decode())

    cnt[0] = 0
    ref()
    cnt[0] = 0
    new()
    torch.cuda.synchronize()
    d = (C0.float() - C1.float()).abs()
    scale = float(C0.float().abs().max())
    s0, s1 = st0.sum(0), st1.sum(0)
    ds = float(((s0 - s1).abs() / (s0.abs() + 1e-3 * s0.abs().max())).max())
    ok = float(d.max()) <= 2e-2 * max(1.0, scale) and ds < 2e-2
    t0, t1 = bench(ref), bench(new)
    print("waves %s  M%-7d N%-4d K%-5d: product %6.1f us (%4.0f GB/s)   prototype %6.1f us (%4.0f GB/s)   max |diff| %.3g of %.3g, stats rel %.2g  %s"
          % (os.environ.get("ATOMNAS_SWG_WAVES", "4"), M, N, K, t0, M * K * 2 / t0 / 1e3, t1, M * K * 2 / t1 / 1e3, float(d.max()), scale, ds, "OK" if ok else "MISMATCH"), flush=True)
