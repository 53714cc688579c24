"""Full-size check of the expand-forward GEMM (atomnas_pw_gemm_nt, slab-major output, statistics) against torch on the GPU.
    python tools/gemmcheck.py     shapes: GEMMCHECK="12544,3456,192;50176,1728,96" (M,N,K)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import ops
from atomnas_amd.ops import Slab
SH = [(12544, 3456, 192), (12544, 1280, 320), (50176, 1728, 96), (50176, 1440, 80), (200704, 720, 40), (802816, 432, 24), (3211264, 288, 16)]
if os.environ.get("GEMMCHECK"):
    SH = [tuple(int(v) for v in c.split(",")) for c in os.environ["GEMMCHECK"].split(";")]
torch.manual_seed(0)
for (M, N, K) in SH:
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5)
    Wp = torch.zeros((N + 63) // 64 * 64, (K + 31) // 32 * 32, dtype=torch.bfloat16, device="cuda"); Wp[:N, :K] = W.bfloat16()
    ref = A.float() @ Wp[:N, :K].float().t()
    for rep in range(3):
        rows = ops.stat_rows_for(N)
        for slab in (True, False):
            C = Slab(M, N, torch.bfloat16, "cuda") if slab else torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
            (C.t if slab else C).fill_(float("nan"))
            st = torch.full((rows, 2, N), float("nan"), device="cuda")
            ops.gemm_nt(A, Wp, C, M, N, K, stats=st, stat_mode=ops.STAT_SQ, stat_rows=rows)
            torch.cuda.synchronize()
            Cp = (C.to_plain() if slab else C)[:, :N].float()
            bad = ~((Cp - ref).abs() <= 0.02 * ref.abs() + 0.05)
            msg = "M%d N%d K%d %s rep %d: bad %d" % (M, N, K, "slab " if slab else "plain", rep, int(bad.sum()))
            if bad.any():
                r_ = torch.nonzero(bad.any(1)).flatten(); c_ = torch.nonzero(bad.any(0)).flatten()
                msg += " rows %s (n=%d) cols %s (n=%d) e.g. got %.4g ref %.4g" % (r_[:6].tolist(), r_.numel(), c_[:8].tolist(), c_.numel(),
                                                                                  float(Cp[bad][0]), float(ref[bad][0]))
            s = st.sum(0)
            msg += " | stat err %.2e / %.2e" % (float((s[0] - Cp.sum(0)).abs().max() / Cp.sum(0).abs().max()), float((s[1] - (Cp * Cp).sum(0)).abs().max() / (Cp * Cp).sum(0).abs().max()))
            print(msg, flush=True)
    del ref
