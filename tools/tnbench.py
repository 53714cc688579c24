"""Micro-benchmark of gemm_tn / dw / gemm_nt single shapes for PMC profiling (experiments; not a test)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from atomnas_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "tn"
p8 = lambda c: (c + 7) // 8 * 8
torch.manual_seed(0)
if which == "tn":
    M, NU, NV = 200704, 24, 432
    U = torch.randn(M, p8(NU), device="cuda").bfloat16(); V = torch.randn(M, p8(NV), device="cuda").bfloat16(); V2 = torch.randn(M, p8(NV), device="cuda").bfloat16()
    c = [torch.rand(p8(NV), device="cuda") for _ in range(3)]
    out = torch.zeros(NV, NU, device="cuda")
    fn = lambda: ops.gemm_tn(U, NU, V, NV, out, 1, NU, M, v_mode=ops.PRO_BNBWD, v2=V2, vc1=c[0], vc2=c[1], vc3=c[2])
elif which == "dwb":
    N, H, C, k, s = 64, 56, 144, 7, 1
    x = torch.randn(N * H * H, C, device="cuda").bfloat16(); y = torch.randn(N * H * H, C, device="cuda").bfloat16(); g = torch.randn(N * H * H, C, device="cuda").bfloat16()
    h = torch.zeros(N * H * H, C, device="cuda", dtype=torch.bfloat16); w = torch.randn(k * k, C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    c1, c2, c3 = torch.rand(C, device="cuda"), torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    st = torch.zeros(64 * 2 * C, device="cuda"); dw = torch.zeros(C * k * k, device="cuda")
    fn = lambda: ops.dwconv_bwd(g, y, c1, c2, c3, x, sc, sh, True, w, h, dw, st, C, N, H, H, C, k, s)
elif which == "nt":
    M, N, K = 200704, 24, 432
    A = torch.randn(M, p8(K), device="cuda").bfloat16()
    W = torch.zeros(64, 448, device="cuda", dtype=torch.bfloat16); W[:N, :K] = torch.randn(N, K) / K ** 0.5
    Cc = torch.zeros(M, p8(N), device="cuda", dtype=torch.bfloat16); st = torch.zeros(64 * 2 * N, device="cuda")
    sc = torch.rand(p8(K), device="cuda"); sh = torch.randn(p8(K), device="cuda")
    fn = lambda: ops.gemm_nt(A, W, Cc, M, N, K, a_mode=ops.PRO_BNRELU, ac1=sc, ac2=sh, a_relu=True, stats=st, stat_mode=ops.STAT_SQ)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): fn()
e1.record(); torch.cuda.synchronize()
print(which, "ms", e0.elapsed_time(e1) / 5)
