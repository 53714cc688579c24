"""Micro-benchmark of gemm_nt shapes through the C ABI (experiments; not a test)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from atomnas_amd import ops
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
p8 = lambda c: (c + 7) // 8 * 8
for (M, N, K) in [(200704, 432, 24), (802816, 288, 16), (50176, 720, 40), (200704, 24, 432), (3136, 192, 3456), (12544, 96, 1728)]:
    A = torch.randn(M, p8(K), device="cuda").bfloat16()
    W = torch.zeros((N + 63) // 64 * 64, (K + 31) // 32 * 32, device="cuda", dtype=torch.bfloat16); W[:N, :K] = torch.randn(N, K) / K ** 0.5
    C = torch.zeros(M, p8(N), device="cuda", dtype=torch.bfloat16)
    Z = torch.randn(M, p8(N), device="cuda").bfloat16()
    st = torch.zeros(2 * N, device="cuda")
    zs = torch.ones(p8(N), device="cuda"); zh = torch.zeros(p8(N), device="cuda")
    t0 = bench(lambda: ops.gemm_nt(A, W, C, M, N, K))
    t1 = bench(lambda: ops.gemm_nt(A, W, C, M, N, K, stats=st, stat_mode=ops.STAT_SQ))
    t2 = bench(lambda: ops.gemm_nt(A, W, C, M, N, K, z=Z, zscale=zs, zshift=zh, mask=True, stats=st, stat_mode=ops.STAT_Z))
    by = (M * K + M * N) * 2
    print("M%d N%d K%d: plain %.3f ms (%.0f GB/s) | +stats %.3f | +mask+statz %.3f" % (M, N, K, t0, by / t0 / 1e6, t1, t2))
