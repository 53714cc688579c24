"""Experiment: do two of the step's kernels overlap usefully when issued on two streams?  (dw backward k=7 56x56x144 and the
weight-gradient GEMM M=802816 NU=24 NV=432; cold tensors.)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from atomnas_amd import ops
N, H, C, k, s = 256, 56, 144, 7, 1
p8 = lambda c: (c + 7) // 8 * 8
def mk():
    x = torch.randn(N * H * H, C, device="cuda").bfloat16(); y = torch.randn(N * H * H, C, device="cuda").bfloat16()
    g = torch.randn(N * H * H, C, device="cuda").bfloat16(); h = torch.zeros(N * H * H, C, device="cuda", dtype=torch.bfloat16)
    return x, y, g, h
sets = [mk() for _ in range(2)]
w = torch.randn(k * k, C, device="cuda"); sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
c1, c2, c3 = torch.rand(C, device="cuda"), torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
st = torch.zeros(ops.STAT_ROWS * 2 * C, device="cuda"); dw = torch.zeros(C * k * k, device="cuda")
M, NU, NV = 802816, 24, 432
tsets = [(torch.randn(M, p8(NU), device="cuda").bfloat16(), torch.randn(M, p8(NV), device="cuda").bfloat16(), torch.randn(M, p8(NV), device="cuda").bfloat16()) for _ in range(2)]
cv = [torch.rand(p8(NV), device="cuda") for _ in range(3)]; out = torch.zeros(NV, NU, device="cuda")
i = [0]
def dwb():
    x, y, g, h = sets[i[0] % 2]
    ops.dwconv_bwd(g, y, c1, c2, c3, x, sc, sh, True, w, h, dw, st, C, N, H, H, C, k, s)
def tn():
    U, V, V2 = tsets[i[0] % 2]
    ops.gemm_tn(U, NU, V, NV, out, 1, NU, M, v_mode=ops.PRO_BNBWD, v2=V2, vc1=cv[0], vc2=cv[1], vc3=cv[2])
def timeit(fn, n=6):
    for _ in range(2): fn(); i[0] += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(); i[0] += 1
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
side = torch.cuda.Stream()
def both_serial(): dwb(); tn()
def both_par():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): tn()
    dwb()
    torch.cuda.current_stream().wait_stream(side)
def both_par_lowprio_factory():
    lo = torch.cuda.Stream(priority=0); 
    return lo
print("dw_bwd alone %.3f ms" % timeit(dwb)); print("tn alone %.3f ms" % timeit(tn))
print("serial %.3f ms" % timeit(both_serial)); print("two streams %.3f ms" % timeit(both_par))
hi = torch.cuda.Stream(priority=-1)
def both_prio():
    side.wait_stream(torch.cuda.current_stream()); hi.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side): tn()
    with torch.cuda.stream(hi): dwb()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.current_stream().wait_stream(hi)
print("two streams, dw high priority %.3f ms" % timeit(both_prio))
