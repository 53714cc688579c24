"""Micro-benchmark of the depthwise kernels through the C ABI (experiments; not a test)."""
import ctypes, sys, os, torch
sys.path.insert(0, "/root/repo")
from atomnas_amd import _lib
libpath = sys.argv[1] if len(sys.argv) > 1 else _lib.LIB_PATH
_lib.LIB_PATH = libpath
from atomnas_amd import ops
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
N = 64
for (H, C, k, s) in [(56, 144, 3, 1), (56, 144, 7, 1), (14, 480, 7, 1), (112, 96, 7, 2), (7, 1152, 5, 1)]:
    Ho = (H - 1) // s + 1
    x = torch.randn(N * H * H, C, device="cuda").bfloat16()
    y = torch.zeros(N * Ho * Ho, C, device="cuda", dtype=torch.bfloat16)
    g = torch.randn(N * Ho * Ho, C, device="cuda").bfloat16()
    h = torch.zeros(N * H * H, C, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(k * k, C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    c1, c2, c3 = torch.rand(C, device="cuda"), torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    st = torch.zeros(2 * C, device="cuda"); dw = torch.zeros(C * k * k, device="cuda")
    tf = bench(lambda: ops.dwconv_fwd(x, sc, sh, True, w, y, st, C, N, H, H, C, k, s))
    tb = bench(lambda: ops.dwconv_bwd(g, y, c1, c2, c3, x, sc, sh, True, w, h, dw, st, C, N, H, H, C, k, s))
    bf = (x.numel() + y.numel()) * 2; bb = (2 * x.numel() + y.numel()) * 2
    print("H%d C%d k%d s%d: fwd %.3f ms (%.0f GB/s)  bwd %.3f ms (%.0f GB/s algorithmic)" % (H, C, k, s, tf, bf / tf / 1e6, tb, bb / tb / 1e6))
