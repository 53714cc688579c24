"""Experiment: the three depthwise launches of a block work on channel SEGMENTS of fused [M, HT] tensors (row pitch 864 B, 288 B per
segment).  How much does that cost against dense [M, 144] tensors, and do three streams help?  (bs 256, 56x56, cold tensors.)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from atomnas_amd import ops
N, H, C, s = 256, 56, 144, 1
M = N * H * H
def mk(c):
    return [torch.randn(M, c, device="cuda").bfloat16() for _ in range(3)] + [torch.zeros(M, c, device="cuda", dtype=torch.bfloat16)]
dense = [[mk(C) for _ in range(3)] for _ in range(2)]      # [set][branch] -> x, y, g, h
fused = [mk(3 * C) for _ in range(2)]                      # [set] -> x, y, g, h of width 432
ks = [3, 5, 7]
w = [torch.randn(k * k, C, device="cuda") for k in ks]
sc = torch.rand(3 * C, device="cuda") + 0.5; sh = torch.randn(3 * C, device="cuda")
c1, c2, c3 = torch.rand(3 * C, device="cuda"), torch.randn(3 * C, device="cuda") * 0.1, torch.randn(3 * C, device="cuda") * 0.1
st = [torch.zeros(ops.STAT_ROWS * 2 * 3 * C, device="cuda") for _ in range(3)]; dw = [torch.zeros(C * k * k, device="cuda") for k in ks]
it = [0]
def run_dense():
    S = dense[it[0] % 2]
    for b, k in enumerate(ks):
        x, y, g, h = S[b]
        ops.dwconv_bwd(g, y, c1, c2, c3, x, sc, sh, True, w[b], h, dw[b], st[b], 3 * C, N, H, H, C, k, s)
def run_fused(streams=None):
    x, y, g, h = fused[it[0] % 2]
    for b, k in enumerate(ks):
        o = b * C
        def go():
            ops.dwconv_bwd(g[:, o:], y[:, o:], c1[o:], c2[o:], c3[o:], x[:, o:], sc[o:], sh[o:], True, w[b], h[:, o:], dw[b], st[b][o:], 3 * C, N, H, H, C, k, s)
        if streams is None: go()
        else:
            streams[b].wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(streams[b]): go()
    if streams is not None:
        for q in streams: torch.cuda.current_stream().wait_stream(q)
def timeit(fn, n=6):
    for _ in range(2): fn(); it[0] += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn(); it[0] += 1
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("bwd k3+k5+k7 dense tensors      %.3f ms" % timeit(run_dense))
print("bwd k3+k5+k7 fused segments     %.3f ms" % timeit(run_fused))
ss = [torch.cuda.Stream() for _ in range(3)]
print("bwd fused segments, 3 streams   %.3f ms" % timeit(lambda: run_fused(ss)))
# forward
yd = None
def fwd_dense():
    S = dense[it[0] % 2]
    for b, k in enumerate(ks):
        x, y, g, h = S[b]
        ops.dwconv_fwd(x, sc, sh, True, w[b], y, st[b], 3 * C, N, H, H, C, k, s)
def fwd_fused():
    x, y, g, h = fused[it[0] % 2]
    for b, k in enumerate(ks):
        o = b * C
        ops.dwconv_fwd(x[:, o:], sc[o:], sh[o:], True, w[b], y[:, o:], st[b][o:], 3 * C, N, H, H, C, k, s)
print("fwd k3+k5+k7 dense tensors      %.3f ms" % timeit(fwd_dense))
print("fwd k3+k5+k7 fused segments     %.3f ms" % timeit(fwd_fused))
