"""Does the depthwise forward run faster when its input was written right before it (256 MB Infinity Cache)?  Experiment for a per-branch
interleave of the expand GEMM and the depthwise launches.    python tools/mallprobe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import ops, _lib
from atomnas_amd.ops import Slab
_lib.load()
N = 256
for (H, C, k) in [(56, 144, 3), (56, 144, 5), (56, 144, 7), (28, 240, 3), (28, 240, 5), (112, 96 // 3, 3)]:
    M = N * H * H
    sets = [(Slab.from_plain(torch.randn(M, C, device="cuda").bfloat16()), Slab.from_plain(torch.zeros(M, C, device="cuda").bfloat16())) for _ in range(3)]
    w = torch.randn(k * k, C, device="cuda"); sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    rows = ops.stat_rows_for(C); st = torch.empty(rows * 2 * C, device="cuda")
    def run(i): 
        x, y = sets[i % 3]
        ops.dwconv_fwd(x, sc, sh, True, w, y, st, C, N, H, H, C, k, 1, stat_rows=rows)
    res = {}
    for mode in ("cold", "warm"):
        ts = []
        for it in range(12):
            x, y = sets[it % 3]
            if mode == "warm":
                x.t.mul_(1.0)   # read-modify-write of the input: leaves it in the cache the way a producer would (plus its own read)
                x.t.fill_(0.25) if it % 2 else x.t.fill_(0.5)   # write-only pass over the input right before the launch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(it); e1.record(); torch.cuda.synchronize()
            if it >= 3: ts.append(e0.elapsed_time(e1))
        res[mode] = sum(ts) / len(ts)
    print("H%-3d C%-4d k%d  input %4d MB  cold %.3f ms  input just written %.3f ms" % (H, C, k, M * C * 2 >> 20, res["cold"], res["warm"]), flush=True)
    del sets
