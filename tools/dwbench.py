"""Micro-benchmark of the depthwise kernels through the C ABI (experiments; not a test).

    python tools/dwbench.py [lib.so] [fwd|bwd|both] [N] [plain|slab]
CASES can be narrowed with DWBENCH_CASES="56,144,3,1;56,144,7,1" (H,C,k,s).
"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import _lib
args = sys.argv[1:]
libpath = args.pop(0) if args and args[0].endswith(".so") else _lib.LIB_PATH
_lib.LIB_PATH = libpath
from atomnas_amd import ops
from atomnas_amd.ops import Slab
which = args[0] if len(args) > 0 else "both"
N = int(args[1]) if len(args) > 1 else 256
slab = (args[2] if len(args) > 2 else "slab") == "slab"
lib = _lib.load()


def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


CASES = [(112, 32, 3, 1), (112, 96, 3, 2), (112, 96, 7, 2), (56, 144, 3, 1), (56, 144, 5, 1), (56, 144, 7, 1), (28, 240, 5, 1), (14, 480, 3, 1),
         (14, 480, 7, 1), (14, 576, 5, 1), (7, 1152, 5, 1), (7, 1152, 7, 1)]
if os.environ.get("DWBENCH_CASES"):
    CASES = [tuple(int(v) for v in c.split(",")) for c in os.environ["DWBENCH_CASES"].split(";")]
print(os.path.basename(libpath), "N", N, "slab" if slab else "plain")
totf = totb = 0.0
NSET = int(os.environ.get("DWBENCH_SETS", "3"))   # rotate over several tensor sets: the 256 MiB Infinity Cache must not serve re-runs
ITERS = int(os.environ.get("DWBENCH_ITERS", "10"))
for (H, C, k, s) in CASES:
    Ho = (H - 1) // s + 1
    mk = (lambda M: Slab.from_plain(torch.randn(M, C, device="cuda").bfloat16())) if slab else (lambda M: torch.randn(M, C, device="cuda").bfloat16())
    sets = [(mk(N * H * H), mk(N * Ho * Ho), mk(N * Ho * Ho), mk(N * H * H)) for _ in range(NSET)]
    w = torch.randn(k * k, C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    c1, c2, c3 = torch.rand(C, device="cuda"), torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    rows = ops.stat_rows_for(C)
    st = torch.empty(rows * 2 * C, device="cuda"); dw = torch.zeros(C * k * k, device="cuda")
    ws = torch.empty(rows * C * k * k, device="cuda")
    line = "H%-3d C%-4d k%d s%d:" % (H, C, k, s)
    cnt = [0]
    def fwd():
        x, y, g, h = sets[cnt[0] % NSET]; cnt[0] += 1
        ops.dwconv_fwd(x, sc, sh, True, w, y, st, C, N, H, H, C, k, s, stat_rows=rows)
    def bwd():
        x, y, g, h = sets[cnt[0] % NSET]; cnt[0] += 1
        ops.dwconv_bwd(g, y, c1, c2, c3, x, sc, sh, True, w, h, dw, st, C, N, H, H, C, k, s, stat_rows=rows, dw_ws=ws)
    nx, ny = N * H * H * C, N * Ho * Ho * C
    if which in ("fwd", "both"):
        tf = bench(fwd, ITERS)
        totf += tf
        line += "  fwd %.3f ms %5.0f GB/s" % (tf, (nx + ny) * 2 / tf / 1e6)
    if which in ("bwd", "both"):
        tb = bench(bwd, ITERS)
        totb += tb
        line += "  bwd %.3f ms %5.0f GB/s" % (tb, (2 * nx + ny) * 2 / tb / 1e6)
    print(line, flush=True)
    del sets
print("sum fwd %.3f ms  bwd %.3f ms" % (totf, totb))
