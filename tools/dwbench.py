"""Micro-benchmark of the depthwise kernels through the C ABI (experiments; not a test).

    python tools/dwbench.py [lib.so] [fwd|bwd|both] [N]
Experiment builds with -DDW_TIMING=1 also print the backward kernel's phase accounting (cycles per tile per wave).
"""
import ctypes, sys, os, torch
sys.path.insert(0, "/root/repo")
from atomnas_amd import _lib
args = sys.argv[1:]
libpath = args.pop(0) if args and args[0].endswith(".so") else _lib.LIB_PATH
_lib.LIB_PATH = libpath
from atomnas_amd import ops
which = args[0] if len(args) > 0 else "both"
N = int(args[1]) if len(args) > 1 else 256
lib = _lib.load()
timing = getattr(lib, "atomnas_debug_dw_timing", None) if hasattr(lib, "atomnas_debug_dw_timing") else None
PH = ["sync1", "commit", "sync2", "issue", "x+fma", "epilog", "looptop"]


def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    if timing: timing(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


CASES = [(112, 32, 3, 1), (112, 96, 3, 2), (112, 96, 7, 2), (56, 144, 3, 1), (56, 144, 5, 1), (56, 144, 7, 1), (28, 240, 5, 1), (14, 480, 3, 1),
         (14, 480, 7, 1), (14, 576, 5, 1), (7, 1152, 5, 1), (7, 1152, 7, 1)]
print(os.path.basename(libpath), "N", N)
totf = totb = 0.0
NSET = int(os.environ.get("DWBENCH_SETS", "3"))   # rotate over several tensor sets: the 256 MiB Infinity Cache must not serve re-runs
for (H, C, k, s) in CASES:
    Ho = (H - 1) // s + 1
    sets = []
    for i in range(NSET):
        x = torch.randn(N * H * H, C, device="cuda").bfloat16()
        y = torch.randn(N * Ho * Ho, C, device="cuda").bfloat16()
        g = torch.randn(N * Ho * Ho, C, device="cuda").bfloat16()
        h = torch.zeros(N * H * H, C, device="cuda", dtype=torch.bfloat16)
        sets.append((x, y, g, h))
    w = torch.randn(k * k, C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    c1, c2, c3 = torch.rand(C, device="cuda"), torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    st = torch.zeros(ops.STAT_ROWS * 2 * C, device="cuda"); dw = torch.zeros(C * k * k, device="cuda")
    line = "H%-3d C%-4d k%d s%d:" % (H, C, k, s)
    cnt = [0]
    def fwd():
        x, y, g, h = sets[cnt[0] % NSET]; cnt[0] += 1
        ops.dwconv_fwd(x, sc, sh, True, w, y, st, C, N, H, H, C, k, s)
    def bwd():
        x, y, g, h = sets[cnt[0] % NSET]; cnt[0] += 1
        ops.dwconv_bwd(g, y, c1, c2, c3, x, sc, sh, True, w, h, dw, st, C, N, H, H, C, k, s)
    x, y, g, h = sets[0]
    if which in ("fwd", "both"):
        tf = bench(fwd)
        bf = (x.numel() + y.numel()) * 2; totf += tf
        line += "  fwd %.3f ms %5.0f GB/s" % (tf, bf / tf / 1e6)
    if which in ("bwd", "both"):
        tb = bench(bwd)
        bb = (2 * x.numel() + y.numel()) * 2; totb += tb
        line += "  bwd %.3f ms %5.0f GB/s" % (tb, bb / tb / 1e6)
        if timing:
            out = (ctypes.c_ulonglong * 8)()
            timing(out, 1)
            tiles = max(1, out[7])  # summed over waves
            line += "  | cyc/tile/wave " + " ".join("%s %d" % (PH[i], out[i] // tiles) for i in range(7))
    print(line, flush=True)
    del sets, x, y, g, h
print("sum fwd %.3f ms  bwd %.3f ms" % (totf, totb))
