"""Phase accounting of the channel-pair-per-wave depthwise backward (experiment build with -DCW_TIMING=1):
    ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libcwt.so python tools/cwtiming.py
Prints, per shape, the share of wave cycles spent in: wait at barrier A, store+commit, wait at barrier B, issue, compute, tail."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import _lib, ops
from atomnas_amd.ops import Slab
lib = _lib.load()
fn = lib.atomnas_debug_cw_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = 256
CASES = [(56, 144, 3), (56, 144, 7), (28, 240, 3), (14, 480, 3), (14, 480, 7), (7, 1152, 7)]
names = ["waitA", "store+commit", "waitB", "issue", "compute", "final store", "loop top", "reduce+flush"]
for (H, C, k) in CASES:
    mk = lambda M: Slab.from_plain(torch.randn(M, C, device="cuda").bfloat16())
    x, y, g, h = mk(N * H * H), mk(N * H * H), mk(N * H * H), mk(N * H * H)
    w = torch.randn(k * k, C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    c1, c2, c3 = torch.rand(C, device="cuda"), torch.randn(C, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    rows = ops.stat_rows_for(C)
    st = torch.empty(rows * 2 * C, device="cuda"); dw = torch.zeros(C * k * k, device="cuda")
    ws = torch.empty(rows * C * k * k, device="cuda")
    run = lambda: ops.dwconv_bwd(g, y, c1, c2, c3, x, sc, sh, True, w, h, dw, st, C, N, H, H, C, k, 1, stat_rows=rows, dw_ws=ws)
    run(); run()
    out = (ctypes.c_ulonglong * 8)()
    fn(None, 1)
    run()
    fn(out, 0)
    tot = float(sum(out))
    print("H%d C%d k%d: total %.3g wave-cycles; " % (H, C, k, tot) + "  ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, out)), flush=True)
