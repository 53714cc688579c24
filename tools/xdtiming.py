"""Phase accounting of the fused expand + depthwise forward (experiment build with -DXD_TIMING=1):
    tools/build_xdw_experiment.sh xdt -DXD_TIMING=1
    ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libxdt.so python tools/xdtiming.py
Prints, per shape, the share of wave cycles per phase of a slab-tile and the cycles per slab-tile and wave."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import _lib, ops
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments"))
import xdw_ops
assert xdw_ops.available(), "load the experiment library (tools/build_xdw_experiment.sh xdt -DXD_TIMING=1)"
from atomnas_amd.ops import Slab
lib = _lib.load()
fn = lib.atomnas_debug_xd_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = 256
BF = torch.bfloat16
CASES = [(56, 24, 144, 3), (56, 24, 144, 5), (56, 24, 144, 7), (28, 40, 240, 3), (28, 40, 240, 7)]
names = ["prefetch issue (+loop top)", "wait barrier 1", "tap stage", "wait barrier 2", "store", "tail", "wait fragments", "mfma + epilogue"]
pad = lambda n, m: (n + m - 1) // m * m
for (H, inp, C, k) in CASES:
    M = N * H * H
    x = torch.randn(M, inp, device="cuda").to(BF)
    D = Slab(M, C, BF, "cuda")
    wexp = torch.zeros(pad(C, 64), pad(inp, 32), dtype=BF, device="cuda")
    wexp[:C, :inp] = (torch.randn(C, inp, device="cuda") / inp ** 0.5).to(BF)
    w = torch.randn(k * k, C, device="cuda")
    sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    rows = ops.stat_rows_for(C)
    st = torch.empty(rows * 2 * C, device="cuda")
    run = lambda: xdw_ops.xdw_fwd(x, inp, wexp, sc, sh, 1, w, D, st, C, N, H, H, C, k, stat_rows=rows)
    run(); run()
    out = (ctypes.c_ulonglong * 8)()
    fn(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record()
    fn(out, 0)
    tot = float(sum(out))
    print("H%d inp%d C%d k%d: %.3f ms, total %.3g wave-cycles; " % (H, inp, C, k, e0.elapsed_time(e1), tot) +
          "  ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, out) if n != "-"), flush=True)
