"""Phase accounting of k_gemm_nt_ws (experiment build with -DWS_TIMING=1):
    bash tools/variant.sh wst pwconv.hip -DWS_TIMING=1
    ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libwst.so python tools/wstiming.py
Per shape: share of wave cycles in: loop top, wait for the chunk's activations, issue + prologue + MFMAs, staging wait + LDS store,
barrier, epilogue, statistics flush."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import _lib, ops
from atomnas_amd.ops import Slab, PRO_BNRELU, PRO_BNBWD, STAT_SQ
lib = _lib.load()
fn = lib.atomnas_debug_ws_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = 256
names = ["top", "wait A", "compute", "stage", "barrier", "epilogue", "flush"]


def pack_w(w):
    n, k = w.shape
    buf = torch.zeros((n + 63) // 64 * 64, (k + 31) // 32 * 32, dtype=torch.bfloat16, device="cuda")
    buf[:n, :k] = w.bfloat16()
    return buf


for (H, inp, hid) in [(56, 24, 432), (28, 40, 720), (14, 96, 1728), (7, 192, 3456)]:
    M = N * H * H
    a = Slab.from_plain(torch.randn(M, hid, device="cuda").bfloat16())
    e = Slab.from_plain(torch.randn(M, hid, device="cuda").bfloat16())
    WT = pack_w(torch.randn(inp, hid, device="cuda") / hid ** 0.5)
    out = torch.empty(M, inp, dtype=torch.bfloat16, device="cuda")
    c1, c2, c3 = torch.rand(hid, device="cuda") + 0.5, torch.randn(hid, device="cuda") * 0.2, torch.randn(hid, device="cuda") * 0.2
    rows = ops.stat_rows_for(inp)
    st = torch.empty(rows * 2 * inp, device="cuda")
    res = torch.randn(M, inp, device="cuda").bfloat16()
    for name, run in (("project", lambda: ops.gemm_nt(a, WT, out, M, inp, hid, a_mode=PRO_BNRELU, ac1=c1, ac2=c2, a_relu=1, stats=st,
                                                      stat_mode=STAT_SQ, stat_rows=rows)),
                      ("dgrad", lambda: ops.gemm_nt(a, WT, out, M, inp, hid, a_mode=PRO_BNBWD, a2=e, ac1=c1, ac2=c2, ac3=c3, add=res))):
        run(); run()
        o = (ctypes.c_ulonglong * 8)()
        fn(None, 1)
        run()
        torch.cuda.synchronize()
        fn(o, 0)
        tot = float(sum(o))
        print("H%d %s %d->%d: %.3g wave-cycles; " % (H, name, hid, inp, tot) + "  ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, o)), flush=True)
