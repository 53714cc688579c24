"""Phase accounting of k_gemm_tn2 (experiment build with -DTN_TIMING=1):
    bash tools/variant.sh tnt pwconv.hip -DTN_TIMING=1
    ATOMNAS_HIP_LIB=atomnas_amd/csrc/build/variants/libtnt.so python tools/tntiming.py
Per shape: share of wave cycles in: stage (wait for the slab's loads, prologue, transposed LDS stores), issue of the next slab's
loads, barrier, MFMA phase, barrier, loop top."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import _lib, ops
from atomnas_amd.ops import Slab, PRO_BNRELU, PRO_BNBWD
lib = _lib.load()
fn = lib.atomnas_debug_tn_timing
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
N = 256
names = ["stage", "issue next", "barrier A", "MFMA", "barrier B", "top"]
for (H, inp, hid) in [(28, 40, 720), (14, 96, 1728), (7, 192, 3456)]:
    M = N * H * H
    d = Slab.from_plain(torch.randn(M, hid, device="cuda").bfloat16())
    e = Slab.from_plain(torch.randn(M, hid, device="cuda").bfloat16())
    x = torch.randn(M, inp, device="cuda").bfloat16()
    c1, c2, c3 = torch.rand(hid, device="cuda") + 0.5, torch.randn(hid, device="cuda") * 0.2, torch.randn(hid, device="cuda") * 0.2
    out = torch.zeros(inp * hid, device="cuda")
    ws = ops.tn_workspace(inp, hid, x.device)
    for name, run in (("project wgrad (dP, act(bn(D)))", lambda: ops.gemm_tn(x, inp, d, hid, out, hid, 1, M, v_mode=PRO_BNRELU, vc1=c1, vc2=c2, v_relu=1, ws=ws)),
                      ("expand wgrad (x, dE)", lambda: ops.gemm_tn(x, inp, d, hid, out, 1, inp, M, v_mode=PRO_BNBWD, v2=e, vc1=c1, vc2=c2, vc3=c3, ws=ws))):
        run(); run()
        o = (ctypes.c_ulonglong * 8)()
        fn(None, 1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        fn(o, 0)
        tot = float(sum(o[:6]))
        print("H%d %s %d x %d: %.0f us, %d slabs; " % (H, name, inp, hid, e0.elapsed_time(e1) * 1e3, o[7]) +
              "  ".join("%s %.1f%%" % (n, 100.0 * v / tot) for n, v in zip(names, o)), flush=True)
