"""Micro-benchmark of the single-stream weight-gradient GEMMs of the late stages (atomnas_pw_gemm_tn, (NONE, BNRELU) prologue pair).
    python tools/tnbench3.py      env: ATOMNAS_TN_DMA=0/1, ATOMNAS_TN3_DEPTH=2/3/4, ATOMNAS_TN_TR=0/1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from atomnas_amd import ops
from atomnas_amd.ops import Slab
BF = torch.bfloat16
def bench(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
tot = 0.0
for (M, NU, NV, cnt) in [(12544, 192, 3456, 3), (12544, 320, 3456, 1), (12544, 192, 1728, 1), (50176, 96, 1728, 3), (50176, 80, 1440, 3), (50176, 80, 720, 1), (200704, 40, 720, 4)]:
    sets = [(torch.randn(M, NU, device="cuda").to(BF), Slab.from_plain(torch.randn(M, NV, device="cuda").to(BF))) for _ in range(3)]
    sc, sh = torch.rand(NV, device="cuda") + 0.5, torch.randn(NV, device="cuda")
    out = torch.zeros(NU, NV, device="cuda")
    ws = ops.tn_workspace(NU, NV, "cuda")
    i = [0]
    def run():
        u, v = sets[i[0] % 3]; i[0] += 1
        ops.gemm_tn(u, NU, v, NV, out, NV, 1, M, v_mode=ops.PRO_BNRELU, vc1=sc, vc2=sh, v_relu=True, ws=ws)
    t = bench(run)
    tot += t * cnt
    print("M%-7d NU%-4d NV%-5d: %.3f ms  (%4.0f GB/s on V, %5.1f TFLOP/s)  x%d in the step" % (M, NU, NV, t, M * NV * 2 / t / 1e6, 2.0 * M * NU * NV / t / 1e9, cnt), flush=True)
    del sets
print("sum over the step's launches %.3f ms" % tot)
