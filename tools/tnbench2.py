"""gemm_tn shapes of the bs-256 step through the C ABI, cold caches (tensor sets rotated); with a -DTN_TIMING=1 build also the
phase accounting of k_gemm_tn2 (cycles per 128-row slab per wave).   python tools/tnbench2.py [lib.so]"""
import ctypes, sys, os, torch
sys.path.insert(0, "/root/repo")
from atomnas_amd import _lib
if len(sys.argv) > 1: _lib.LIB_PATH = sys.argv[1]
from atomnas_amd import ops
lib = _lib.load()
timing = getattr(lib, "atomnas_debug_tn_timing", None) if hasattr(lib, "atomnas_debug_tn_timing") else None
PH = ["stageV", "stageU", "sync1", "mfma", "sync2", "top"]
p8 = lambda c: (c + 7) // 8 * 8
SHAPES = [(802816, 24, 432, "0,2"), (802816, 24, 432, "2,1"), (200704, 40, 720, "0,2"), (50176, 80, 1440, "0,2"), (50176, 96, 1728, "0,2"),
          (50176, 96, 1728, "2,1"), (12544, 192, 3456, "0,2"), (12544, 192, 3456, "2,1"), (12544, 320, 1280, "0,2")]
tot = 0.0
for (M, NU, NV, mode) in SHAPES:
    sets = []
    nset = max(2, int(1.2e9 // (M * (NU + 2 * NV) * 2)) + 1)
    nset = min(nset, 12)
    for i in range(nset):
        sets.append([torch.randn(M, p8(NU), device="cuda").bfloat16(), torch.randn(M, p8(NU), device="cuda").bfloat16(),
                     torch.randn(M, p8(NV), device="cuda").bfloat16(), torch.randn(M, p8(NV), device="cuda").bfloat16()])
    cu = [torch.rand(p8(NU), device="cuda") for _ in range(3)]; cv = [torch.rand(p8(NV), device="cuda") for _ in range(3)]
    out = torch.zeros(NV, NU, device="cuda")
    cnt = [0]
    def fn():
        U, U2, V, V2 = sets[cnt[0] % nset]; cnt[0] += 1
        if mode == "0,2":
            ops.gemm_tn(U, NU, V, NV, out, 1, NU, M, v_mode=ops.PRO_BNBWD, v2=V2, vc1=cv[0], vc2=cv[1], vc3=cv[2])
        else:
            ops.gemm_tn(U, NU, V, NV, out, 1, NU, M, u_mode=ops.PRO_BNBWD, u2=U2, uc1=cu[0], uc2=cu[1], uc3=cu[2], v_mode=ops.PRO_BNRELU,
                        vc1=cv[0], vc2=cv[1], v_relu=True)
    for _ in range(2): fn()
    torch.cuda.synchronize()
    if timing: timing(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 8
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n; tot += ms
    streams = (NU * (2 if mode == "2,1" else 1) + NV * (2 if mode == "0,2" else 1)) * M * 2
    line = "M%-7d NU%-4d NV%-5d pro%s: %.3f ms  %5.0f GB/s (streams read once)" % (M, NU, NV, mode, ms, streams / ms / 1e6)
    if timing:
        o = (ctypes.c_ulonglong * 8)(); timing(o, 1)
        slabs = max(1, o[7])
        line += " | cyc/slab/wave " + " ".join("%s %d" % (PH[i], o[i] // slabs) for i in range(6))
    print(line, flush=True)
    del sets
print("sum %.3f ms" % tot)
