"""GPU-box probe: validates the C-ABI/ctypes/torch-stream/hipGraph story. Not product code."""
import ctypes, os, sys, json, time
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "libprobe.so"))
out = {"torch": torch.__version__, "hip": torch.version.hip, "rt": lib.probe_runtime_version()}
with open("/proc/self/maps") as f:
    out["hip_libs"] = sorted({l.split()[-1] for l in f if "libamdhip64" in l})
dev = torch.device("cuda:0")
p = torch.cuda.get_device_properties(0)
out["dev"] = {"name": p.name, "cus": p.multi_processor_count, "mem_gb": p.total_memory / 2**30, "gcn": getattr(p, "gcnArchName", "")}
vp = ctypes.c_void_p
def s(): return vp(torch.cuda.current_stream().cuda_stream)
x = torch.ones(1000, device=dev); y = torch.zeros(1000, device=dev)
rc = lib.probe_axpy(vp(y.data_ptr()), vp(x.data_ptr()), ctypes.c_float(2.0), 1000, s())
torch.cuda.synchronize(); out["axpy_rc"] = rc; out["axpy_ok"] = bool((y == 2).all())
# mfma layout check with asymmetric operands
A = torch.randn(16, 32, device=dev).bfloat16(); Bt = torch.randn(16, 32, device=dev).bfloat16()
D = torch.zeros(16, 16, device=dev)
rc = lib.probe_mfma(vp(A.data_ptr()), vp(Bt.data_ptr()), vp(D.data_ptr()), s())
torch.cuda.synchronize()
ref = A.float() @ Bt.float().t()
out["mfma_rc"] = rc; out["mfma_maxerr"] = float((D - ref).abs().max()); out["mfma_T_maxerr"] = float((D - ref.t()).abs().max())
# graph capture of a ctypes launch
y.zero_()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    lib.probe_axpy(vp(y.data_ptr()), vp(x.data_ptr()), ctypes.c_float(1.0), 1000, s())
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
y.zero_()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10):
        lib.probe_axpy(vp(y.data_ptr()), vp(x.data_ptr()), ctypes.c_float(1.0), 1000, s())
    z = y * 2
torch.cuda.synchronize(); out["after_capture_y"] = float(y[0])
for _ in range(3): g.replay()
torch.cuda.synchronize(); out["after_replay_y"] = float(y[0]); out["z"] = float(z[0])
# launch overhead: eager ctypes vs graph
t0 = time.perf_counter()
for _ in range(1000): lib.probe_axpy(vp(y.data_ptr()), vp(x.data_ptr()), ctypes.c_float(0.0), 1000, s())
torch.cuda.synchronize(); out["eager_us_per_launch"] = (time.perf_counter() - t0) * 1e3
t0 = time.perf_counter()
for _ in range(100): g.replay()
torch.cuda.synchronize(); out["graph_us_per_replay(11 kernels)"] = (time.perf_counter() - t0) * 1e4
# HBM copy bandwidth sanity
big = torch.empty(1 << 30, dtype=torch.uint8, device=dev); big2 = torch.empty_like(big)
big2.copy_(big); torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): big2.copy_(big)
e1.record(); torch.cuda.synchronize()
out["copy_TBps"] = 10 * 2 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e12
out["cpu_count"] = os.cpu_count()
print(json.dumps(out, indent=1))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/probe.json", "w"), indent=1)
