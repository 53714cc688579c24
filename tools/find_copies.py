"""Which host-side op issues the burst of tiny device copies at the end of backward?  (torch.profiler over one eager step.)"""
import os, sys, torch, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
model, ts, hp = bench.build("atomnas_c_supernet", torch.bfloat16, 32, 1995)[:3]
ts.use_graph = False
ts.set_batch(torch.randn(32, 3, 224, 224, device="cuda"), torch.randint(0, 1000, (32,), device="cuda"))
for _ in range(2): ts.step(rho=1e-4)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    ts.step(rho=1e-4)
    torch.cuda.synchronize()
rows = prof.key_averages()
rows = sorted(rows, key=lambda r: -r.count)
for r in rows:
    if r.count < 20 and not any(w in r.key.lower() for w in ("memcpy", "copy", "memset")): continue
    print("%6d  %-60s cpu %.1f us  cuda %.1f us" % (r.count, r.key[:60], r.cpu_time_total, getattr(r, "device_time_total", getattr(r, "cuda_time_total", 0))))
