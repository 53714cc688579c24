"""cProfile of one forced shrink of the AtomNAS-C supernet (config 3 of bench.py): where the host time of shrink_model + arena rebuild goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dtype = torch.bfloat16
model, ts, hp, opt, ema, pinfo = bench.build("atomnas_c_supernet", dtype, 64, seed=1995)
x = torch.randn(64, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (64,), device="cuda")
ts.set_batch(x, y)
ts.use_graph = False
for _ in range(2):
    ts.step(lr=0.016, rho=1e-4)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
ms, m0, m1 = bench.forced_shrink(model, ts, opt, ema, pinfo, 0.3, seed=11)
pr.disable()
print("shrink %.1f ms, MACs %d -> %d" % (ms, m0, m1))
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
