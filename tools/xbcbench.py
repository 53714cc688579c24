import torch, sys, os
sys.path.insert(0,'/root/repo')
from atomnas_amd import ops
BF=torch.bfloat16
def pad(n,m): return (n+m-1)//m*m
def bench(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (M,inp,C) in [(3211264,16,288),(802816,24,432),(200704,40,720)]:
    x=torch.randn(M,inp,device='cuda').to(BF)
    we=torch.zeros(pad(C,64),pad(inp,32),dtype=BF,device='cuda'); we[:C,:inp]=torch.randn(C,inp,device='cuda').to(BF)
    gram=torch.empty(inp*inp,device='cuda'); sx=torch.empty(inp,device='cuda')
    ws=torch.empty(2048*(inp*inp+inp),device='cuda')
    c2=torch.randn(pad(C,8),device='cuda'); c3=torch.randn(pad(C,8),device='cuda')
    mp=torch.zeros(pad(inp,64),pad(inp,32),dtype=BF,device='cuda'); vb=torch.empty(pad(inp,8),device='cuda'); dwe=torch.zeros(C*inp,device='cuda')
    tg=bench(lambda: ops.gram(x,M,inp,gram,sx,ws=ws))
    tc=bench(lambda: ops.xb_coeffs(c2,c3,we,gram,sx,inp,C,mp,vb,dwe))
    print("M%d inp%d C%d: gram %.1f us (x at %.0f GB/s), xb_coeffs %.1f us"%(M,inp,C,tg,M*inp*2/tg/1e3,tc))
