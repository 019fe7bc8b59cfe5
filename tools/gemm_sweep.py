"""Tuning aid (not part of the product path): time pn_gemm_f32 variants on the encoder shapes."""
import sys, time, torch
sys.path.insert(0, '.')
from pairnet_amd import hip
dev = "cuda:0"
def T(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/n
shapes = [(21950,544,256),(21950,256,256),(21950,1024,256),(21950,256,1024),(66800,256,256),(100,66800,256),(16700,256,256),(4200,256,256)]
for M,N,K in shapes:
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev); o=torch.empty(M,N,device=dev)
    row=[]
    for f in ("tile","tile128x64","tile64","skinny"):
        if f=="skinny" and M*N>4e6: row.append("   -  "); continue
        us=T(lambda: hip.linear(x,w,None,o,force=f))
        row.append("%s %6.1fus %5.1fTF" % (f, us, 2.0*M*N*K/us/1e6))
    print(M,N,K," | ".join(row))
