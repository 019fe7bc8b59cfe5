"""Tuning aid: the ResNet-50 1x1-convolution GEMM shapes at 800x1333 under the library's tile
variants (default = 64x64 persistent with automatic split-K, 128x64, 128x128), with the
residual + ReLU epilogue of the conv3 layers where it applies."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
def T(fn, n=100):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/n
shapes = [(66800,64,64,0),(66800,256,64,1),(66800,64,256,0),(66800,128,256,0),(16700,512,128,1),(16700,128,512,0),
          (16700,256,512,0),(4200,1024,256,1),(4200,256,1024,0),(4200,512,1024,0),(1050,2048,512,1),(1050,512,2048,0)]
sc = torch.empty(32*1024*1024, device=dev)
for M,N,K,res in shapes:
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.05; b=torch.randn(N,device=dev); o=torch.empty(M,N,device=dev)
    r=torch.randn(M,N,device=dev) if res else None
    row=[]
    for name,kw in (("default",dict()),("tile64 no splitk",dict(force="tile64")),("128x64",dict(force="tile128x64")),("128x128",dict(force="tile"))):
        try:
            us=T(lambda: hip.linear(x,w,b,o,res=r,relu=not res,relu_after=bool(res),scratch=sc if name=="default" else None,**({k:v for k,v in kw.items()})))
            row.append("%s %6.1fus %5.1fTF"%(name,us,2.0*M*N*K/us/1e6))
        except Exception as e:
            row.append("%s ERR"%name)
    print("%6d %5d %5d res=%d | %s"%(M,N,K,res," | ".join(row)))
