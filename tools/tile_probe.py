"""Tuning aid: 64x64 vs 128x64 tiles of the persistent fp32 GEMM under SUSTAINED load
(200 back-to-back launches, N(0,1) data) on the shapes of the head and the Swin-L stages."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
def T(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/n
shapes = [(21950,544,256),(21950,256,256),(21950,1024,256),(21950,256,1024),(66800,256,256),(16700,256,256),
          (66800,576,192),(66800,192,192),(66800,768,192),(66800,192,768),
          (16700,1152,384),(16700,384,384),(16700,1536,384),(16700,384,1536),
          (4200,2304,768),(4200,768,768),(4200,3072,768),(4200,768,3072),
          (1050,4608,1536),(1050,1536,1536),(1050,6144,1536),(1050,1536,6144)]
for M,N,K in shapes:
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.1; o=torch.empty(M,N,device=dev)
    sc=torch.empty(16*1024*1024,device=dev)
    row=[]
    ref=(x[:512].double()@w.double().t())
    for name,kw in (("tile64",dict(force="tile64")),("t128x64",dict(force="tile128x64")),("tile64 again",dict(force="tile64"))):
        o.zero_()
        us=T(lambda: hip.linear(x,w,None,o,**kw))
        err=(o[:512].double()-ref).abs().max().item()/ref.abs().max().item()
        row.append("%s %7.1fus %5.1fTF e%.0e" % (name, us, 2.0*M*N*K/us/1e6, err))
    print("%6d %5d %5d"%(M,N,K)," | ".join(row))
