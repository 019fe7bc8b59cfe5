"""Tuning aid (not part of the product path): time pn_gemm_f32 variants on the encoder shapes."""
import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
def T(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/n
shapes = [(21950,544,256),(21950,256,256),(21950,1024,256),(21950,256,1024),(66800,256,256),(100,66800,256),(16700,256,256)]
for M,N,K in shapes:
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.1; o=torch.empty(M,N,device=dev); o2=torch.empty(M,N,device=dev)
    ref=(x[:2048].double()@w.double().t())
    row=[]
    for name,kw in (("tile64",dict(force=None)),("t128x64",dict(force="tile128x64")),("t128",dict(force="tile")),("split64",dict(split=True))):
        us=T(lambda: hip.linear(x,w,None,o,**kw))
        err=(o[:2048].double()-ref).abs().max().item()/ref.abs().max().item()
        row.append("%s %6.1fus %5.1fTF err %.2e" % (name, us, 2.0*M*N*K/us/1e6, err))
    print(M,N,K," | ".join(row))
# conv 3x3
B,H,W,C=1,200,334,256
x=torch.randn(B,H,W,C,device=dev); wp=torch.randn(256,9*C,device=dev)*0.02; o=torch.empty(B,H,W,256,device=dev)
ref=torch.nn.functional.conv2d(x[:, :40].permute(0,3,1,2).double().cpu(), wp.view(256,3,3,C).permute(0,3,1,2).double().cpu(), padding=1)[:, :, :39]
for name,sp,bt,tl in (("f32 64x64",False,False,None),("f32 128x64",False,False,"128x64"),("f32 128x128",False,False,"128"),("split64",True,False,None),("split128",True,True,None)):
    us=T(lambda: hip.conv2d_nhwc(x,wp,None,o,B,H,W,C,256,3,3,1,False,split=sp,big_tile=bt,tile=tl),5)
    err=(o[:, :39].permute(0,3,1,2).double().cpu()-ref).abs().max().item()/ref.abs().max().item()
    print("conv3x3", name, "%.1fus %.1fTF err %.2e" % (us, 2.0*B*H*W*256*9*C/us/1e6, err))
