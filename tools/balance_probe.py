"""Tuning aid: how much of the persistent GEMM's loss is tile-count quantisation?  The head's
shapes (M = 21950 rows) against neighbours whose 64x64 tile count is a whole number of
rounds of the 1024 resident workgroups."""
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
for N,K in ((256,256),(544,256),(1024,256),(256,1024)):
    nt=(N+63)//64
    for M in (21950, 16384, 32768, 65536//nt*4*64//64*16 if False else 24576, 20480):
        x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.1; o=torch.empty(M,N,device=dev)
        us=T(lambda: hip.linear(x,w,None,o))
        tiles=((M+63)//64)*nt
        print("M %6d N %5d K %5d tiles %6d = %.2f rounds of 1024  %7.1f us %6.1f TF" % (M,N,K,tiles,tiles/1024,us,2.0*M*N*K/us/1e6))
