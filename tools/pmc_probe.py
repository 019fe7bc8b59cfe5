"""Tuning aid: run a few GEMM launches for rocprofv3 --pmc collection."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev="cuda:0"
M,N,K=21950,1024,256
x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.1; o=torch.empty(M,N,device=dev)
for _ in range(3):
    hip.linear(x,w,None,o)
    hip.linear(x,w,None,o,split=True)
    hip.linear(x,w,None,o,split=True,force="tile")
torch.cuda.synchronize()
