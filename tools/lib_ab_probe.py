"""Tuning aid: run the sustained GEMM shapes of the head against an alternative build of the
library (A/B of a kernel change without touching the product .so).
usage: lib_ab_probe.py path/to/libpairnet_hip_variant.so"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
if len(sys.argv) > 1:
    hip.LIB_PATH = os.path.abspath(sys.argv[1])
dev = "cuda:0"
def T(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)*1e3/n
tot = 0.0
for M,N,K in [(21950,544,256),(21950,256,256),(21950,1024,256),(21950,256,1024),(66800,256,256),(16700,256,256)]:
    x=torch.randn(M,K,device=dev); w=torch.randn(N,K,device=dev)*0.1; o=torch.empty(M,N,device=dev)
    T(lambda: hip.linear(x,w,None,o), 50)
    us=T(lambda: hip.linear(x,w,None,o)); tot += us
    print("%6d %5d %5d %7.1f us %6.1f TF" % (M,N,K,us,2.0*M*N*K/us/1e6))
print("sum %.1f us (%s)" % (tot, sys.argv[1] if len(sys.argv) > 1 else "product library"))
