"""Profiling aid: the encoder FFN1 GEMM shape, a few launches, for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
torch.manual_seed(0)
M, N, K = 21950, 1024, 256
x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1
o = torch.empty(M, N, device=dev)
for _ in range(4):
    hip.linear(x, w, None, o)
torch.cuda.synchronize()
