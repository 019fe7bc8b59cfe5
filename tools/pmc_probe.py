"""Profiling aid: a few launches of the dominant GEMM shapes + MSDA for rocprofv3 --pmc."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pairnet_amd import hip
dev = "cuda:0"
torch.manual_seed(0)
M = 21950
for N, K in ((1024, 256), (256, 1024), (544, 256), (256, 256)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.1
    o = torch.empty(M, N, device=dev)
    for _ in range(3):
        hip.linear(x, w, None, o)
shapes = [(25, 42), (50, 84), (100, 167)]
voa = torch.randn(1, M, 544, device=dev)
out = torch.empty(1, M, 256, device=dev)
for _ in range(3):
    hip.msda(voa, 544, voa.view(-1)[256:], 544, out, 1, shapes)
torch.cuda.synchronize()
